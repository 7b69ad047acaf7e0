"""Adversarial-IRL discriminator step + reward relabelling: restatement of
rlkit/torch/algorithms/adv_irl/adv_irl.py:133-216 (_do_reward_training: BCE-with-logits + WGAN-GP gradient
penalty, double backward), :238-314 (_do_policy_training reward modes) and
rlkit/torch/algorithms/adv_irl/disc_models/simple_disc_models.py:8-48 (MLPDisc: Linear-act-Linear-act-Linear,
output clamped to +-clamp_magnitude; use_bn=False as in exp_specs/gail/gail_walker.yaml:24-28).
numpy fp32 with a hand-derived second-order backward.  Test infrastructure.
"""
import numpy as np

from . import mlp, optim

F32 = np.float32
RELU, TANH = 0, 1


def _act(z, act):
    return np.maximum(z, F32(0)) if act == RELU else np.tanh(z).astype(F32)


def _dact(h, act):   # phi'(z) from the output h
    return (h > 0).astype(F32) if act == RELU else (F32(1) - h * h).astype(F32)


def _d2act_over(h, act):  # d phi'(z) / d z  expressed with h:  tanh: -2 h (1-h^2) ; relu: 0
    return np.zeros_like(h) if act == RELU else (F32(-2) * h * (F32(1) - h * h)).astype(F32)


def bce_with_logits(x, t):
    """torch.nn.BCEWithLogitsLoss (mean): max(x,0) - x t + log(1 + exp(-|x|))."""
    return np.mean(np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x))), dtype=F32)


def sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


class DiscOracle:
    def __init__(self, in_dim, hid_dim, flat, act=TANH, clamp=10.0, disc_lr=3e-4, disc_momentum=0.9,
                 use_grad_pen=True, grad_pen_weight=10.0, num_layer_blocks=2):
        self.L = int(num_layer_blocks)   # simple_disc_models.py:29-39: L x (Linear, act) then Linear(hid, 1); train_step_blocks for L != 2
        self.D, self.H, self.act, self.clamp = in_dim, hid_dim, act, F32(clamp)
        self.p = flat.copy()
        self.lr, self.b1 = disc_lr, disc_momentum
        self.use_gp, self.gp_w = use_grad_pen, grad_pen_weight
        self.opt = optim.AdamState(flat.size)
        # split run (SURVEY section 8e "Disc: same"): one of grad_world ranks with B / G rows per class; the means are over B * G rows and the flat
        # gradient is summed over the ranks (`allreduce`) before Adam.  grad_world = 1, allreduce = None: the reference's single process.
        self.grad_world, self.allreduce = 1, None

    def _layers(self):
        return mlp.unpack(self.p, self.D, [self.H] * self.L, 1)

    def forward_blocks(self, x):
        """any depth: (clamped logit, raw logit, [h_1 .. h_L])"""
        lay, h, hs = self._layers(), x, []
        for W, b in lay[:-1]:
            h = _act(h @ W.T + b, self.act).astype(F32)
            hs.append(h)
        raw = (h @ lay[-1][0].T + lay[-1][1]).astype(F32)
        return np.clip(raw, -self.clamp, self.clamp), raw, hs

    def train_step_blocks(self, x_exp, x_pol, eps_gp):
        """train_step for any number of layer blocks (same loss, adv_irl.py:133-216).  The gradient penalty's gradient, by reverse
        over reverse: with u_L = phi'_L w gate, u_l = phi'_l (W_{l+1}^T u_{l+1}), g = W_1^T u_1 and g-bar = dGP/dg, the cotangents
        run UP the net like a forward pass without biases — x_0 = g-bar, u-bar_l = W_l x_{l-1}, x_l = v-bar_l = phi'_l u-bar_l —
        leaving dW_l += u_l x_{l-1}^T, dw += v-bar_L (gate inside u) and, where phi'' != 0, z-bar_l = (phi''_l / phi'_l) u_l u-bar_l
        (tanh: -2 h_l u_l u-bar_l), which then goes DOWN the ordinary net: delta_L = z-bar_L, delta_l = z-bar_l + phi'_l (W_{l+1}^T
        delta_{l+1}), dW_l += delta_l h_{l-1}^T, db_l += delta_l."""
        lay = self._layers()
        Ws, bs = [W for W, _ in lay], [b for _, b in lay]
        L, act, B = self.L, self.act, x_exp.shape[0]
        x = np.concatenate([x_exp, x_pol], 0).astype(F32)
        t = np.concatenate([np.ones((B, 1), F32), np.zeros((B, 1), F32)], 0)
        logit, raw, hs = self.forward_blocks(x)
        ce = bce_with_logits(logit, t)
        acc = np.mean(((logit > 0).astype(F32) == t).astype(F32))
        gate = ((raw >= -self.clamp) & (raw <= self.clamp)).astype(F32)
        dlogit = (sigmoid(logit) - t) / F32(2 * B * self.grad_world) * gate
        gW, gb = [None] * (L + 1), [None] * (L + 1)
        gW[L], gb[L] = dlogit.T @ hs[-1], dlogit.sum(0)
        d = (dlogit @ Ws[L]) * _dact(hs[L - 1], act)
        for l in range(L - 1, -1, -1):   # layer index l: W_{l+1} in the docstring's numbering
            inp = hs[l - 1] if l > 0 else x
            gW[l], gb[l] = d.T @ inp, d.sum(0)
            if l > 0:
                d = (d @ Ws[l]) * _dact(hs[l - 1], act)
        out = dict(logits=logit, ce_loss=ce, accuracy=acc)
        gp_loss = F32(0)
        if self.use_gp:
            e = eps_gp.astype(F32).reshape(B, 1)
            xh = (e * x_exp + (F32(1) - e) * x_pol).astype(F32)
            _, rawh, gh = self.forward_blocks(xh)
            gt = ((rawh >= -self.clamp) & (rawh <= self.clamp)).astype(F32)
            ph = [_dact(h, act) for h in gh]
            us = [None] * L
            us[L - 1] = (ph[L - 1] * Ws[L]) * gt                     # gate folded into the top of the u chain
            for l in range(L - 2, -1, -1):
                us[l] = ph[l] * (us[l + 1] @ Ws[l + 1])
            g = us[0] @ Ws[0]
            n = np.sqrt(np.sum(g * g, 1, keepdims=True)).astype(F32)
            gp = np.mean((n - F32(1)) ** 2, dtype=F32)
            gp_loss = F32(gp * F32(self.gp_w))
            with np.errstate(divide="ignore", invalid="ignore"):
                gbar = np.where(n > 0, F32(self.gp_w) / F32(B * self.grad_world) * F32(2) * (n - F32(1)) / n * g, F32(0)).astype(F32)
            xs, zb = gbar, [None] * L
            for l in range(L):                                       # up: the linearised forward pass
                gW[l] = gW[l] + us[l].T @ xs
                ub = xs @ Ws[l].T
                zb[l] = (F32(-2) * gh[l] * us[l] * ub).astype(F32) if act == TANH else np.zeros_like(ub)   # (phi''/phi') u u-bar; relu: 0
                xs = (ub * ph[l]).astype(F32)
            gW[L] = gW[L] + xs.sum(0, keepdims=True) * F32(1)       # dw += sum_r v-bar_L  (gate already in u, hence in g-bar's path)
            d = zb[L - 1]                                           # down: the ordinary backward of the z-bar terms
            for l in range(L - 1, -1, -1):
                inp = gh[l - 1] if l > 0 else xh
                gW[l] = gW[l] + d.T @ inp
                gb[l] = gb[l] + d.sum(0)
                if l > 0:
                    d = (d @ Ws[l]) * ph[l - 1] + zb[l - 1]
            out.update(interp=xh, dDdx=g, grad_norm=n)
        out["grad_pen_loss"] = gp_loss
        grad = mlp.pack([(gW[l].astype(F32), gb[l].astype(F32)) for l in range(L + 1)])
        out["grad"] = grad
        if self.allreduce is not None:
            grad = self.allreduce(np.ascontiguousarray(grad, F32))
        optim.adam_step(self.p, grad, self.opt, self.lr, self.b1)
        return out

    def forward(self, x):
        (W1, b1), (W2, b2), (W3, b3) = self._layers()
        h1 = _act(x @ W1.T + b1, self.act).astype(F32)
        h2 = _act(h1 @ W2.T + b2, self.act).astype(F32)
        raw = (h2 @ W3.T + b3).astype(F32)
        return np.clip(raw, -self.clamp, self.clamp), raw, h1, h2

    def logits(self, x):
        return self.forward(np.ascontiguousarray(x, F32))[0]

    def train_step(self, x_exp, x_pol, eps_gp):
        """x_exp / x_pol: [B, D] expert / policy (s,a) rows; eps_gp: [B,1] U(0,1) interpolation weights."""
        (W1, b1), (W2, b2), (W3, b3) = self._layers()
        act, B = self.act, x_exp.shape[0]
        x = np.concatenate([x_exp, x_pol], 0).astype(F32)
        t = np.concatenate([np.ones((B, 1), F32), np.zeros((B, 1), F32)], 0)   # adv_irl.py:80-86
        logit, raw, h1, h2 = self.forward(x)
        ce = bce_with_logits(logit, t)
        acc = np.mean(((logit > 0).astype(F32) == t).astype(F32))
        gate = ((raw >= -self.clamp) & (raw <= self.clamp)).astype(F32)       # torch.clamp passes grad on [min,max]
        dlogit = (sigmoid(logit) - t) / F32(2 * B * self.grad_world) * gate
        d2 = (dlogit @ W3) * _dact(h2, act)
        d1 = (d2 @ W2) * _dact(h1, act)
        gW3, gb3 = dlogit.T @ h2, dlogit.sum(0)
        gW2, gb2 = d2.T @ h1, d2.sum(0)
        gW1, gb1 = d1.T @ x, d1.sum(0)
        out = dict(logits=logit, ce_loss=ce, accuracy=acc)
        gp_loss = F32(0)
        if self.use_gp:
            e = eps_gp.astype(F32).reshape(B, 1)
            xh = (e * x_exp + (F32(1) - e) * x_pol).astype(F32)                 # adv_irl.py:187-189
            _, rawh, g1, g2 = self.forward(xh)
            gt = ((rawh >= -self.clamp) & (rawh <= self.clamp)).astype(F32)     # [B,1]
            p1, p2 = _dact(g1, act), _dact(g2, act)
            u2 = p2 * W3                                                        # [B,H]
            v1 = u2 @ W2                                                        # [B,H]  (W2^T u2 per row)
            u1 = p1 * v1
            g = gt * (u1 @ W1)                                                  # [B,D]  dD/dx
            n = np.sqrt(np.sum(g * g, 1, keepdims=True)).astype(F32)
            gp = np.mean((n - F32(1)) ** 2, dtype=F32)
            gp_loss = F32(gp * F32(self.gp_w))
            # dGP/dg; a clamped interpolate has g == 0 and torch's norm backward is 0 there (masked_fill), not 0/0
            with np.errstate(divide="ignore", invalid="ignore"):
                gbar = np.where(n > 0, F32(self.gp_w) / F32(B * self.grad_world) * F32(2) * (n - F32(1)) / n * g, F32(0)).astype(F32)
            gW1 += (gt * u1).T @ gbar
            u1b = gt * (gbar @ W1.T)
            v1b, p1b = u1b * p1, u1b * v1
            gW2 += u2.T @ v1b
            u2b = v1b @ W2.T
            gW3 += (u2b * p2).sum(0, keepdims=True)
            p2b = u2b * W3
            z2b = p2b * _d2act_over(g2, act)
            gW2 += z2b.T @ g1
            gb2 += z2b.sum(0)
            h1b = z2b @ W2
            z1b = h1b * p1 + p1b * _d2act_over(g1, act)
            gW1 += z1b.T @ xh
            gb1 += z1b.sum(0)
            out.update(interp=xh, dDdx=g, grad_norm=n)
        out["grad_pen_loss"] = gp_loss
        grad = mlp.pack([(gW1.astype(F32), gb1.astype(F32)), (gW2.astype(F32), gb2.astype(F32)),
                         (gW3.astype(F32), gb3.astype(F32))])
        out["grad"] = grad
        if self.allreduce is not None:
            grad = self.allreduce(np.ascontiguousarray(grad, F32))
        optim.adam_step(self.p, grad, self.opt, self.lr, self.b1)   # adv_irl.py:75-77 betas (disc_momentum, 0.999)
        return out


def softplus(x, beta):
    """torch F.softplus(x, beta, threshold=20): (1/beta) log(1 + exp(beta x)), linear where beta*x > 20."""
    bx = F32(beta) * x
    return np.where(bx > 20, x, np.log1p(np.exp(np.minimum(bx, F32(20)))) / F32(beta)).astype(F32)


def disc_reward(logits, mode, rew_clip_min=None, rew_clip_max=None):
    """adv_irl.py:277-298."""
    x = logits.astype(F32)
    if mode == "airl":
        r = x
    elif mode == "gail":
        r = softplus(x, 1.0)
    elif mode == "gail2":
        r = softplus(x, -1.0)
    elif mode == "fairl":
        r = np.exp(x) * (F32(-1) * x)
    else:
        raise ValueError(mode)
    if rew_clip_max is not None:
        r = np.minimum(r, F32(rew_clip_max))
    if rew_clip_min is not None:
        r = np.maximum(r, F32(rew_clip_min))
    return r.astype(F32)


class DiscBNOracle:
    """MLPDisc(use_bn=True) — the reference constructor's DEFAULT (simple_disc_models.py:15,30-31,36-37): every block is
    Linear -> BatchNorm1d -> act.  One discriminator step = AdvIRL._do_reward_training (adv_irl.py:133-216) with the module in train mode:
    the cross-entropy forward normalises the 2B rows with THEIR batch statistics, the gradient-penalty forward the B interpolates with
    theirs, both update the running statistics (momentum 0.1, unbiased variance), and the penalty's gradient is a double backward
    THROUGH the batch statistics.  `logits(x)` is the eval-mode forward (running statistics) that _do_policy_training uses
    (adv_irl.py:268-274).  numpy fp32; derivation in train_step's comments.  Test infrastructure.

    Flat parameter order = torch's parameters(): per block W [H, in], b [H], gamma [H], beta [H]; then w [1, H], c [1].
    Buffers (not parameters): running_mean / running_var per block."""

    EPS, MOM = F32(1e-5), F32(0.1)   # torch.nn.BatchNorm1d defaults

    def __init__(self, in_dim, hid_dim, flat, act=TANH, clamp=10.0, disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=True,
                 grad_pen_weight=10.0, num_layer_blocks=2):
        self.L, self.D, self.H, self.act, self.clamp = int(num_layer_blocks), in_dim, hid_dim, act, F32(clamp)
        self.p = np.asarray(flat, F32).copy()
        assert self.p.size == self.n_params(in_dim, hid_dim, self.L)
        self.lr, self.b1, self.use_gp, self.gp_w = disc_lr, disc_momentum, use_grad_pen, grad_pen_weight
        self.opt = optim.AdamState(self.p.size)
        self.rm = [np.zeros(hid_dim, F32) for _ in range(self.L)]
        self.rv = [np.ones(hid_dim, F32) for _ in range(self.L)]

    @staticmethod
    def n_params(D, H, L):
        return sum((D if l == 0 else H) * H + 3 * H for l in range(L)) + H + 1

    @staticmethod
    def init(rng, D, H, L, w_scale=0.3):
        """a parameter vector in the flat order: weights U(+-w_scale / sqrt(in)), small biases, gamma near 1, beta near 0"""
        parts = []
        for l in range(L):
            i = D if l == 0 else H
            parts += [rng.uniform(-1, 1, (H, i)) * w_scale * 3 / np.sqrt(i), rng.normal(0, 0.05, H), 1.0 + rng.normal(0, 0.1, H), rng.normal(0, 0.1, H)]
        parts += [rng.uniform(-1, 1, (1, H)) * w_scale, rng.normal(0, 0.05, 1)]
        return np.concatenate([np.asarray(a, F32).ravel() for a in parts]).astype(F32)

    def unpack(self, flat=None):
        flat = self.p if flat is None else flat
        o, blocks = 0, []
        for l in range(self.L):
            i = self.D if l == 0 else self.H
            W = flat[o:o + self.H * i].reshape(self.H, i); o += self.H * i
            b, g, be = flat[o:o + self.H], flat[o + self.H:o + 2 * self.H], flat[o + 2 * self.H:o + 3 * self.H]; o += 3 * self.H
            blocks.append((W, b, g, be))
        w = flat[o:o + self.H].reshape(1, self.H); c = flat[o + self.H:o + self.H + 1]
        return blocks, w, c

    def dead_bias_mask(self):
        """flat-vector mask of the Linear biases that sit under a BatchNorm: their gradient is exactly 0 (the batch mean is subtracted), what any
        implementation holds there is rounding noise that Adam turns into +-lr steps; they do not influence the function"""
        m, o = np.zeros(self.p.size, bool), 0
        for l in range(self.L):
            o += self.H * (self.D if l == 0 else self.H)
            m[o:o + self.H] = True
            o += 3 * self.H
        return m

    def pack(self, gblocks, gw, gc):
        return np.concatenate([np.asarray(a, F32).ravel() for blk in gblocks for a in blk] + [np.asarray(gw, F32).ravel(), np.asarray(gc, F32).ravel()])

    # ---- forward passes
    def forward_train(self, x, update_running=True):
        """train-mode forward of n rows: (clamped logit, raw, per block dict(x_in, chat = a - mu, s = 1/sqrt(var + eps), ahat, y, h, p = phi'(y)))"""
        blocks, w, c = self.unpack()
        n_rows, h, tape = x.shape[0], x.astype(F32), []
        for l, (W, b, g, be) in enumerate(blocks):
            a = (h @ W.T + b).astype(F32)
            mu = a.mean(0, dtype=F32)
            ch = (a - mu).astype(F32)
            var = np.mean(ch * ch, 0, dtype=F32)
            s = (F32(1) / np.sqrt(var + self.EPS)).astype(F32)
            ah = (ch * s).astype(F32)
            y = (g * ah + be).astype(F32)
            hh = _act(y, self.act).astype(F32)
            tape.append(dict(x=h, ch=ch, s=s, ah=ah, y=y, h=hh, p=_dact(hh, self.act)))
            if update_running:   # torch: running = (1 - momentum) running + momentum batch; the variance enters UNBIASED
                self.rm[l] = ((F32(1) - self.MOM) * self.rm[l] + self.MOM * mu).astype(F32)
                self.rv[l] = ((F32(1) - self.MOM) * self.rv[l] + self.MOM * var * F32(n_rows / (n_rows - 1))).astype(F32)
            h = hh
        raw = (h @ w.T + c).astype(F32)
        return np.clip(raw, -self.clamp, self.clamp), raw, tape

    def logits(self, x):
        """eval-mode forward (module.eval(), adv_irl.py:268-274): running statistics"""
        blocks, w, c = self.unpack()
        h = np.ascontiguousarray(x, F32)
        for l, (W, b, g, be) in enumerate(blocks):
            a = (h @ W.T + b).astype(F32)
            y = (g * ((a - self.rm[l]) / np.sqrt(self.rv[l] + self.EPS)) + be).astype(F32)
            h = _act(y, self.act).astype(F32)
        return np.clip((h @ w.T + c).astype(F32), -self.clamp, self.clamp)

    def _bn_backward(self, t, dy, g):
        """cotangent of the pre-normalisation activations a from the cotangent dy of y = gamma ahat + beta (batch statistics are functions
        of a): da = s (dahat - mean(dahat) - ahat mean(dahat ahat)); also (dgamma, dbeta)"""
        dah = (dy * g).astype(F32)
        m1, m2 = dah.mean(0, dtype=F32), (dah * t["ah"]).mean(0, dtype=F32)
        return (t["s"] * (dah - m1 - t["ah"] * m2)).astype(F32), (dy * t["ah"]).sum(0, dtype=F32), dy.sum(0, dtype=F32)

    def train_step(self, x_exp, x_pol, eps_gp):
        blocks, w, c = self.unpack()
        L, act, B = self.L, self.act, x_exp.shape[0]
        x = np.concatenate([x_exp, x_pol], 0).astype(F32)
        tgt = np.concatenate([np.ones((B, 1), F32), np.zeros((B, 1), F32)], 0)
        logit, raw, tape = self.forward_train(x)
        ce = bce_with_logits(logit, tgt)
        acc = np.mean(((logit > 0).astype(F32) == tgt).astype(F32))
        gate = ((raw >= -self.clamp) & (raw <= self.clamp)).astype(F32)
        dlogit = ((sigmoid(logit) - tgt) / F32(2 * B) * gate).astype(F32)
        gB = [[np.zeros_like(W), np.zeros_like(b), np.zeros_like(g), np.zeros_like(be)] for (W, b, g, be) in blocks]
        gw, gc = (dlogit.T @ tape[-1]["h"]).astype(F32), dlogit.sum(0, dtype=F32)
        dh = (dlogit @ w).astype(F32)
        for l in range(L - 1, -1, -1):   # ordinary backward through Linear -> BN -> act
            W, b, g, be = blocks[l]
            t = tape[l]
            da, dg, dbe = self._bn_backward(t, (dh * t["p"]).astype(F32), g)
            gB[l][0] += da.T @ t["x"]; gB[l][1] += da.sum(0, dtype=F32); gB[l][2] += dg; gB[l][3] += dbe
            dh = (da @ W).astype(F32)
        out = dict(logits=logit, ce_loss=ce, accuracy=acc)
        gp_loss = F32(0)
        if self.use_gp:
            e = eps_gp.astype(F32).reshape(B, 1)
            xh = (e * x_exp + (F32(1) - e) * x_pol).astype(F32)                          # adv_irl.py:187-189
            _, rawh, tp = self.forward_train(xh)                                         # its own batch statistics (and a running-statistics update)
            gt = ((rawh >= -self.clamp) & (rawh <= self.clamp)).astype(F32)
            n_rows = F32(B)
            # ---- first backward: g = d(sum_r D(xh_r)) / d xh, THROUGH the batch statistics.  Per block (top down), with uh the cotangent of h:
            #      uy = uh p ; uah = uy gamma ; m1 = mean(uah) ; m2 = mean(uah ahat) ; tt = uah - m1 - ahat m2 ; ua = s tt ; ux = ua W
            fb = [None] * L
            uh = (gt * w).astype(F32)                                                    # [B, H]: the clamp's gate times the output weights
            for l in range(L - 1, -1, -1):
                W, b, g, be = blocks[l]
                t = tp[l]
                uy = (uh * t["p"]).astype(F32)
                uah = (uy * g).astype(F32)
                m1, m2 = uah.mean(0, dtype=F32), (uah * t["ah"]).mean(0, dtype=F32)
                tt = (uah - m1 - t["ah"] * m2).astype(F32)
                ua = (t["s"] * tt).astype(F32)
                fb[l] = dict(uh=uh, uy=uy, uah=uah, m2=m2, tt=tt, ua=ua)
                uh = (ua @ W).astype(F32)
            gvec = uh                                                                    # dD/dx [B, D]
            nrm = np.sqrt(np.sum(gvec * gvec, 1, keepdims=True)).astype(F32)
            gp = np.mean((nrm - F32(1)) ** 2, dtype=F32)
            gp_loss = F32(gp * F32(self.gp_w))
            with np.errstate(divide="ignore", invalid="ignore"):
                xbar = np.where(nrm > 0, F32(self.gp_w) / F32(B) * F32(2) * (nrm - F32(1)) / nrm * gvec, F32(0)).astype(F32)
            # ---- reverse of the first backward, bottom up (xbar = adjoint of ux of the block below).  Leaves: adjoints of the parameters the
            #      first backward read (W, gamma, w) and of the FORWARD quantities it read: ybar (through p = phi'(y)), ahbar, sbar.
            ybar, ahbar, sbar = [None] * L, [None] * L, [None] * L
            for l in range(L):
                W, b, g, be = blocks[l]
                t, f = tp[l], fb[l]
                gB[l][0] += f["ua"].T @ xbar                                             # ux = ua W
                uabar = (xbar @ W.T).astype(F32)
                sbar[l] = (uabar * f["tt"]).sum(0, dtype=F32)                            # ua = s tt
                ttbar = (uabar * t["s"]).astype(F32)
                m1bar, m2bar = -ttbar.sum(0, dtype=F32), -(ttbar * t["ah"]).sum(0, dtype=F32)
                uahbar = (ttbar + m1bar / n_rows + (m2bar / n_rows) * t["ah"]).astype(F32)   # tt = uah - mean(uah) - ahat mean(uah ahat)
                ahbar[l] = (-ttbar * f["m2"] + (m2bar / n_rows) * f["uah"]).astype(F32)
                gB[l][2] += (uahbar * f["uy"]).sum(0, dtype=F32)                         # uah = uy gamma
                uybar = (uahbar * g).astype(F32)
                ybar[l] = (uybar * f["uh"] * _d2act_over(t["h"], act)).astype(F32)        # uy = uh phi'(y): d phi'/dy = phi''  (relu: 0)
                xbar = (uybar * t["p"]).astype(F32)                                       # adjoint of uh = ux of the block above
            gw = gw + (xbar * gt).sum(0, keepdims=True, dtype=F32)                       # uh_L = gate w
            # ---- and those adjoints go DOWN the forward graph like an ordinary backward with extra sources
            hbar = np.zeros((B, self.H), F32)
            for l in range(L - 1, -1, -1):
                W, b, g, be = blocks[l]
                t = tp[l]
                yb = (ybar[l] + hbar * t["p"]).astype(F32)
                gB[l][2] += (yb * t["ah"]).sum(0, dtype=F32); gB[l][3] += yb.sum(0, dtype=F32)      # y = gamma ahat + beta
                ahb = (ahbar[l] + yb * g).astype(F32)
                chb = (ahb * t["s"]).astype(F32)                                          # ahat = chat s
                sb = (sbar[l] + (ahb * t["ch"]).sum(0, dtype=F32)).astype(F32)
                vb = (F32(-0.5) * sb * t["s"] ** 3).astype(F32)                           # s = (var + eps)^(-1/2)
                chb = (chb + vb * F32(2) * t["ch"] / n_rows).astype(F32)                  # var = mean(chat^2)
                ab = (chb - chb.mean(0, dtype=F32)).astype(F32)                           # chat = a - mean(a)
                gB[l][0] += ab.T @ t["x"]; gB[l][1] += ab.sum(0, dtype=F32)
                hbar = (ab @ W).astype(F32)
            out.update(interp=xh, dDdx=gvec, grad_norm=nrm)
        out["grad_pen_loss"] = gp_loss
        grad = self.pack(gB, gw, gc)
        out["grad"] = grad
        optim.adam_step(self.p, grad, self.opt, self.lr, self.b1)
        return out
