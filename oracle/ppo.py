"""PPO: restatement of rlkit/torch/algorithms/ppo/ppo.py:57-100 (calc_adv: per-trajectory GAE with zero
bootstrap, per-trajectory advantage standardisation with the UNBIASED std) and :102-170 (train_step:
update_epoch x shuffled minibatches; value loss = MSE + value_l2_reg * sum_p ||p||^2 over ALL vf parameters;
clipped surrogate; grad-norm clip 20; the entropy term is computed but never used), with the policy of
rlkit/torch/common/policies.py:348-478 (ReparamMultivariateGaussianPolicy, conditioned_std=False: tanh-hidden
MLP mean, state-independent `action_log_std` parameter, last_fc.weight*0.1 / bias*0) and the tanh value net of
run_scripts/ppo_exp_script.py:82-96.  numpy fp32.  Test infrastructure.

Flat policy layout here (and in libilsx): fc0.W|fc0.b|fc1.W|fc1.b|last_fc.W|last_fc.b|action_log_std[a]
(torch's parameters() yields action_log_std FIRST; tools/make_golden.py reorders).
"""
import numpy as np

from . import mlp, optim
from .tanh_gaussian import LOG_SIG_MAX, LOG_SIG_MIN, gaussian_log_prob

F32 = np.float32


def gae_one_traj(values, rewards, discount, tau, bootstrap=0.0):
    """ppo.py:73-86.  values, rewards: [T,1].  Returns (returns, normalised advantages, raw advantages).  bootstrap: V of the
    observation after the last sample — 0 in the reference (ppo.py:74), V(s_T) for a segment cut by the end of a fixed rollout."""
    T = rewards.shape[0]
    deltas = np.zeros_like(values)
    adv = np.zeros_like(values)
    prev_v, prev_a = F32(bootstrap), F32(0)
    for i in reversed(range(T)):
        deltas[i] = rewards[i] + F32(discount) * prev_v - values[i]
        adv[i] = deltas[i] + F32(discount) * F32(tau) * prev_a
        prev_v, prev_a = values[i, 0], adv[i, 0]
    returns = (values + adv).astype(F32)
    std = np.std(adv.astype(np.float64), ddof=1) if T > 1 else np.nan   # torch.std is unbiased; T=1 -> nan
    nadv = ((adv - np.mean(adv, dtype=F32)) / F32(std)).astype(F32)
    return returns, nadv, adv


class PPOOracle:
    def __init__(self, obs_dim, act_dim, hidden, pi_flat, vf_flat, reward_scale=1.0, discount=0.99, clip_eps=0.2,
                 policy_lr=3e-4, value_lr=3e-4, gae_tau=0.9, value_l2_reg=1e-3, mini_batch_size=64, update_epoch=10,
                 use_value_clip=False, conditioned_std=False):
        # conditioned_std (policies.py:368-374,401-405): log_std = clamp(last_fc_log_std(h), LOG_SIG_MIN, LOG_SIG_MAX), a second head of the
        # policy net (flat layout fc.. | last_fc | last_fc_log_std: torch's parameters() order) instead of the action_log_std parameter
        self.cond = bool(conditioned_std)
        self.o, self.a, self.hidden = obs_dim, act_dim, list(hidden)
        self.pi, self.vf = pi_flat.copy(), vf_flat.copy()
        self.reward_scale, self.discount, self.clip_eps = reward_scale, discount, clip_eps
        self.policy_lr, self.value_lr, self.tau, self.l2 = policy_lr, value_lr, gae_tau, value_l2_reg
        self.mb, self.epochs = mini_batch_size, update_epoch
        self.use_value_clip = use_value_clip
        self.opt_pi, self.opt_vf = optim.AdamState(pi_flat.size), optim.AdamState(vf_flat.size)
        # split run (SURVEY section 8e "PPO split: same, per-minibatch grads"): this instance is one of grad_world ranks, holds mini_batch / G rows of
        # every minibatch, scales its mean-loss gradients by 1 / (rows * G) and sums them over the ranks (`allreduce`: flat gradient -> summed
        # flat gradient) before the L2 term, the norm clip and Adam.  grad_world = 1, allreduce = None: the reference's single process.
        self.grad_world, self.allreduce = 1, None

    # ---- networks (tanh hidden)
    def v(self, obs):
        outs, hs = mlp.forward(self.vf, obs, self.o, self.hidden, 1, act=mlp.TANH)
        return outs[0], hs

    def pi_mean(self, obs):
        if self.cond:
            outs, hs = mlp.forward(self.pi, obs, self.o, self.hidden, self.a, n_heads=2, act=mlp.TANH)
            self._lsr = outs[1]
            return outs[0], hs
        n = self.pi.size - self.a
        outs, hs = mlp.forward(self.pi[:n], obs, self.o, self.hidden, self.a, act=mlp.TANH)
        return outs[0], hs

    def log_prob(self, obs, act):
        mu, hs = self.pi_mean(obs)
        if self.cond:
            ls = np.clip(self._lsr, F32(LOG_SIG_MIN), F32(LOG_SIG_MAX)).astype(F32)
        else:
            ls = np.broadcast_to(self.pi[-self.a:], mu.shape).astype(F32)
        return gaussian_log_prob(mu, ls, act), mu, ls, hs

    # ---- ppo.py:57-100
    def calc_adv(self, trajs):
        obs, act, ret, adv, val = [], [], [], [], []
        for tr in trajs:
            o = tr["observations"].astype(F32)
            r = (F32(self.reward_scale) * tr["rewards"].astype(F32)).reshape(-1, 1)
            v, _ = self.v(o)
            R, A, _ = gae_one_traj(v, r, self.discount, self.tau)
            obs.append(o); act.append(tr["actions"].astype(F32)); ret.append(R); adv.append(A); val.append(v)
        return (np.concatenate(obs), np.concatenate(act), np.concatenate(ret), np.concatenate(adv), np.concatenate(val))

    def value_step(self, ob, R, v_old=None):
        mbn = ob.shape[0]
        v, hs = self.v(ob)
        if self.use_value_clip:   # ppo.py:137-143
            eps = F32(self.clip_eps)
            dv = v - v_old
            vc = v_old + np.clip(dv, -eps, eps)
            l1, l2 = (v - R) ** 2, (vc - R) ** 2
            mse = np.mean(np.maximum(l1, l2), dtype=F32)
            w = np.where(l1 > l2, F32(1), np.where(l1 == l2, F32(0.5), F32(0)))            # torch.max tie rule
            inside = ((dv >= -eps) & (dv <= eps)).astype(F32)                               # clamp passes grad on [-eps, eps]
            dhead = F32(2) * (w * (v - R) + (F32(1) - w) * (vc - R) * inside) / F32(mbn * self.grad_world)
        else:
            mse = np.mean((v - R) ** 2, dtype=F32)
            dhead = F32(2) * (v - R) / F32(mbn * self.grad_world)
        loss = mse + F32(self.l2) * np.sum(self.vf ** 2, dtype=F32)   # ppo.py:145-148
        g, _ = mlp.backward(self.vf, hs, [dhead.astype(F32)], self.o, self.hidden, 1, act=mlp.TANH, need_dx=False)
        if self.allreduce is not None:
            g = self.allreduce(np.ascontiguousarray(g, F32))
        g = (g + F32(2 * self.l2) * self.vf).astype(F32)
        optim.adam_step(self.vf, g, self.opt_vf, self.value_lr)
        return loss, g

    def policy_step(self, ob, ac, A, lp_old):
        mbn = ob.shape[0]
        lp, mu, ls, hs = self.log_prob(ob, ac)
        ratio = np.exp(lp - lp_old)
        clipped = np.clip(ratio, F32(1 - self.clip_eps), F32(1 + self.clip_eps))
        s1, s2 = ratio * A, clipped * A
        loss = -np.mean(np.minimum(s1, s2), dtype=F32)                                              # ppo.py:158-164
        inside = (ratio >= F32(1 - self.clip_eps)) & (ratio <= F32(1 + self.clip_eps))             # clamp passes grad on [lo,hi]
        w = np.where(s1 < s2, F32(1), np.where(s1 == s2, F32(0.5), F32(0)))                        # torch.min tie rule
        dratio = -(w * A + (F32(1) - w) * A * inside) / F32(mbn * self.grad_world)
        dlp = dratio * ratio
        var = np.exp(F32(2) * ls)
        dmu = dlp * (ac - mu) / var
        if self.cond:   # per-row gradient into the log-std head, through the clamp's gate
            gate = ((self._lsr >= F32(LOG_SIG_MIN)) & (self._lsr <= F32(LOG_SIG_MAX))).astype(F32)
            dlsr = (dlp * ((ac - mu) ** 2 / var - F32(1)) * gate).astype(F32)
            g, _ = mlp.backward(self.pi, hs, [dmu.astype(F32), dlsr], self.o, self.hidden, self.a, n_heads=2, act=mlp.TANH, need_dx=False)
        else:
            dls = np.sum(dlp * ((ac - mu) ** 2 / var - F32(1)), axis=0)
            n = self.pi.size - self.a
            gm, _ = mlp.backward(self.pi[:n], hs, [dmu.astype(F32)], self.o, self.hidden, self.a, act=mlp.TANH, need_dx=False)
            g = np.concatenate([gm, dls.astype(F32)])
        if self.allreduce is not None:
            g = self.allreduce(np.ascontiguousarray(g, F32))
        norm = np.sqrt(np.sum(g.astype(np.float64) ** 2))
        coef = 20.0 / (norm + 1e-6)                                                                 # clip_grad_norm_(…, 20)
        gc = (g * F32(coef)).astype(F32) if coef < 1.0 else g
        optim.adam_step(self.pi, gc, self.opt_pi, self.policy_lr)
        return loss, g, norm

    def train_step(self, trajs, perms):
        """perms: list (one per epoch) of index permutations (torch.randperm in the reference, ppo.py:116)."""
        obs, act, R, A, V = self.calc_adv(trajs)
        lp_old = self.log_prob(obs, act)[0]
        out = dict(returns=R, advantages=A, fixed_log_probs=lp_old)
        for perm in perms:
            for s in range(0, len(perm), self.mb):
                ind = perm[s:s + self.mb]
                out["vf_loss"], out["vf_grad"] = self.value_step(obs[ind], R[ind], V[ind])
                out["pg_loss"], out["pi_grad"], out["pi_grad_norm"] = self.policy_step(obs[ind], act[ind], A[ind], lp_old[ind])
        return out
