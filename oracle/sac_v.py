"""SAC with a state-value function: restatement of rlkit/torch/algorithms/sac/sac.py:23-68 (ctor), :70-179
(train_step) and :242-243 (soft update of V only).  Fixed temperature `alpha` (default 1.0).  numpy fp32.
Test infrastructure.

Order that parity depends on (sac.py:93-139): Q and V losses are all evaluated at the PRE-update parameters and with
ONE policy sample (a~, log pi); qf1/qf2/vf Adam; the policy loss re-evaluates Q1/Q2(s, a~) with the UPDATED critics
but re-uses that same sample (:150-156).
"""
import numpy as np

from . import mlp, optim
from . import tanh_gaussian as tg

F32 = np.float32


class SacVOracle:
    def __init__(self, obs_dim, act_dim, hidden, pi, q1, q2, vf, reward_scale=1.0, discount=0.99, alpha=1.0,
                 policy_lr=1e-3, qf_lr=1e-3, vf_lr=1e-3, soft_target_tau=1e-2, policy_mean_reg_weight=1e-3,
                 policy_std_reg_weight=1e-3, beta_1=0.9):
        self.o, self.a, self.hidden = obs_dim, act_dim, list(hidden)
        self.pi, self.q1, self.q2, self.vf, self.tvf = pi.copy(), q1.copy(), q2.copy(), vf.copy(), vf.copy()
        self.reward_scale, self.discount, self.alpha, self.tau, self.beta_1 = reward_scale, discount, alpha, soft_target_tau, beta_1
        self.policy_lr, self.qf_lr, self.vf_lr = policy_lr, qf_lr, vf_lr
        self.w_mu, self.w_std = policy_mean_reg_weight, policy_std_reg_weight
        self.opt = {k: optim.AdamState(getattr(self, k).size) for k in ("pi", "q1", "q2", "vf")}

    def _q(self, flat, s, a):
        outs, hs = mlp.forward(flat, np.concatenate([s, a], axis=1).astype(F32), self.o + self.a, self.hidden, 1)
        return outs[0], hs

    def _v(self, flat, s):
        outs, hs = mlp.forward(flat, s, self.o, self.hidden, 1)
        return outs[0], hs

    def train_step(self, batch, eps):
        B = batch["observations"].shape[0]
        s, a, s2 = (batch[k].astype(F32) for k in ("observations", "actions", "next_observations"))
        r = (F32(self.reward_scale) * batch["rewards"].astype(F32)).reshape(B, 1)
        d = batch["terminals"].astype(F32).reshape(B, 1)
        inv, alpha = F32(1.0) / F32(B), F32(self.alpha)
        out = {}
        # ---- QF loss (sac.py:93-105)
        q1, h1 = self._q(self.q1, s, a)
        q2, h2 = self._q(self.q2, s, a)
        y = (r + (F32(1) - d) * F32(self.discount) * self._v(self.tvf, s2)[0]).astype(F32)
        # ---- VF loss (sac.py:120-131), policy sampled once
        v, hv = self._v(self.vf, s)
        outs, hs_pi = mlp.forward(self.pi, s, self.o, self.hidden, self.a, n_heads=2)
        fw = tg.head_forward(outs[0], outs[1], eps)
        qmin0 = np.minimum(self._q(self.q1, s, fw["action"])[0], self._q(self.q2, s, fw["action"])[0])
        vt = (qmin0 - alpha * fw["log_prob"]).astype(F32)
        out.update(q1_pred=q1, q2_pred=q2, q_target=y, v_pred=v, v_target=vt, log_pi=fw["log_prob"], new_actions=fw["action"],
                   qf1_loss=F32(0.5) * np.mean((q1 - y) ** 2, dtype=F32), qf2_loss=F32(0.5) * np.mean((q2 - y) ** 2, dtype=F32),
                   vf_loss=F32(0.5) * np.mean((v - vt) ** 2, dtype=F32))
        g1, _ = mlp.backward(self.q1, h1, [(q1 - y) * inv], self.o + self.a, self.hidden, 1, need_dx=False)
        g2, _ = mlp.backward(self.q2, h2, [(q2 - y) * inv], self.o + self.a, self.hidden, 1, need_dx=False)
        gv, _ = mlp.backward(self.vf, hv, [(v - vt) * inv], self.o, self.hidden, 1, need_dx=False)
        optim.adam_step(self.q1, g1, self.opt["q1"], self.qf_lr, self.beta_1)
        optim.adam_step(self.q2, g2, self.opt["q2"], self.qf_lr, self.beta_1)
        optim.adam_step(self.vf, gv, self.opt["vf"], self.vf_lr, self.beta_1)
        out.update(q1_grad=g1, q2_grad=g2, vf_grad=gv)
        # ---- policy loss with the updated critics, same sample (sac.py:150-165)
        q1n, hq1 = self._q(self.q1, s, fw["action"])
        q2n, hq2 = self._q(self.q2, s, fw["action"])
        qmin = np.minimum(q1n, q2n)
        mu, ls = outs[0], fw["log_std"]
        ploss = np.mean(alpha * fw["log_prob"] - qmin, dtype=F32)
        out["policy_loss"] = F32(ploss + F32(self.w_mu) * np.mean(mu ** 2, dtype=F32) + F32(self.w_std) * np.mean(ls ** 2, dtype=F32))
        w1 = np.where(q1n < q2n, F32(1), np.where(q1n == q2n, F32(0.5), F32(0)))
        _, dx1 = mlp.backward(self.q1, hq1, [(-w1 * inv).astype(F32)], self.o + self.a, self.hidden, 1)
        _, dx2 = mlp.backward(self.q2, hq2, [(-(F32(1) - w1) * inv).astype(F32)], self.o + self.a, self.hidden, 1)
        g_action = (dx1[:, self.o:] + dx2[:, self.o:]).astype(F32)
        inv_ba = inv / F32(self.a)
        d_mu, d_ls = tg.head_backward(fw, eps, outs[1], g_action, np.full((B, 1), alpha * inv, dtype=F32),
                                      g_mu_extra=F32(2.0 * self.w_mu) * mu * inv_ba, g_ls_extra=F32(2.0 * self.w_std) * ls * inv_ba)
        gpi, _ = mlp.backward(self.pi, hs_pi, [d_mu, d_ls], self.o, self.hidden, self.a, n_heads=2, need_dx=False)
        optim.adam_step(self.pi, gpi, self.opt["pi"], self.policy_lr, self.beta_1)
        out.update(pi_grad=gpi)
        optim.polyak(self.tvf, self.vf, self.tau)                                 # sac.py:242-243
        return out
