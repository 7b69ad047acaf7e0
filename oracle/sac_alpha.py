"""SAC with twin Q + auto-tuned alpha: restatement of
rlkit/torch/algorithms/sac/sac_alpha.py:21-76 (ctor) and :78-181 (train_step), :245-247 (targets).
numpy fp32 (+ float64 scalar log_alpha, sac_alpha.py:51-53).  Test infrastructure.
"""
import numpy as np

from . import mlp, optim
from . import tanh_gaussian as tg

F32 = np.float32


class SacAlphaOracle:
    """State + one train_step.  Networks are flat fp32 vectors in the oracle/mlp.py layout."""

    def __init__(self, obs_dim, act_dim, hidden, pi, q1, q2, reward_scale=1.0, discount=0.99,
                 policy_lr=1e-3, qf_lr=1e-3, alpha_lr=3e-4, soft_target_tau=1e-2, alpha=0.2,
                 train_alpha=True, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3,
                 beta_1=0.9, target_entropy=None):
        self.o, self.a, self.hidden = obs_dim, act_dim, list(hidden)
        self.pi, self.q1, self.q2 = pi.copy(), q1.copy(), q2.copy()
        self.tq1, self.tq2 = q1.copy(), q2.copy()  # sac_alpha.py:60-61 (qf.copy())
        self.reward_scale, self.discount = reward_scale, discount
        self.policy_lr, self.qf_lr, self.alpha_lr = policy_lr, qf_lr, alpha_lr
        self.tau, self.beta_1 = soft_target_tau, beta_1
        self.train_alpha = train_alpha
        self.w_mu, self.w_std = policy_mean_reg_weight, policy_std_reg_weight
        self.log_alpha = np.array([np.log(alpha)], dtype=np.float64)  # sac_alpha.py:51-53
        # default target entropy = -dim(A)/2 (sac_alpha.py:56-58)
        self.target_entropy = -act_dim / 2.0 if target_entropy is None else target_entropy
        self.opt_pi = optim.AdamState(pi.size)
        self.opt_q1 = optim.AdamState(q1.size)
        self.opt_q2 = optim.AdamState(q2.size)
        self.opt_alpha = optim.AdamState(1, np.float64)

    @property
    def alpha(self):
        return F32(np.exp(self.log_alpha[0]))  # float64 0-dim tensor used at fp32 in tensor ops

    def _q(self, flat, s, a):
        x = np.concatenate([s, a], axis=1).astype(F32)  # FlattenMlp, networks.py:108-115
        outs, hs = mlp.forward(flat, x, self.o + self.a, self.hidden, 1)
        return outs[0], hs

    def _pi(self, s, eps):
        outs, hs = mlp.forward(self.pi, s, self.o, self.hidden, self.a, n_heads=2)
        fw = tg.head_forward(outs[0], outs[1], eps)
        return outs, hs, fw

    def train_step(self, batch, eps_next, eps_cur):
        """batch: dict observations/actions/rewards/terminals/next_observations (fp32 [B,.]).
        eps_next / eps_cur: the two N(0,1) draws [B,a] (sac_alpha.py:102 and :142).
        Returns dict of intermediates (all fp32).  == the four phases below run back to back."""
        out = self.critic_backward(batch, eps_next)
        self.critic_update()
        out.update(self.actor_backward(eps_cur))
        out.update(self.actor_update())
        return out

    # ---- the same step as four phases (the split-run / all-reduce seam of SURVEY §8e): gradients of the
    #      mean losses are scaled by 1/(B*grad_world) so that summing them over ranks gives the global mean
    grad_world = 1

    def critic_backward(self, batch, eps_next):
        B = batch["observations"].shape[0]
        s = batch["observations"].astype(F32)
        a = batch["actions"].astype(F32)
        s2 = batch["next_observations"].astype(F32)
        r = (F32(self.reward_scale) * batch["rewards"].astype(F32)).reshape(B, 1)
        d = batch["terminals"].astype(F32).reshape(B, 1)
        alpha = self.alpha
        self._s, self._B, self._alpha_used = s, B, alpha
        inv = F32(1.0) / F32(B * self.grad_world)
        out = {}
        # ---- critic (sac_alpha.py:96-133)
        q1, hs1 = self._q(self.q1, s, a)
        q2, hs2 = self._q(self.q2, s, a)
        _, _, fw_n = self._pi(s2, eps_next)
        tq1, _ = self._q(self.tq1, s2, fw_n["action"])
        tq2, _ = self._q(self.tq2, s2, fw_n["action"])
        y = r + (F32(1) - d) * F32(self.discount) * (np.minimum(tq1, tq2) - alpha * fw_n["log_prob"])
        y = y.astype(F32)
        out.update(q1_pred=q1, q2_pred=q2, q_target=y, next_log_pi=fw_n["log_prob"],
                   next_actions=fw_n["action"])
        out["qf1_loss"] = F32(0.5) * np.mean((q1 - y) ** 2, dtype=F32)
        out["qf2_loss"] = F32(0.5) * np.mean((q2 - y) ** 2, dtype=F32)
        g1, _ = mlp.backward(self.q1, hs1, [(q1 - y) * inv], self.o + self.a, self.hidden, 1, need_dx=False)
        g2, _ = mlp.backward(self.q2, hs2, [(q2 - y) * inv], self.o + self.a, self.hidden, 1, need_dx=False)
        out.update(q1_grad=g1, q2_grad=g2)
        self.g_critic = np.concatenate([g1, g2])
        return out

    def critic_update(self):
        nq = self.q1.size
        optim.adam_step(self.q1, self.g_critic[:nq], self.opt_q1, self.qf_lr, self.beta_1)
        optim.adam_step(self.q2, self.g_critic[nq:], self.opt_q2, self.qf_lr, self.beta_1)

    def actor_backward(self, eps_cur):
        # ---- actor, evaluated with the JUST-UPDATED critics and a fresh eps (sac_alpha.py:142-155)
        s, B, alpha = self._s, self._B, self._alpha_used
        inv = F32(1.0) / F32(B * self.grad_world)
        out = {}
        outs, hs_pi, fw = self._pi(s, eps_cur)
        q1n, hq1 = self._q(self.q1, s, fw["action"])
        q2n, hq2 = self._q(self.q2, s, fw["action"])
        qmin = np.minimum(q1n, q2n)
        mu, ls = outs[0], fw["log_std"]
        ploss = np.mean(alpha * fw["log_prob"] - qmin, dtype=F32)
        ploss = ploss + F32(self.w_mu) * np.mean(mu ** 2, dtype=F32) + F32(self.w_std) * np.mean(ls ** 2, dtype=F32)
        out.update(policy_loss=F32(ploss), log_pi=fw["log_prob"], new_actions=fw["action"],
                   policy_mean=mu, policy_log_std=ls, q_new_actions=qmin)
        # d(-mean qmin)/d q_i : torch.minimum splits ties evenly
        w1 = np.where(q1n < q2n, F32(1), np.where(q1n == q2n, F32(0.5), F32(0)))
        gq1 = (-w1 * inv).astype(F32)
        gq2 = (-(F32(1) - w1) * inv).astype(F32)
        _, dx1 = mlp.backward(self.q1, hq1, [gq1], self.o + self.a, self.hidden, 1)
        _, dx2 = mlp.backward(self.q2, hq2, [gq2], self.o + self.a, self.hidden, 1)
        g_action = (dx1[:, self.o:] + dx2[:, self.o:]).astype(F32)
        g_logp = np.full((B, 1), alpha * inv, dtype=F32)
        inv_ba = inv / F32(self.a)
        d_mu, d_ls_raw = tg.head_backward(
            fw, eps_cur, outs[1], g_action, g_logp,
            g_mu_extra=F32(2.0 * self.w_mu) * mu * inv_ba,
            g_ls_extra=F32(2.0 * self.w_std) * ls * inv_ba)
        gpi, _ = mlp.backward(self.pi, hs_pi, [d_mu, d_ls_raw], self.o, self.hidden, self.a,
                              n_heads=2, need_dx=False)
        out.update(pi_grad=gpi, g_action=g_action)
        # ---- alpha (sac_alpha.py:160-166): loss in fp32
        lp = (fw["log_prob"] + F32(self.target_entropy)).astype(F32)
        out["alpha_loss"] = F32(-np.mean(F32(self.log_alpha[0]) * lp, dtype=F32))
        g_alpha = F32(-np.sum(lp, dtype=F32) * inv)
        self.g_actor = np.concatenate([gpi, np.array([g_alpha, 0, 0, 0], dtype=F32)])  # policy grads | alpha slot
        return out

    def actor_update(self):
        out = {}
        npi = self.pi.size
        optim.adam_step(self.pi, self.g_actor[:npi], self.opt_pi, self.policy_lr, self.beta_1)
        if self.train_alpha:  # parameter + Adam in float64 (sac_alpha.py:51-53,160-166)
            g_alpha = np.array([np.float64(self.g_actor[npi])])
            optim.adam_step(self.log_alpha, g_alpha, self.opt_alpha, self.alpha_lr, self.beta_1)
        out["log_alpha"] = self.log_alpha.copy()
        # ---- targets from post-Adam critics (sac_alpha.py:181,245-247)
        optim.polyak(self.tq1, self.q1, self.tau)
        optim.polyak(self.tq2, self.q2, self.tau)
        return out
