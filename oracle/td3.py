"""TD3: restatement of rlkit/torch/algorithms/td3/td3.py:21-70 (ctor), :72-124 (train_step), :180-183 (targets)
with the policy of rlkit/torch/common/policies.py:130-188 (MlpGaussianNoisePolicy: relu MLP, tanh or identity output, clipped
Gaussian noise INSIDE the module; the target policy is `policy.copy()`, so it carries the policy's own
policy_noise / policy_noise_clip and is called stochastically, td3.py:84-85 — TD3's `target_policy_noise*`
kwargs are never read).  numpy fp32.  Test infrastructure.
"""
import numpy as np

from . import mlp, optim

F32 = np.float32


class TD3Oracle:
    def __init__(self, obs_dim, act_dim, hidden, pi, q1, q2, reward_scale=1.0, discount=0.99, policy_lr=1e-3, qf_lr=1e-3,
                 policy_and_target_update_period=2, soft_target_tau=0.005, policy_noise=0.2, policy_noise_clip=0.5,
                 max_act=1.0, her=False, clip_return_l=None, clip_return_r=None, output_activation="tanh"):
        # output_activation: "tanh" (what td3_exp_script.py:75 passes) or "identity" (Mlp's default, networks.py:31)
        assert output_activation in ("tanh", "identity")
        self.linear_out = output_activation == "identity"
        self.o, self.a, self.hidden = obs_dim, act_dim, list(hidden)
        self.pi, self.q1, self.q2 = pi.copy(), q1.copy(), q2.copy()
        self.tpi, self.tq1, self.tq2 = pi.copy(), q1.copy(), q2.copy()   # td3.py:53-55
        self.reward_scale, self.discount, self.tau = reward_scale, discount, soft_target_tau
        self.policy_lr, self.qf_lr, self.period = policy_lr, qf_lr, policy_and_target_update_period
        self.noise, self.noise_clip, self.max_act = policy_noise, policy_noise_clip, max_act
        self.opt_pi, self.opt_q1, self.opt_q2 = (optim.AdamState(pi.size), optim.AdamState(q1.size), optim.AdamState(q2.size))
        self.n_steps = 0
        # rlkit/torch/algorithms/her/td3.py:79-86 (goal-conditioned TD3; inputs arrive concatenated observation | desired_goal)
        self.her = bool(her)
        self.clip_l = -1.0 / (1.0 - discount) if clip_return_l is None else clip_return_l
        self.clip_r = 0.0 if clip_return_r is None else clip_return_r

    def policy(self, flat, s, eps=None):
        """policies.py:166-188.  eps = N(0,1) draws [B,a] or None (deterministic).  Returns (action, head pre-activation, hs)."""
        outs, hs = mlp.forward(flat, s, self.o, self.hidden, self.a)
        clean = (F32(self.max_act) * (outs[0] if self.linear_out else np.tanh(outs[0]))).astype(F32)
        act = clean
        if eps is not None:
            act = (clean + np.clip(F32(self.noise) * eps.astype(F32), -F32(self.noise_clip), F32(self.noise_clip))).astype(F32)
        return act, outs[0], hs

    def _q(self, flat, s, a):
        outs, hs = mlp.forward(flat, np.concatenate([s, a], axis=1).astype(F32), self.o + self.a, self.hidden, 1)
        return outs[0], hs

    def train_step(self, batch, eps_target):
        """eps_target: the N(0,1) draws of the target policy's noise (policies.py:182-184)."""
        B = batch["observations"].shape[0]
        s, a, s2 = (batch[k].astype(F32) for k in ("observations", "actions", "next_observations"))
        r = (F32(self.reward_scale) * batch["rewards"].astype(F32)).reshape(B, 1)
        d = batch["terminals"].astype(F32).reshape(B, 1)
        inv = F32(1.0) / F32(B)
        out = {}
        if self.her:   # her/td3.py:104-114: `noisy_next_actions = torch.clamp(noise, min_act, max_act)` — the clamped NOISE, as written
            a2 = np.clip(F32(self.noise) * eps_target.astype(F32), -F32(self.max_act), F32(self.max_act)).astype(F32)
        else:
            a2, _, _ = self.policy(self.tpi, s2, eps_target)                   # NOT re-clipped to [-1,1]
        tq = np.minimum(self._q(self.tq1, s2, a2)[0], self._q(self.tq2, s2, a2)[0])
        if self.her:   # her/td3.py:116-122
            tq = np.clip(tq, F32(self.clip_l), F32(self.clip_r)).astype(F32)
        y = (r + (F32(1) - d) * F32(self.discount) * tq).astype(F32)
        q1, h1 = self._q(self.q1, s, a)
        q2, h2 = self._q(self.q2, s, a)
        out.update(noisy_next_actions=a2, q_target=y, q1_pred=q1, q2_pred=q2,
                   qf1_loss=np.mean((q1 - y) ** 2, dtype=F32), qf2_loss=np.mean((q2 - y) ** 2, dtype=F32))   # no 1/2
        g1, _ = mlp.backward(self.q1, h1, [F32(2) * (q1 - y) * inv], self.o + self.a, self.hidden, 1, need_dx=False)
        g2, _ = mlp.backward(self.q2, h2, [F32(2) * (q2 - y) * inv], self.o + self.a, self.hidden, 1, need_dx=False)
        optim.adam_step(self.q1, g1, self.opt_q1, self.qf_lr)
        optim.adam_step(self.q2, g2, self.opt_q2, self.qf_lr)
        out.update(q1_grad=g1, q2_grad=g2)
        pa, pre, hp = self.policy(self.pi, s)                                   # deterministic=True, td3.py:111
        qn, hq = self._q(self.q1, s, pa)                                        # updated qf1
        out.update(policy_loss=-np.mean(qn, dtype=F32), policy_actions=pa)
        if self.her and self.n_steps % self.period == 0:   # her/td3.py:150-152: + mean(a^2); the statistics of a step without a policy
            out["policy_loss"] = F32(out["policy_loss"] + np.mean(pa * pa, dtype=F32))   # update report -mean(Q) alone (:165-170)
        if self.n_steps % self.period == 0:                                     # td3.py:109-122
            _, dx = mlp.backward(self.q1, hq, [np.full((B, 1), -inv, F32)], self.o + self.a, self.hidden, 1)
            ga = dx[:, self.o:]
            if self.her:
                ga = (ga + F32(2.0) * pa / F32(pa.size)).astype(F32)
            dpre = (ga * F32(self.max_act) * (F32(1) if self.linear_out else (F32(1) - np.tanh(pre) ** 2))).astype(F32)
            gp, _ = mlp.backward(self.pi, hp, [dpre], self.o, self.hidden, self.a, need_dx=False)
            optim.adam_step(self.pi, gp, self.opt_pi, self.policy_lr)
            for t, src in ((self.tpi, self.pi), (self.tq1, self.q1), (self.tq2, self.q2)):
                optim.polyak(t, src, self.tau)
            out.update(pi_grad=gp)
        self.n_steps += 1
        return out

