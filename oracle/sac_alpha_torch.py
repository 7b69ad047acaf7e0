"""SAC (twin Q, auto alpha) in PyTorch on the CPU: the SECOND restatement of rlkit/torch/algorithms/sac/sac_alpha.py:21-181, in the
reference's own idiom — Linear layers as tensors, autograd for the backward passes, torch.optim.Adam, the python-loop soft update
(pytorch_util.py:10-12) — where oracle/sac_alpha.py does the same arithmetic by hand in numpy.  It exists to be TIMED: SURVEY.md
§8d asks for "the build's own fp32 PyTorch-CPU implementation ... at 1 and all threads" beside the GPU number, because that is the
shape of the reference's CPU path (~1.8k ATen calls per step).  Pinned by the same reference-generated vectors (g4) as the numpy
oracle (tests/test_oracle_golden.py).  Test / bench infrastructure: nothing under ilswiss_amd/ imports it.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import mlp as omlp

LOG_SIG_MIN, LOG_SIG_MAX, EPS = -20.0, 2.0, 1e-6


def _params(flat, in_dim, hidden, out_dim, n_heads=1):
    return [torch.tensor(np.array(x), dtype=torch.float32, requires_grad=True)
            for W, b in omlp.unpack(np.asarray(flat, np.float32), in_dim, hidden, out_dim, n_heads) for x in (W, b)]


def _mlp(ps, x, n_hidden, n_heads=1):   # networks.py:85-101
    h = x
    for l in range(n_hidden):
        h = F.relu(F.linear(h, ps[2 * l], ps[2 * l + 1]))
    return [F.linear(h, ps[2 * (n_hidden + k)], ps[2 * (n_hidden + k) + 1]) for k in range(n_heads)]


def _flat(ps):
    return np.concatenate([p.detach().numpy().ravel() for p in ps]).astype(np.float32)


class SacAlphaTorch:
    def __init__(self, obs_dim, act_dim, hidden, pi, q1, q2, reward_scale=1.0, discount=0.99, policy_lr=1e-3, qf_lr=1e-3,
                 alpha_lr=3e-4, soft_target_tau=1e-2, alpha=0.2, train_alpha=True, policy_mean_reg_weight=1e-3,
                 policy_std_reg_weight=1e-3, beta_1=0.9, target_entropy=None):
        self.o, self.a, self.nh = obs_dim, act_dim, len(hidden)
        self.pi_p = _params(pi, obs_dim, hidden, act_dim, 2)
        self.q1_p, self.q2_p = _params(q1, obs_dim + act_dim, hidden, 1), _params(q2, obs_dim + act_dim, hidden, 1)
        self.tq1_p = [p.detach().clone() for p in self.q1_p]      # qf.copy() (sac_alpha.py:60-61)
        self.tq2_p = [p.detach().clone() for p in self.q2_p]
        self.reward_scale, self.discount, self.tau = reward_scale, discount, soft_target_tau
        self.w_mu, self.w_std, self.train_alpha = policy_mean_reg_weight, policy_std_reg_weight, train_alpha
        self.log_alpha = torch.tensor(np.log(alpha), requires_grad=True)     # float64 0-dim (sac_alpha.py:51-53)
        self.alpha = self.log_alpha.detach().exp()
        self.target_entropy = -act_dim / 2.0 if target_entropy is None else target_entropy   # :56-58
        mk = lambda ps, lr: torch.optim.Adam(ps, lr=lr, betas=(beta_1, 0.999))   # noqa: E731  (:65-76)
        self.opt_pi, self.opt_q1, self.opt_q2 = mk(self.pi_p, policy_lr), mk(self.q1_p, qf_lr), mk(self.q2_p, qf_lr)
        self.opt_alpha = mk([self.log_alpha], alpha_lr)

    def _policy(self, obs, eps):   # policies.py:248-307 + distributions.py:23-28,43-50,74-97
        mu, ls_raw = _mlp(self.pi_p, obs, self.nh, 2)
        ls = torch.clamp(ls_raw, LOG_SIG_MIN, LOG_SIG_MAX)
        z = mu + torch.exp(ls) * eps
        act = torch.tanh(z)
        lp = -0.5 * torch.sum((mu - z) ** 2 / torch.exp(2.0 * ls), 1, keepdim=True)
        lp = lp - (torch.sum(ls, 1, keepdim=True) + 0.5 * math.log(2.0 * math.pi))
        lp = lp - torch.sum(torch.log(1.0 - act * act + EPS), 1, keepdim=True)
        return act, mu, ls, lp

    def train_step(self, batch, eps_next, eps_cur):   # sac_alpha.py:78-181
        t = lambda x: torch.as_tensor(np.asarray(x, np.float32))   # noqa: E731  (np_to_pytorch_batch, core.py:124-143)
        obs, act, nobs = t(batch["observations"]), t(batch["actions"]), t(batch["next_observations"])
        B = obs.shape[0]
        rew = self.reward_scale * t(batch["rewards"]).reshape(B, 1)
        term = t(batch["terminals"]).reshape(B, 1)
        out = {}
        # ---- critics (:96-133)
        self.opt_q1.zero_grad(); self.opt_q2.zero_grad()
        q1 = _mlp(self.q1_p, torch.cat([obs, act], 1), self.nh)[0]
        q2 = _mlp(self.q2_p, torch.cat([obs, act], 1), self.nh)[0]
        with torch.no_grad():
            na, _, _, nlp = self._policy(nobs, t(eps_next))
            xq = torch.cat([nobs, na], 1)
            tq = torch.min(_mlp(self.tq1_p, xq, self.nh)[0], _mlp(self.tq2_p, xq, self.nh)[0])
            y = rew + (1.0 - term) * self.discount * (tq - self.alpha.float() * nlp)
        l1, l2 = 0.5 * torch.mean((q1 - y) ** 2), 0.5 * torch.mean((q2 - y) ** 2)
        l1.backward(); l2.backward()
        self.opt_q1.step(); self.opt_q2.step()
        out.update(qf1_loss=l1.item(), qf2_loss=l2.item(), q1_pred=q1.detach().numpy())
        # ---- actor with the updated critics (:142-155)
        a_new, mu, ls, lp = self._policy(obs, t(eps_cur))
        xq = torch.cat([obs, a_new], 1)
        qn = torch.min(_mlp(self.q1_p, xq, self.nh)[0], _mlp(self.q2_p, xq, self.nh)[0])
        pl = torch.mean(self.alpha.float() * lp - qn) + self.w_mu * torch.mean(mu ** 2) + self.w_std * torch.mean(ls ** 2)
        self.opt_pi.zero_grad()
        pl.backward()
        self.opt_pi.step()
        out.update(policy_loss=pl.item(), log_pi=lp.detach().numpy())
        # ---- alpha (:160-166)
        al = -(self.log_alpha * (lp.detach() + self.target_entropy)).mean()
        out["alpha_loss"] = al.item()
        if self.train_alpha:
            self.opt_alpha.zero_grad()
            al.backward()
            self.opt_alpha.step()
            self.alpha = self.log_alpha.detach().exp()
        # ---- targets (:181, pytorch_util.py:10-12)
        with torch.no_grad():
            for tp, p in zip(self.tq1_p + self.tq2_p, self.q1_p + self.q2_p):
                tp.copy_(tp * (1.0 - self.tau) + p * self.tau)
        return out

    def flat(self, name):
        return _flat(dict(pi=self.pi_p, q1=self.q1_p, q2=self.q2_p, tq1=self.tq1_p, tq2=self.tq2_p)[name])
