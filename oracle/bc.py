"""Behaviour cloning: restatement of rlkit/torch/algorithms/bc/bc.py:14-41 (ctor: Adam(lr, betas=(momentum, 0.999)) over the
policy) and :81-106 (_do_update_step): mode "MLE" = -mean(policy.get_log_prob(obs, acts)) (policies.py:329-345); mode "MSE"
= mean_rows(sum_j (policy(obs)[0] - acts)^2) where policy(obs)[0] is a SAMPLED tanh-Gaussian action (forward with
deterministic=False, policies.py:248-307), so the gradient also reaches log_std through sigma*eps.  numpy fp32.
Test infrastructure."""
import numpy as np

from . import mlp, optim
from . import tanh_gaussian as tg

F32 = np.float32


class BCOracle:
    def __init__(self, obs_dim, act_dim, hidden, pi, mode="MLE", lr=1e-3, momentum=0.0):
        assert mode in ("MLE", "MSE")
        self.o, self.a, self.hidden, self.mode, self.lr, self.momentum = obs_dim, act_dim, list(hidden), mode, lr, momentum
        self.pi = pi.copy()
        self.opt = optim.AdamState(pi.size)

    def update(self, obs, acts, eps=None):
        B = obs.shape[0]
        obs, acts = obs.astype(F32), acts.astype(F32)
        outs, hs = mlp.forward(self.pi, obs, self.o, self.hidden, self.a, n_heads=2)
        mu, lsr = outs
        ls = np.clip(lsr, F32(tg.LOG_SIG_MIN), F32(tg.LOG_SIG_MAX))
        gate = (lsr >= F32(tg.LOG_SIG_MIN)) & (lsr <= F32(tg.LOG_SIG_MAX))
        inv = F32(1.0) / F32(B)
        if self.mode == "MLE":
            lp = tg.log_prob_of_action(mu, lsr, acts)
            stat = F32(np.mean(lp, dtype=F32))                     # "Log-Likelihood" (bc.py:94)
            z = F32(0.5) * (np.log(F32(1) + acts + F32(tg.EPS)) - np.log(F32(1) - acts + F32(tg.EPS)))
            var = np.exp(F32(2) * ls)
            d_mu = ((mu - z) / var * inv).astype(F32)              # d(-mean lp)/d mu
            d_ls = (-((mu - z) ** 2 / var - F32(1)) * inv * gate).astype(F32)
        else:
            fw = tg.head_forward(mu, lsr, eps)
            pred = fw["action"]
            stat = F32(np.mean(np.sum((pred - acts) ** 2, axis=1), dtype=F32))   # "MSE" (bc.py:101)
            dz = (F32(2) * (pred - acts) * inv * (F32(1) - pred * pred)).astype(F32)
            d_mu = dz
            d_ls = (dz * fw["std"] * eps.astype(F32) * gate).astype(F32)
        g, _ = mlp.backward(self.pi, hs, [d_mu, d_ls], self.o, self.hidden, self.a, n_heads=2, need_dx=False)
        optim.adam_step(self.pi, g, self.opt, self.lr, self.momentum)
        return dict(stat=stat, grad=g)
