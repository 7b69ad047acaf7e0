"""SAC with a state-value function over libilsx: the reference's `rlkit/torch/algorithms/sac/sac.py:13-262`
(`SoftActorCritic` of sac_exp_script.py).  Named SoftActorCriticV here because `ilswiss_amd.sac.SoftActorCritic` is
the twin-Q / learned-alpha trainer (sac_alpha.py) that the BASELINE configs use.  Constructor kwargs are the YAML
`sac_params` keys; unknown keys are swallowed like the reference's **kwargs.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .sac import Trainer, check_swallowed_kwargs
from .td3 import _batch_ptrs, _stat_block


class SoftActorCriticV(Trainer):
    WHICH = dict(qf1=0, qf2=1, vf=2, policy=3, target_vf=6, q1=0, q2=1, pi=3, tvf=6)

    def __init__(self, policy, qf1, qf2, vf, reward_scale=1.0, discount=0.99, alpha=1.0, policy_lr=1e-3, qf_lr=1e-3,
                 vf_lr=1e-3, soft_target_tau=1e-2, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9,
                 max_batch=1024, **kwargs):
        check_swallowed_kwargs(kwargs, "SoftActorCriticV")
        self.policy, self.qf1, self.qf2, self.vf, self.ctx = policy, qf1, qf2, vf, policy.ctx
        self.reward_scale = reward_scale
        cfg = _lib.SacvCfg(reward_scale, discount, alpha, policy_lr, qf_lr, vf_lr, soft_target_tau, policy_mean_reg_weight,
                           policy_std_reg_weight, beta_1, int(max_batch))
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_sacv_create(self.ctx.h, C.byref(cfg), policy.h, qf1.h, qf2.h, vf.h, C.byref(self.h)))
        self.eval_statistics = None
        self._stats = _lib.SacvStats()

    def train_step(self, batch, eps=None):
        keep = []
        B, p, dev = _batch_ptrs(self.ctx, batch, keep)
        e = dev(eps) if eps is not None else None
        want = self.eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_sacv_train_step(self.h, *p, B, e, C.byref(self._stats) if want else None))
        if want:
            self._fill_stats()
        else:
            self.ctx.sync()

    def train_from_replay(self, replay_buffer, n_steps, batch_size):
        want = self.eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_sacv_train_from_replay(self.h, replay_buffer.h, int(n_steps), int(batch_size),
                                                            C.byref(self._stats) if want else None))
        if want:
            self._fill_stats()

    def _fill_stats(self):  # sac.py:181-240
        s, st = self._stats, OrderedDict()
        st["Reward Scale"] = self.reward_scale
        st["QF1 Loss"], st["QF2 Loss"], st["VF Loss"], st["Policy Loss"] = s.qf1_loss, s.qf2_loss, s.vf_loss, s.policy_loss
        for name, vals in (("Q1 Predictions", s.q1_pred), ("Q2 Predictions", s.q2_pred), ("V Predictions", s.v_pred),
                           ("Log Pis", s.log_pi), ("Policy mu", s.policy_mu), ("Policy log std", s.policy_log_std)):
            _stat_block(st, name, vals)
        self.eval_statistics = st

    def get_eval_statistics(self):
        return self.eval_statistics

    def end_epoch(self):
        self.eval_statistics = None

    @property
    def networks(self):
        return [self.policy, self.qf1, self.qf2, self.vf]

    def _n(self, w):
        return {0: self.qf1, 1: self.qf2, 2: self.vf, 3: self.policy}[w % 4].num_params

    def get_flat_params(self, name):
        w = self.WHICH[name]
        out = np.empty(self._n(w), np.float32)
        _lib.check(self.ctx.lib.ilsx_sacv_get_params(self.h, w, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set_flat_params(self, name, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        _lib.check(self.ctx.lib.ilsx_sacv_set_params(self.h, self.WHICH[name], flat.ctypes.data_as(C.c_void_p), flat.size))

    _OPT = (("qf1", 0), ("qf2", 1), ("vf", 2), ("policy", 3))

    def get_snapshot(self):  # sac.py:245-257, as plain arrays (+ the four optimisers' Adam state)
        from .snapshot import get_opt
        snap = {k: self.get_flat_params(k) for k in ("qf1", "qf2", "policy", "vf", "target_vf")}
        for k, w in self._OPT:
            snap[k + "_optimizer"] = get_opt(self.ctx.lib, "sacv", self.h, snap[k].size, w)
        return snap

    def load_snapshot(self, snap):  # sac.py:259-270
        from .snapshot import set_opt
        for k in ("qf1", "qf2", "policy", "vf", "target_vf"):
            self.set_flat_params(k, snap[k])
        for k, w in self._OPT:
            if k + "_optimizer" in snap:
                set_opt(self.ctx.lib, "sacv", self.h, snap[k + "_optimizer"], w)
