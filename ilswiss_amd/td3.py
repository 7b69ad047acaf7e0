"""TD3 over libilsx: the reference's `TD3` trainer (rlkit/torch/algorithms/td3/td3.py:13-190) and its policy
`MlpGaussianNoisePolicy` (rlkit/torch/common/policies.py:130-188).  Constructor kwargs are the YAML `td3_params`
keys (exp_specs/td3/td3_hopper.yaml:39-45); unknown keys are swallowed like the reference's **kwargs.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .device import as_dev
from .networks import Mlp
from .sac import Trainer, check_swallowed_kwargs

_STAT4 = ("Mean", "Std", "Max", "Min")


def _stat_block(st, name, vals):
    for k, v in zip(_STAT4, vals):
        st[f"{name} {k}"] = float(v)


def _batch_ptrs(ctx, batch, keep):
    def dev(x):
        k, p = as_dev(ctx, x)
        keep.append(k)
        return p
    B = int(batch["observations"].shape[0])
    flat = lambda v: v.reshape(B) if hasattr(v, "reshape") else v  # noqa: E731
    return B, [dev(batch["observations"]), dev(batch["actions"]), dev(flat(batch["rewards"])),
               dev(flat(batch["terminals"])), dev(batch["next_observations"])], dev


class MlpGaussianNoisePolicy(Mlp):
    """policies.py:130-188: relu Mlp, `max_act * output_activation(last_fc)`, plus clip(policy_noise * N(0,1), +-policy_noise_clip)
    unless deterministic.  `output_activation` is Mlp's keyword (networks.py:31): identity by default, as in the reference; the run
    script passes tanh (td3_exp_script.py:71-78).  Accepted: "tanh" / "identity" or a callable of that name (torch.tanh, ptu.identity)."""

    def __init__(self, hidden_sizes, obs_dim, action_dim, init_w=1e-3, policy_noise=0.1, policy_noise_clip=0.5, max_act=1.0,
                 output_activation="identity", **kwargs):
        name = getattr(output_activation, "__name__", output_activation)
        if name not in ("tanh", "identity"):
            raise NotImplementedError(f"output_activation={name!r}: libilsx implements tanh and identity")
        super().__init__(hidden_sizes, input_size=obs_dim, output_size=action_dim, init_w=init_w, **kwargs)
        self.obs_dim, self.action_dim = int(obs_dim), int(action_dim)
        self.noise, self.noise_clip, self.max_act = float(policy_noise), float(policy_noise_clip), float(max_act)
        _lib.check(self.ctx.lib.ilsx_net_set_noise_policy(self.h, self.noise, self.noise_clip, self.max_act))
        self.output_activation = name
        _lib.check(self.ctx.lib.ilsx_net_set_output_linear(self.h, int(name == "identity")))

    def get_actions(self, obs_np, deterministic=False):  # policies.py:163-164
        obs = np.ascontiguousarray(obs_np, np.float32)
        keep, p = as_dev(self.ctx, obs)
        act = self.ctx.empty((obs.shape[0], self.action_dim))
        _lib.check(self.ctx.lib.ilsx_policy_act(self.h, p, obs.shape[0], int(bool(deterministic)), None, act.ptr, None))
        return act.numpy()

    def get_action(self, obs_np, deterministic=False):  # policies.py:154-161
        return self.get_actions(np.asarray(obs_np)[None], deterministic)[0], {}

    def set_num_steps_total(self, t):
        pass


class TD3(Trainer):
    WHICH = dict(qf1=0, qf2=1, policy=2, target_qf1=3, target_qf2=4, target_policy=5,
                 q1=0, q2=1, pi=2, tq1=3, tq2=4, tpi=5)

    def __init__(self, policy, qf1, qf2, reward_scale=1.0, discount=0.99, target_policy_noise=0.2,
                 target_policy_noise_clip=0.5, policy_lr=1e-3, qf_lr=1e-3, policy_and_target_update_period=2,
                 soft_target_tau=0.005, max_batch=1024, her=False, clip_return_l=0.0, clip_return_r=0.0, **kwargs):
        # target_policy_noise* are accepted and, like in the reference (td3.py:46-47 store them, nothing reads them),
        # unused: the target policy is policy.copy() and adds the policy module's own noise.
        check_swallowed_kwargs(kwargs, "TD3")
        self.policy, self.qf1, self.qf2, self.ctx = policy, qf1, qf2, policy.ctx
        self.reward_scale = reward_scale
        cfg = _lib.Td3Cfg(reward_scale, discount, policy_lr, qf_lr, int(policy_and_target_update_period), soft_target_tau,
                          policy.noise, policy.noise_clip, policy.max_act, int(max_batch), int(bool(her)), float(clip_return_l),
                          float(clip_return_r))   # her: rlkit/torch/algorithms/her/td3.py (ilswiss_amd/her.py:TD3)
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_td3_create(self.ctx.h, C.byref(cfg), policy.h, qf1.h, qf2.h, C.byref(self.h)))
        self.eval_statistics = None
        self._stats = _lib.Td3Stats()

    def train_step(self, batch, eps_target=None):
        keep = []
        B, p, dev = _batch_ptrs(self.ctx, batch, keep)
        e = dev(eps_target) if eps_target is not None else None
        want = self.eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_td3_train_step(self.h, *p, B, e, C.byref(self._stats) if want else None))
        if want:
            self._fill_stats()
        else:
            self.ctx.sync()

    def train_from_replay(self, replay_buffer, n_steps, batch_size):
        want = self.eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_td3_train_from_replay(self.h, replay_buffer.h, int(n_steps), int(batch_size),
                                                           C.byref(self._stats) if want else None))
        if want:
            self._fill_stats()

    def _fill_stats(self):  # td3.py:131-176
        s, st = self._stats, OrderedDict()
        st["QF1 Loss"], st["QF2 Loss"], st["Policy Loss"] = s.qf1_loss, s.qf2_loss, s.policy_loss
        for name, vals in (("Q1 Predictions", s.q1_pred), ("Q2 Predictions", s.q2_pred), ("Q Targets", s.q_target),
                           ("Bellman Errors 1", s.bellman1), ("Bellman Errors 2", s.bellman2),
                           ("Policy Action", s.policy_action)):
            _stat_block(st, name, vals)
        self.eval_statistics = st

    def get_eval_statistics(self):
        return self.eval_statistics

    def end_epoch(self):
        self.eval_statistics = None

    @property
    def networks(self):
        return [self.policy, self.qf1, self.qf2]

    def get_flat_params(self, name):
        w = self.WHICH[name]
        out = np.empty(self.policy.num_params if w % 3 == 2 else self.qf1.num_params, np.float32)
        _lib.check(self.ctx.lib.ilsx_td3_get_params(self.h, w, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set_flat_params(self, name, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        _lib.check(self.ctx.lib.ilsx_td3_set_params(self.h, self.WHICH[name], flat.ctypes.data_as(C.c_void_p), flat.size))

    def get_snapshot(self):  # td3.py:185-196, as plain arrays (+ the three optimisers' Adam state)
        from .snapshot import get_opt
        snap = {k: self.get_flat_params(k) for k in ("qf1", "qf2", "policy", "target_policy", "target_qf1", "target_qf2")}
        for k, w in (("qf1", 0), ("qf2", 1), ("policy", 2)):
            snap[k + "_optimizer"] = get_opt(self.ctx.lib, "td3", self.h, snap[k].size, w)
        return snap

    def load_snapshot(self, snap):  # td3.py:198-206
        from .snapshot import set_opt
        for k in ("qf1", "qf2", "policy", "target_policy", "target_qf1", "target_qf2"):
            self.set_flat_params(k, snap[k])
        for k, w in (("qf1", 0), ("qf2", 1), ("policy", 2)):
            if k + "_optimizer" in snap:
                set_opt(self.ctx.lib, "td3", self.h, snap[k + "_optimizer"], w)
