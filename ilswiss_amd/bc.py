"""Behaviour cloning over libilsx: the update of the reference's `BC` algorithm (rlkit/torch/algorithms/bc/bc.py:14-106).
`BC(mode, policy, expert_replay_buffer, batch_size, lr, momentum, num_updates_per_train_call)` keeps the reference's
kwargs; `train_from_replay` is `_do_training` (:77-79) with the expert batches drawn on the device."""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .device import as_dev
from .sac import Trainer, check_swallowed_kwargs

_MODES = dict(MLE=0, MSE=1)


class BC(Trainer):
    def __init__(self, mode, policy, expert_replay_buffer=None, num_updates_per_train_call=1, batch_size=1024, lr=1e-3,
                 momentum=0.0, **kwargs):
        assert mode in _MODES, "Invalid mode!"           # bc.py:26
        if kwargs.get("wrap_absorbing"):
            raise NotImplementedError()                  # bc.py:27-28
        check_swallowed_kwargs(kwargs, "BC")
        self.mode, self.policy, self.ctx = mode, policy, policy.ctx
        self.expert_replay_buffer, self.batch_size = expert_replay_buffer, int(batch_size)
        self.num_updates_per_train_call = int(num_updates_per_train_call)
        cfg = _lib.BcCfg(_MODES[mode], lr, momentum, self.batch_size)
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_bc_create(self.ctx.h, C.byref(cfg), policy.h, C.byref(self.h)))
        self.eval_statistics = None

    def _record(self, stat):
        self.eval_statistics = OrderedDict({"Log-Likelihood" if self.mode == "MLE" else "MSE": stat})

    def train_step(self, batch, eps=None):
        """_do_update_step (bc.py:81-106) on an explicit batch with keys observations / actions."""
        ko, po = as_dev(self.ctx, np.ascontiguousarray(batch["observations"], np.float32))
        ka, pa = as_dev(self.ctx, np.ascontiguousarray(batch["actions"], np.float32))
        ke, pe = as_dev(self.ctx, np.ascontiguousarray(eps, np.float32)) if eps is not None else (None, None)
        want, st = self.eval_statistics is None, C.c_float()
        _lib.check(self.ctx.lib.ilsx_bc_train_step(self.h, po, pa, int(np.shape(batch["observations"])[0]), pe,
                                                   C.byref(st) if want else None))
        if want:
            self._record(st.value)
        else:
            self.ctx.sync()

    def train_from_replay(self, replay_buffer=None, n_updates=None, batch_size=None):
        rb = self.expert_replay_buffer if self.expert_replay_buffer is not None else replay_buffer   # use_expert_buffer=True
        want, st = self.eval_statistics is None, C.c_float()
        _lib.check(self.ctx.lib.ilsx_bc_train_from_replay(self.h, rb.h, int(n_updates or self.num_updates_per_train_call),
                                                          int(batch_size or self.batch_size), C.byref(st) if want else None))
        if want:
            self._record(st.value)

    def get_eval_statistics(self):
        return self.eval_statistics

    def end_epoch(self):
        self.eval_statistics = None

    @property
    def networks(self):
        return [self.policy]

    def get_snapshot(self):   # bc.py:108-113 (+ the optimiser's Adam state)
        from .snapshot import get_opt
        flat = self.policy.get_flat_params()
        return dict(policy=flat, optimizer=get_opt(self.ctx.lib, "bc", self.h, flat.size))

    def load_snapshot(self, snap):
        from .snapshot import set_opt
        self.policy.set_flat_params(snap["policy"])
        if "optimizer" in snap:
            set_opt(self.ctx.lib, "bc", self.h, snap["optimizer"])


class DAgger(BC):
    """rlkit/torch/algorithms/dagger/dagger.py:4-82: BC whose batches come from the policy's own rollouts relabelled with
    the expert's actions.  The expert demonstrations are copied into the replay buffer at construction (:27-35); the first
    train call spends `num_initial_train_steps` updates on the expert buffer alone (:37-40); the sampling loop stores the
    expert's action for every visited observation (`HipVectorEnv.rollout_step(label_policy=...)`, :45-71).
    `unscale_for_expert` concerns ScaledEnv wrappers, which libilsx does not have (observations are raw)."""

    def __init__(self, expert_policy, mode, policy, expert_replay_buffer, replay_buffer, num_initial_train_steps=100, **kwargs):
        kwargs.pop("unscale_for_expert", None)
        super().__init__(mode, policy, expert_replay_buffer=expert_replay_buffer, **kwargs)
        self.expert_policy, self.replay_buffer = expert_policy, replay_buffer
        self.num_initial_train_steps, self._first_call = int(num_initial_train_steps), True
        n = expert_replay_buffer.num_steps_can_sample()
        b = expert_replay_buffer._gather(np.arange(n))
        replay_buffer.add_rows(b["observations"], b["actions"], b["rewards"], b["terminals"], b["next_observations"])

    def train_from_replay(self, replay_buffer=None, n_updates=None, batch_size=None):
        n = int(n_updates or self.num_updates_per_train_call)
        B = int(batch_size or self.batch_size)
        if self._first_call:          # `epoch == 0` in the reference
            self._first_call = False
            _lib.check(self.ctx.lib.ilsx_bc_train_from_replay(self.h, self.expert_replay_buffer.h, self.num_initial_train_steps, B, None))
        rb = replay_buffer if replay_buffer is not None else self.replay_buffer
        want, st = self.eval_statistics is None, C.c_float()
        _lib.check(self.ctx.lib.ilsx_bc_train_from_replay(self.h, rb.h, n, B, C.byref(st) if want else None))
        if want:
            self._record(st.value)

    @property
    def networks(self):
        return [self.policy, self.expert_policy]
