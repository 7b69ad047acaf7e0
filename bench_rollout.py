"""Rollout leg of bench.py: one vec-env step for N_ENV parallel envs, all on the device:
policy inference (ReparamTanhMultivariateGaussianPolicy.get_actions, policies.py:245-246) ->
env physics -> replay insert (base_algorithm.py:183-263).
"""
import ctypes as C

import numpy as np

from ilswiss_amd import _lib


class Rollout:
    def __init__(self, ctx, policy, replay, n_env, seed=0):
        self.ctx, self.pol, self.rb, self.n = ctx, policy, replay, n_env
        rng = np.random.default_rng(seed)
        o, a = policy.obs_dim, policy.action_dim
        self.obs = ctx.from_numpy(rng.standard_normal((n_env, o), dtype=np.float32))
        self.nobs = ctx.from_numpy(rng.standard_normal((n_env, o), dtype=np.float32))
        self.rew = ctx.from_numpy(rng.standard_normal(n_env, dtype=np.float32))
        self.done = ctx.from_numpy(np.zeros(n_env, np.uint8), np.uint8)
        self.act = ctx.empty((n_env, a))

    def describe(self):
        return "PLACEHOLDER: policy inference + replay insert only, NO physics kernel yet"

    def vec_step(self):
        lib, ctx = self.ctx.lib, self.ctx
        _lib.check(lib.ilsx_policy_act(self.pol.h, self.obs.ptr, self.n, 0, None, self.act.ptr, None))
        _lib.check(lib.ilsx_replay_add(self.rb.h, self.obs.ptr, self.act.ptr, self.rew.ptr, self.done.ptr,
                                       self.nobs.ptr, self.n, None, 1))
