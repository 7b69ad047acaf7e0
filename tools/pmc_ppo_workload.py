#!/usr/bin/env python
"""One PPO update epoch at BASELINE config 4 sizes (8192 x 128 samples, 32 minibatches of 32768, both nets) for tools/pmc_ppo.sh."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd import _lib  # noqa: E402
from ilswiss_amd.networks import FlattenMlp  # noqa: E402
from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy  # noqa: E402

ctx = ia.Context(0, seed=3)
rng = np.random.default_rng(0)
o, a, H, n_env, T = 11, 3, 256, 8192, 128
N = n_env * T
pol = ReparamMultivariateGaussianPolicy([H, H], o, a, conditioned_std=False, hidden_activation="tanh", ctx=ctx, seed=4)
vf = FlattenMlp([H, H], 1, o, hidden_activation="tanh", ctx=ctx, seed=5)
ppo = PPO(pol, vf, mini_batch_size=32768, update_epoch=1, gae_tau=0.95, max_samples=N)
obs = ctx.from_numpy(rng.normal(0, 1, (N, o)).astype(np.float32))
act = ctx.from_numpy(rng.normal(0, 0.5, (N, a)).astype(np.float32))
rew = ctx.from_numpy(rng.normal(1, 1, (N,)).astype(np.float32))
offs = (np.arange(n_env + 1) * T).astype(np.int32)
_lib.check(ctx.lib.ilsx_ppo_train(ppo.h, obs.ptr, act.ptr, rew.ptr, offs.ctypes.data_as(C.c_void_p), n_env, None, None))
ctx.sync()
print("pmc ppo workload done")
