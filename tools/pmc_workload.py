#!/usr/bin/env python
"""Workload for the rocprofv3 counter passes (tools/pmc_collect.sh): the HBM-bound replay gather, the SAC step's
MFMA kernels (hipGraph bypassed: ILSX_NO_GRAPH=1, rocprofv3 cannot trace graph launches here) and the PPO GAE scan,
a few launches each at bench.py's sizes."""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("ILSX_NO_GRAPH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd  # noqa: E402
from ilswiss_amd import _lib  # noqa: E402
from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy  # noqa: E402
from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy  # noqa: E402
from ilswiss_amd.replay import SimpleReplayBuffer  # noqa: E402
from ilswiss_amd.sac import SoftActorCritic  # noqa: E402

ctx = ilswiss_amd.Context(0, seed=0)
o, a, H, B, CAP = 11, 3, 256, 256, 1_000_000
rng = np.random.default_rng(0)
rb = SimpleReplayBuffer(CAP, o, a, random_seed=1, ctx=ctx)
rb.add_rows(rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
            rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
rec = C.c_int()
_lib.check(ctx.lib.ilsx_replay_record_floats(rb.h, C.byref(rec)))
nb = 4096
out = ctx.empty((nb * B, rec.value))
for _ in range(4):
    _lib.check(ctx.lib.ilsx_replay_sample_many(rb.h, nb, B, out.ptr))
ctx.sync()
pol = ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=1)
q1, q2 = FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=2), FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3)
tr = SoftActorCritic(pol, q1, q2, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rb, 50, B)
ctx.sync()
n_env, T = 8192, 128
N = n_env * T
ppol = ReparamMultivariateGaussianPolicy([H, H], o, a, conditioned_std=False, hidden_activation="tanh", ctx=ctx, seed=4)
vf = FlattenMlp([H, H], 1, o, hidden_activation="tanh", ctx=ctx, seed=5)
ppo = PPO(ppol, vf, mini_batch_size=32768, update_epoch=1, gae_tau=0.95, max_samples=N)
obs = ctx.from_numpy(rng.normal(0, 1, (N, o)).astype(np.float32))
act = ctx.from_numpy(rng.normal(0, 0.5, (N, a)).astype(np.float32))
rew = ctx.from_numpy(rng.normal(1, 1, (N,)).astype(np.float32))
offs = (np.arange(n_env + 1) * T).astype(np.int32)
for _ in range(3):
    _lib.check(ctx.lib.ilsx_ppo_gae(ppo.h, obs.ptr, act.ptr, rew.ptr, offs.ctypes.data_as(C.c_void_p), n_env, None, None, None, None, None))
ctx.sync()
print("pmc workload done")
