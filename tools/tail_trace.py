"""Tiny SAC training run for a rocprofv3 kernel trace (ILSX_NO_GRAPH=1): 60 gradient steps at bench.py's sizes."""
import os, sys
import numpy as np
os.environ.setdefault("ILSX_NO_GRAPH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
ctx = ia.Context(0, seed=0)
o, a, H, B, CAP = 11, 3, 256, 256, 100_000
rng = np.random.default_rng(0)
rb = ia.SimpleReplayBuffer(CAP, o, a, random_seed=1, ctx=ctx)
rb.add_rows(rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
            rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
pol = ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=1)
q1, q2 = ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=2), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3)
tr = ia.SoftActorCritic(pol, q1, q2, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rb, 60, B)
ctx.sync()
