"""Where does the wall time of one fused vec-env step go?  (1.28 ms per ilsx_rollout_step against 0.49 ms of kernels)
    python tools/rollout_overhead.py [n_env]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia  # noqa: E402
from bench_aux import prof_slots  # noqa: E402
from ilswiss_amd.envs.vecenv import HipVectorEnv  # noqa: E402

n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = ia.Context(0, seed=0)
pol = ia.ReparamTanhMultivariateGaussianPolicy([256, 256], 11, 3, ctx=ctx, seed=1)
rb = ia.SimpleReplayBuffer(1_000_000, 11, 3, random_seed=0, ctx=ctx)
env = HipVectorEnv("hopper", n_env, seed=0, ctx=ctx)


def timed(fn, n=200, sync_each=True):
    for _ in range(20):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each:
            ctx.sync()
    ctx.sync()
    return 1e6 * (time.perf_counter() - t0) / n


full = lambda: env.rollout_step(policy=pol, replay=rb, max_path_length=1000)  # noqa: E731
norb = lambda: env.rollout_step(policy=pol, replay=None, max_path_length=1000)  # noqa: E731
rnd = lambda: env.rollout_step(policy=None, replay=None, max_path_length=1000, random_actions=True)  # noqa: E731
print(f"n_env {n_env}")
print(f"policy + physics + replay insert, sync after each : {timed(full):8.1f} us")
print(f"policy + physics + replay insert, back to back    : {timed(full, sync_each=False):8.1f} us")
print(f"policy + physics, no replay, back to back         : {timed(norb, sync_each=False):8.1f} us")
print(f"random actions + physics, back to back            : {timed(rnd, sync_each=False):8.1f} us")
prof = prof_slots(ctx, lambda: [full() for _ in range(50)] and ctx.sync())
for kid, (name, nl, ms) in prof.items():
    print(f"   slot {kid:2d} {name[:60]:60s} {nl:5d} launches  {1e3 * ms / nl:8.1f} us each")
t0 = time.perf_counter()
for _ in range(200):
    ctx.sync()
print(f"ctx.sync() on an idle stream: {1e6 * (time.perf_counter() - t0) / 200:.1f} us")
