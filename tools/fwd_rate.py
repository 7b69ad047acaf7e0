#!/usr/bin/env python
"""Large-batch forward of a 256-256 net: HIP-event time per launch for relu / tanh hidden layers (k_mlp_fwd; the large-batch kernels this
tool compared it with in round 4 are gone: DESIGN 3g, profiles/r04_ppo_fwd.txt).

    [ACTS="relu tanh"] python tools/fwd_rate.py [rows] [obs_dim]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.networks import as_dev  # noqa: E402
from bench_aux import prof_slots  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
o = int(sys.argv[2]) if len(sys.argv) > 2 else 11
ctx = ia.Context(0, seed=1)
x = np.random.default_rng(0).normal(0, 1, (n, o)).astype(np.float32)
keep, p = as_dev(ctx, x)
flop = 2.0 * n * (16 * 256 + 256 * 256 + 256)
for act in (os.environ.get("ACTS", "relu tanh").split()):
    net = ia.FlattenMlp([256, 256], 1, o, hidden_activation=act, ctx=ctx, seed=3)
    net.forward_dev(p, n); ctx.sync()
    prof = prof_slots(ctx, lambda: ([net.forward_dev(p, n) for _ in range(50)], ctx.sync()))
    for kid, (name, nl, ms) in prof.items():
        us = 1e3 * ms / nl
        print(f"{act} {name}: {us:.1f} us/launch, {flop / us / 1e6:.1f} TFLOP/s ({flop / us / 1e6 / 157.3:.3f} of fp32 MFMA peak)")
