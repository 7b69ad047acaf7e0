"""Discriminator steps/s: the fused use_bn=False path against the BatchNorm phase chain (csrc/disc_bn_step.h), GAIL Walker2d sizes
(23 -> 128 -> 128 -> 1, B = 256 per class), gradient penalty on.   python tools/discbn_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.adv_irl import MLPDisc  # noqa: E402

ctx = ia.Context(0, seed=0)
o, a, B = 17, 6, 256
rng = np.random.default_rng(0)
xe, xp = rng.normal(0, 1, (B, o + a)).astype(np.float32), rng.normal(0.3, 1.5, (B, o + a)).astype(np.float32)
from ilswiss_amd.device import as_dev  # noqa: E402
keep = [as_dev(ctx, np.ascontiguousarray(v)) for v in (xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:])]
ptrs = [k[1] for k in keep]
for bn in (False, True):
    d = MLPDisc(o + a, hid_dim=128, hid_act="tanh", use_bn=bn, ctx=ctx, seed=1).bind(o, max_batch=B, disc_lr=3e-4, disc_momentum=0.9, grad_pen_weight=8.0)
    step = lambda: ia._lib.check(ctx.lib.ilsx_disc_train_step(d.h, *ptrs, B, None, None))  # noqa: E731
    for _ in range(50):
        step()
    ctx.sync()
    n = 1000
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"use_bn={bn}: {n / dt:.0f} discriminator steps/s ({1e6 * dt / n:.0f} us per step)")
