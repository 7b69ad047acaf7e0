#!/usr/bin/env python
"""k_replay_sample_many (the HBM-bound kernel north_star names) at several record widths and ring sizes: us per launch and GB/s, algorithmic
(2 x (2 o + a + 2) x 4 bytes per row: SURVEY section 8d) and moved (the padded records, read + written).   python tools/replay_rate.py"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd import _lib  # noqa: E402

ctx = ia.Context(0, seed=1)
rng = np.random.default_rng(0)
for name, (o, a), cap in (("hopper 1e6 rows (128 MB ring)", (11, 3), 1_000_000), ("hopper 4e6 rows (512 MB ring)", (11, 3), 4_000_000),
                          ("walker 1e6 rows (256 MB)", (17, 6), 1_000_000), ("ant 4e5 rows (410 MB)", (111, 8), 400_000),
                          ("humanoid 2e5 rows (640 MB)", (376, 17), 200_000)):
    rb = ia.SimpleReplayBuffer(cap, o, a, random_seed=1, ctx=ctx)
    n = 200_000
    rows = (rng.standard_normal((n, o), dtype=np.float32), rng.standard_normal((n, a), dtype=np.float32), rng.standard_normal(n, dtype=np.float32),
            np.zeros(n, np.uint8), rng.standard_normal((n, o), dtype=np.float32))
    for _ in range(cap // n):
        rb.add_rows(*rows)
    rec = C.c_int()
    _lib.check(ctx.lib.ilsx_replay_record_floats(rb.h, C.byref(rec)))
    nb, B = (4096, 256) if rec.value <= 64 else (256, 256)
    out = ctx.empty((nb * B, rec.value))
    _lib.check(ctx.lib.ilsx_replay_sample_many(rb.h, nb, B, out.ptr))
    _lib.check(ctx.lib.ilsx_prof_reset(ctx.h))
    _lib.check(ctx.lib.ilsx_prof_enable(ctx.h, 1))
    for _ in range(10):
        _lib.check(ctx.lib.ilsx_replay_sample_many(rb.h, nb, B, out.ptr))
    _lib.check(ctx.lib.ilsx_prof_enable(ctx.h, 0))
    nl, ms = C.c_uint64(), C.c_double()
    _lib.check(ctx.lib.ilsx_prof_read(ctx.h, 6, C.byref(nl), C.byref(ms)))
    alg, mov = 2.0 * nb * B * (2 * o + a + 2) * 4, 2.0 * nb * B * rec.value * 4
    us = ms.value * 1e3 / nl.value
    print(f"{name}: record {rec.value} floats, {nb * B} rows per launch, {us:.1f} us, algorithmic {alg / us / 1e3:.0f} GB/s, moved {mov / us / 1e3:.0f} GB/s", flush=True)
    rb.close() if hasattr(rb, "close") else None
    out.free()
