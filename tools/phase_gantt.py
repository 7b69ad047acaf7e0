"""Stage timestamps of the merged phase kernels (k_sac_phase_a / _c) of ONE SAC step at bench.py's sizes, from the measurement build
(make STAMPS=1): per launch and per task row, the median time of every stage boundary after the launch's first workgroup start.

    python tools/phase_gantt.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

os.environ["ILSX_NO_GRAPH"] = "1"
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.environ.get("ILSX_GANTT_NO_BUILD"):
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "ilswiss_amd", "csrc"), "-j8", "STAMPS=1"], stdout=subprocess.DEVNULL)
os.environ["ILSX_LIB"] = os.path.join(_ROOT, "ilswiss_amd", "libilsx_stamps.so")
sys.path.insert(0, _ROOT)
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd import _lib  # noqa: E402

MAXWG, SLOTS, MAXL = 2048, 8, 14
ctx = ia.Context(0, seed=0)
o, a, H, B, CAP = 11, 3, 256, 256, 100_000
rng = np.random.default_rng(0)
rows = (rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
        rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
rb = ia.SimpleReplayBuffer(CAP, o, a, random_seed=1, ctx=ctx)
rb.add_rows(*rows)
tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=1), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=2),
                        ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3), policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rb, 20, B)
ctx.sync()
buf = ctx.from_numpy(np.zeros((MAXL, MAXWG, SLOTS), np.int64), np.int64)
for rep in range(2):
    buf.copy_from(np.zeros((MAXL, MAXWG, SLOTS), np.int64))
    _lib.check(ctx.lib.ilsx_debug_set_stamp_buffer(ctx.h, buf.ptr, MAXL, None))
    tr.train_from_replay(rb, 3, B)
    ctx.sync()
    n = C.c_int()
    _lib.check(ctx.lib.ilsx_debug_set_stamp_buffer(ctx.h, None, 0, C.byref(n)))
    buf_raw = buf.numpy()
    t = buf_raw.astype(np.float64) * 0.01   # 100 MHz ticks -> us
    print(f"--- rep {rep}: {n.value} instrumented launches (3 steps of A D1 C D2)")
    base = None
    for L in range(4, min(n.value, 8)):
        live = np.flatnonzero(t[L][:, 0] > 0)
        if not live.size:
            continue
        s0 = t[L][live, 0].min()
        if base is None:
            base = s0
        kind = "ACAC"[L % 4] if False else ("A", "D1", "C", "D2")[L % 4]
        print(f"launch {L} ({kind}): {live.size} workgroups, begins {s0 - base:7.2f} us after the step's first launch")
        if kind in ("A", "C"):
            ny = 5 if kind == "A" else 4
            y = (live // 16) % ny
            for yy in range(ny - 1):
                m = live[y == yy]
                row = []
                for sl in range(8):
                    v = t[L][m, sl]
                    ok = v > 0
                    if ok.any():
                        row.append(f"s{sl}@{np.median(v[ok]) - s0:6.2f}(max {v[ok].max() - s0:6.2f})")
                print(f"    task row {yy}: " + "  ".join(row))
                if kind == "A" and yy in (1, 2):     # fine build: slots 5 / 6 hold the shader clock at the same two points as stamps 0 / 4
                    raw = buf_raw[L][m]
                    ok = (raw[:, 5] > 0) & (raw[:, 6] > 0) & (raw[:, 4] > 0)
                    if ok.any():
                        cyc = (raw[ok, 6] - raw[ok, 5]).astype(np.float64)
                        us = (raw[ok, 4] - raw[ok, 0]).astype(np.float64) * 0.01
                        print(f"        shader clock over the workgroup's life: {np.median(cyc / us):.0f} MHz (s_memtime cycles / 100 MHz stamps)")
        else:
            e = t[L][live, 7]
            p1, p2 = t[L][live, 1], t[L][live, 2]
            print(f"    span {e.max() - s0:6.2f}   workgroup starts: median {np.median(t[L][live, 0]) - s0:5.2f} max {t[L][live, 0].max() - s0:5.2f};  "
                  f"operands + MFMA done @ {np.median(p1) - s0:5.2f} (max {p1.max() - s0:5.2f});  partials in LDS @ {np.median(p2) - s0:5.2f} (max {p2.max() - s0:5.2f});  "
                  f"end median {np.median(e) - s0:5.2f}")
    nxt = np.flatnonzero(t[8][:, 0] > 0)
    if nxt.size and base is not None:
        print(f"    step: {t[8][nxt, 0].min() - base:.2f} us from A to the next A")
