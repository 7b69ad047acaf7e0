"""Where does a SAC gradient step spend its time?  Per-workgroup phase timestamps (ilsx_debug_set_stamp_buffer) of every launch of
ONE step at bench.py's sizes, reduced to a text Gantt: for each launch its span (first workgroup start -> last workgroup end),
the start skew (dispatch ramp), the workgroup lifetime (median / max) and the phase medians.  Direct launches (ILSX_NO_GRAPH=1).

    python tools/step_gantt.py [K]        # K > 1: grouped step of K co-resident agents
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["ILSX_NO_GRAPH"] = "1"
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the timestamps are compiled in only in the measurement build of the library
import subprocess  # noqa: E402
subprocess.check_call(["make", "-C", os.path.join(_ROOT, "ilswiss_amd", "csrc"), "-j8", "STAMPS=1"], stdout=subprocess.DEVNULL)
os.environ["ILSX_LIB"] = os.path.join(_ROOT, "ilswiss_amd", "libilsx_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd import _lib  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
MAXWG, SLOTS, MAXL = 2048, 8, 26
ctx = ia.Context(0, seed=0)
o, a, H, B, CAP = 11, 3, 256, 256, 100_000
rng = np.random.default_rng(0)
rows = (rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
        rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
trs, rbs = [], []
for k in range(K):
    rb = ia.SimpleReplayBuffer(CAP, o, a, random_seed=1 + k, ctx=ctx)
    rb.add_rows(*rows)
    tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=3 * k + 1),
                            ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3 * k + 2), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3 * k + 3),
                            policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
    tr.eval_statistics = {}
    trs.append(tr), rbs.append(rb)
grp = ia.SoftActorCriticGroup(trs) if K > 1 else None


def steps(n):
    if grp:
        grp.train_from_replay(rbs, n, B)
    else:
        trs[0].train_from_replay(rbs[0], n, B)


steps(20)
ctx.sync()
buf = ctx.from_numpy(np.zeros((MAXL, MAXWG, SLOTS), np.int64), np.int64)
for rep in range(3):
    buf.copy_from(np.zeros((MAXL, MAXWG, SLOTS), np.int64))
    _lib.check(ctx.lib.ilsx_debug_set_stamp_buffer(ctx.h, buf.ptr, MAXL, None))
    steps(3)
    ctx.sync()
    n = C.c_int()
    _lib.check(ctx.lib.ilsx_debug_set_stamp_buffer(ctx.h, None, 0, C.byref(n)))
    t = buf.numpy().astype(np.float64) * 0.01   # 100 MHz ticks -> us
    t0 = None
    print(f"--- rep {rep}: {n.value} instrumented launches (3 steps of F1 F2 B1 D1 F3 B2 B3 D2)")
    for L in range(8, min(n.value, 16)):
        live = t[L][:, 0] > 0
        s, e = t[L][live, 0], t[L][live, 7]
        ok = e > 0
        if t0 is None:
            t0 = s.min()
        ph = []
        for i in (1, 2, 3):
            v = t[L][live, i]
            m = v > 0
            if m.any():
                ph.append(f"p{i}@{np.median(v[m] - s[m]):.2f}")
        print(f"launch {L}: wgs {live.sum():4d}  begin {s.min() - t0:7.2f}  span {e[ok].max() - s.min():6.2f}  start-skew p50 {np.median(s - s.min()):.2f} max {(s - s.min()).max():.2f}"
              f"  wg-life p50 {np.median(e[ok] - s[ok]):.2f} max {(e[ok] - s[ok]).max():.2f}   {' '.join(ph)}")
        if rep == 2 and L in (8, 11):
            life = e[ok] - s[ok]
            order = np.argsort(-life)[:6]
            print("     longest wgs:", [(int(np.flatnonzero(live)[ok][i]), round(float(life[i]), 2)) for i in order])
    print(f"    second step: {t[16][t[16][:, 0] > 0, 0].min() - t[8][t[8][:, 0] > 0, 0].min():.2f} us from F1 to the next F1")
