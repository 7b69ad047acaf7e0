#!/usr/bin/env python
"""A/B: does the grouped train call (ilsx_sac_group, K = 10 Hopper runs, B = 512) run slower when the process holds other HIP streams?
  a: every agent in one ctx;  b: agents in sibling ctxs with streams of their own, idle;  c: as b, each sibling stream used (one tiny
  policy_act) between train calls;  d: as c with agents in ONE ctx and the extra streams unrelated to the group (no fences)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ilswiss_amd as ia  # noqa: E402

O, A, H, B, K, N = 11, 3, 256, 512, 10, 100_000
rng = np.random.default_rng(0)
rows = (rng.standard_normal((N, O), dtype=np.float32), np.tanh(rng.standard_normal((N, A), dtype=np.float32)), rng.standard_normal(N, dtype=np.float32),
        (rng.random(N) < 1e-3).astype(np.uint8), rng.standard_normal((N, O), dtype=np.float32))


def run(mode):
    base = ia.Context(0, seed=1)
    ctxs = [base] + [base.sibling(1 + k, share_stream=(mode in "ad")) for k in range(1, K)]
    extra = [ia.Context(0, seed=99 + k) for k in range(K - 1)] if mode == "d" else []
    rbs, trs, pols = [], [], []
    for k, c in enumerate(ctxs):
        rb = ia.SimpleReplayBuffer(N, O, A, random_seed=k, ctx=c)
        rb.add_rows(*rows)
        pol = ia.ReparamTanhMultivariateGaussianPolicy([H, H], O, A, ctx=c, seed=3 * k)
        tr = ia.SoftActorCritic(pol, ia.FlattenMlp([H, H], 1, O + A, ctx=c, seed=3 * k + 1), ia.FlattenMlp([H, H], 1, O + A, ctx=c, seed=3 * k + 2),
                                max_batch=B, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005)
        tr.eval_statistics = {}
        rbs.append(rb), trs.append(tr), pols.append(pol)
    epol = [ia.ReparamTanhMultivariateGaussianPolicy([H, H], O, A, ctx=c, seed=7) for c in extra]
    obs = [c.from_numpy(rows[0][:4]) for c in (ctxs if mode == "c" else extra)]
    act = [c.empty((4, A)) for c in (ctxs if mode == "c" else extra)]
    grp = ia.SoftActorCriticGroup(trs, ctx=base)

    def sync():
        for c in ctxs + extra:
            c.sync()
    grp.train_from_replay(rbs, 200, B)
    sync()
    ts = []
    for _ in range(5):
        if mode in "cd":
            for i, (p, c) in enumerate(zip(pols if mode == "c" else epol, ctxs if mode == "c" else extra)):
                ia._lib.check(c.lib.ilsx_policy_act(p.h, obs[i].ptr, 4, 0, None, act[i].ptr, None))
            sync()
        t0 = time.perf_counter()
        grp.train_from_replay(rbs, 1000, B)
        sync()
        ts.append(time.perf_counter() - t0)
    grp.close()
    for c in extra + ctxs[1:] + [base]:
        c.close()
    return min(ts), sorted(ts)[len(ts) // 2]


for mode in (sys.argv[1] if len(sys.argv) > 1 else "abcd"):
    lo, med = run(mode)
    print(json.dumps(dict(mode=mode, us_per_lockstep_min=1e3 * lo, us_per_lockstep_median=1e3 * med)), flush=True)
