#!/usr/bin/env python
"""Grouped (co-resident seeds) SAC lock-step: time and per-kernel averages for one setting of the macro-tile knobs.

    ILSX_GRP_MT=4 [ILSX_LIB=ilswiss_amd/libilsx_<variant>.so] python tools/grp_sweep.py hopper 8 [n_steps]
    python tools/grp_sweep.py humanoid 4

One JSON line: K, dims, us per lock-step, aggregate grad-steps/s, and the HIP-event average of every kernel class of the step
(ilsx_prof_*: the dispatch's own begin / end stamps).  Synthetic replay contents of the task's widths (no env stepping).
tools/grp_sweep.sh runs the grid of settings in separate processes (the group's launch shape is fixed when it is built).
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd  # noqa: E402
from bench_aux import prof_slots, sac_flops, PEAK_F32_MFMA_TFLOPS  # noqa: E402


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "hopper"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    o, a = dict(hopper=(11, 3), walker=(17, 6), ant=(111, 8), humanoid=(376, 17))[task]
    H, B, CAP = 256, 256, 50_000
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.replay import SimpleReplayBuffer
    from ilswiss_amd.sac import SoftActorCritic, SoftActorCriticGroup
    rng = np.random.default_rng(0)
    data = (rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
            rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
    c = ilswiss_amd.Context(0, seed=7)
    rbs, trs = [], []
    for k in range(K):
        rb = SimpleReplayBuffer(CAP, o, a, random_seed=k, ctx=c)
        rb.add_rows(*data)
        tr = SoftActorCritic(ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=c, seed=k),
                             FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 1), FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 2),
                             policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
        tr.eval_statistics = {}
        rbs.append(rb), trs.append(tr)
    grp = SoftActorCriticGroup(trs)
    grp.train_from_replay(rbs, 200, B)
    c.sync()
    t0 = time.perf_counter()
    grp.train_from_replay(rbs, n, B)
    c.sync()
    dt = time.perf_counter() - t0
    npf = 50
    prof = prof_slots(c, lambda: (grp.train_from_replay(rbs, npf, B), c.sync()))
    sf = sac_flops(o, a, H, B)
    kern = {}
    for kid, (name, nl, ms) in prof.items():
        kern[name] = dict(launches_per_step=nl / npf, avg_us=1e3 * ms / nl)
        if kid in sf:
            kern[name]["tflops"] = K * npf * sf[kid] / (ms * 1e-3) / 1e12
            kern[name]["frac"] = kern[name]["tflops"] / PEAK_F32_MFMA_TFLOPS
    finite = all(np.isfinite(t.get_params("policy")).all() for t in trs)
    print(json.dumps(dict(task=task, K=K, mt=os.environ.get("ILSX_GRP_MT", "default"), lib=os.path.basename(os.environ.get("ILSX_LIB", "libilsx.so")),
                          us_per_lockstep=1e6 * dt / n, aggregate_grad_steps_per_s=K * n / dt, finite=finite, kernels=kern)), flush=True)
    grp.close()
    c.close()


if __name__ == "__main__":
    main()
