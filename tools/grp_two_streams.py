#!/usr/bin/env python
"""K co-resident SAC seeds as S groups of K/S on S contexts (= S HIP streams), stepped concurrently from S host threads, against one group
of K: do two lock-steps in different phases hide each other's latencies and launch boundaries?

    python tools/grp_two_streams.py [K] [S] [n_steps]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd  # noqa: E402


def build(c, ks, o, a, H, B, CAP, data):
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.replay import SimpleReplayBuffer
    from ilswiss_amd.sac import SoftActorCritic, SoftActorCriticGroup
    rbs, trs = [], []
    for k in ks:
        rb = SimpleReplayBuffer(CAP, o, a, random_seed=k, ctx=c)
        rb.add_rows(*data)
        tr = SoftActorCritic(ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=c, seed=k), FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 1),
                             FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 2), policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
        tr.eval_statistics = {}
        rbs.append(rb), trs.append(tr)
    return SoftActorCriticGroup(trs), rbs


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    o, a, H, B, CAP = 11, 3, 256, 256, 50_000
    rng = np.random.default_rng(0)
    data = (rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
            rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
    ctxs = [ilswiss_amd.Context(0, seed=7 + s) for s in range(S)]
    groups = [build(ctxs[s], range(s * (K // S), (s + 1) * (K // S)), o, a, H, B, CAP, data) for s in range(S)]

    def run(steps):
        def work(s):
            g, rbs = groups[s]
            g.train_from_replay(rbs, steps, B)
            ctxs[s].sync()
        th = [threading.Thread(target=work, args=(s,)) for s in range(S)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        return time.perf_counter() - t0
    run(200)
    dt = run(n)
    print(f"K={K} as {S} group(s) of {K // S} on {S} stream(s): {1e6 * dt / n:.1f} us per lock-step of all K, {K * n / dt:.0f} aggregate grad-steps/s")


if __name__ == "__main__":
    main()
