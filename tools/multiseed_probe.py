"""Co-resident seeds on one GPU: K independent SAC runs (own context = own HIP stream, own replay ring, own hipGraph)
issued round-robin from one process; aggregate grad-steps/s vs K."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
from ilswiss_amd.replay import SimpleReplayBuffer

o, a, H, B, CAP = 11, 3, 256, 256, 200_000
rng = np.random.default_rng(0)
data = (rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
        rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))


def make(seed):
    ctx = ia.Context(0, seed=seed)
    rb = SimpleReplayBuffer(CAP, o, a, random_seed=seed, ctx=ctx)
    rb.add_rows(*data)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=seed)
    q1, q2 = ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=seed + 1), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=seed + 2)
    tr = ia.SoftActorCritic(pol, q1, q2, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
    tr.eval_statistics = {}
    return ctx, rb, tr


for K in (1, 2, 4, 8):
    runs = [make(100 * k) for k in range(K)]
    for ctx, rb, tr in runs:
        tr.train_from_replay(rb, 300, B)
    for ctx, rb, tr in runs:
        ctx.sync()
    n, chunk = 4000, 500
    t0 = time.perf_counter()
    for _ in range(n // chunk):
        for ctx, rb, tr in runs:
            tr.train_from_replay(rb, chunk, B)
    for ctx, rb, tr in runs:
        ctx.sync()
    dt = time.perf_counter() - t0
    print(f"K={K}: round-robin from one thread: aggregate {K * n / dt:9.0f} grad-steps/s, per run {n / dt:8.0f}", flush=True)
    import threading

    def work(run):
        ctx, rb, tr = run
        for _ in range(n // chunk):
            tr.train_from_replay(rb, chunk, B)
        ctx.sync()
    ths = [threading.Thread(target=work, args=(r,)) for r in runs]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    print(f"K={K}: one host thread per run:       aggregate {K * n / dt:9.0f} grad-steps/s, per run {n / dt:8.0f}", flush=True)
    for ctx, rb, tr in runs:
        ctx.close()
