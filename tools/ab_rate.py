"""A/B of the single-run SAC step time between library builds / environment settings:
    python tools/ab_rate.py [label=ENV1=v,ENV2=v ...]     (label 'prev' loads ilswiss_amd/libilsx_prev.so)
Each variant runs in its own process (the library reads its switches once); REPS x STEPS steps, min and median us/step."""
import os, subprocess, sys, json
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import ilswiss_amd as ia
o, a, H, B, CAP = 11, 3, 256, 256, 100000
ctx = ia.Context(0, seed=0)
rng = np.random.default_rng(0)
rb = ia.SimpleReplayBuffer(CAP, o, a, ctx=ctx)
rb.add_rows(rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32), rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=1), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=2), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3), policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rb, 500, B); ctx.sync()
out = []
for _ in range(%d):
    t0 = time.perf_counter(); tr.train_from_replay(rb, %d, B); ctx.sync()
    out.append(1e6 * (time.perf_counter() - t0) / %d)
print("RES", " ".join("%%.2f" %% v for v in out))
'''
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS, STEPS = 2, 2500
variants = sys.argv[1:] or ["base="]
res = {}
for rnd in range(10):   # interleaved rounds, order reversed every other round: per-PROCESS offsets of +-1 us (placement, clocks) dominate, so many short processes
    for v in (variants if rnd % 2 == 0 else variants[::-1]):
        label, _, envs = v.partition("=")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition(":")
            env[k] = val
        if label.startswith("prev"):
            env["ILSX_LIB"] = os.path.join(ROOT, "ilswiss_amd", "libilsx_prev.so")
        o = subprocess.run([sys.executable, "-c", CHILD % (ROOT, REPS, STEPS, STEPS)], env=env, capture_output=True, text=True).stdout
        vals = [float(x) for l in o.splitlines() if l.startswith("RES") for x in l.split()[1:]]
        res.setdefault(label, []).extend(vals)
for k, v in res.items():
    v = sorted(v)
    print(f"{k:24s} min {v[0]:6.2f}  median {v[len(v)//2]:6.2f}  mean {sum(v)/len(v):6.2f}  max {v[-1]:6.2f}  us/step   ({len(v)} x {STEPS} steps)")
