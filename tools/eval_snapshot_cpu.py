#!/usr/bin/env python
"""Cross-engine evaluation: a policy TRAINED on the HIP engine, PLAYED in the CPU restatement of the simulator.

    python tools/eval_snapshot_cpu.py gpurun_out/r05_returns_snap/seed20.params.pkl [...]

Loads the run's snapshot (ilswiss_amd/algorithm.py: params.pkl = trainer.get_snapshot() + the epoch's statistics), evaluates the policy's
deterministic action tanh(mean) with oracle/mlp.py (numpy) in oracle/planar_env.c (the fp64 C statement of the Hopper model) on the reference's
evaluation protocol — whole rollouts of a 4-env vec env until >= 10000 steps (rlkit/samplers/vec_sampler.py:5-97,126-146) — and prints the mean
return beside the "Test Returns Mean" the HIP engine itself logged for that snapshot's epoch.  The two differ only in the reset noise they drew
(numpy generator / Philox) and in fp32 (device policy) vs fp32-in-numpy arithmetic: if the device-side stepper, policy inference and evaluation
sampler compute what the oracle computes, the two numbers agree to within the spread of ~10 episodes."""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def evaluate(flat_pi, seed=0, eval_steps=10000, env_num=4, max_path_length=1000, net=256, layers=2):
    from returns_cpu import CpuVecEnv, load_stepper
    from ilswiss_amd.envs.models import MODELS
    from oracle import mlp as omlp
    env = CpuVecEnv(load_stepper(), MODELS["hopper"](), env_num, np.random.default_rng(seed))
    hidden = layers * [net]
    rets, lens, total = [], [], 0
    while total < eval_steps:
        ready = np.arange(env_num)
        obs = env.reset(ready)
        ret, ln = np.zeros(env_num), np.zeros(env_num, int)
        for _ in range(max_path_length):
            mu = omlp.forward(np.asarray(flat_pi, np.float32), np.asarray(obs, np.float32), env.o, hidden, env.a, n_heads=2)[0][0]
            nobs, rew, term = env.step(np.tanh(mu).astype(np.float64), ready)
            ret[ready] += rew; ln[ready] += 1
            obs, ready = nobs[~term], ready[~term]
            if len(ready) == 0:
                break
        rets += list(ret); lens += list(ln); total += int(ln.sum())
    return np.array(rets), np.array(lens)


def main():
    print("| snapshot | epoch | HIP engine's own evaluation (Test Returns Mean, paths) | the same policy in the CPU stepper: mean +- sd (paths, mean length) |")
    print("|---|---|---|---|")
    for path in sys.argv[1:]:
        with open(path, "rb") as f:
            snap = pickle.load(f)
        st = snap.get("statistics", {})
        rets, lens = evaluate(snap["policy"])
        print(f"| {os.path.basename(path)} | {snap.get('epoch')} | {st.get('Test Returns Mean', float('nan')):.0f} ({int(st.get('Num Paths', st.get('Test Num Paths', 0)))}) | "
              f"{rets.mean():.0f} +- {rets.std():.0f} ({len(rets)}, {lens.mean():.0f}) |", flush=True)


if __name__ == "__main__":
    main()
