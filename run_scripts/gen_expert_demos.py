#!/usr/bin/env python
"""Roll an expert policy out into the demonstration wire format the adversarial-IRL scripts read — the role of the
reference's run_scripts/gen_expert_demos.py:31-146: a pickled `list[dict]`, one dict per trajectory, with float arrays
observations [T,o], actions [T,a], rewards [T,1], next_observations [T,o], terminals [T,1] (and absorbings [T,2]).

    python run_scripts/gen_expert_demos.py --snapshot logs/.../best.pkl --env hopper --num-trajs 50 --out demos/hopper_sac.pkl

The snapshot is what DeviceRLAlgorithm saves (flat `policy` parameters of a ReparamTanhMultivariateGaussianPolicy)."""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.envs.vecenv import HipVectorEnv  # noqa: E402
from ilswiss_amd.samplers import rollout  # noqa: E402


def paths_to_demos(paths):
    out = []
    for p in paths:
        T = len(p)
        out.append(dict(observations=np.asarray(p["observations"], np.float64).reshape(T, -1),
                        actions=np.asarray(p["actions"], np.float64).reshape(T, -1),
                        rewards=np.asarray(p["rewards"], np.float64).reshape(T, 1),
                        next_observations=np.asarray(p["next_observations"], np.float64).reshape(T, -1),
                        terminals=np.asarray(p["terminals"], np.float64).reshape(T, 1),
                        absorbings=np.zeros((T, 2))))
    return out


def generate(policy, env, num_trajs, max_path_length=1000, no_terminal=False):
    demos = []
    while len(demos) < num_trajs:
        demos += paths_to_demos(rollout(env, ia.MakeDeterministic(policy), max_path_length, no_terminal=no_terminal))
    demos = demos[:num_trajs]
    rets = [float(d["rewards"].sum()) for d in demos]
    print(f"{len(demos)} trajectories, return {np.mean(rets):.1f} +- {np.std(rets):.1f}, length {np.mean([len(d['rewards']) for d in demos]):.0f}")
    return demos


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--snapshot", required=True)
    ap.add_argument("--env", default="hopper")
    ap.add_argument("--num-trajs", type=int, default=50)
    ap.add_argument("--max-path-length", type=int, default=1000)
    ap.add_argument("--net-size", type=int, default=256)
    ap.add_argument("--num-hidden-layers", type=int, default=2)
    ap.add_argument("--out", required=True)
    ap.add_argument("-g", "--gpu", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    ctx = ia.set_gpu_mode(True, a.gpu, seed=a.seed)
    env = HipVectorEnv(a.env, min(a.num_trajs, 64), seed=a.seed, ctx=ctx)
    with open(a.snapshot, "rb") as f:
        snap = pickle.load(f)
    pol = ia.ReparamTanhMultivariateGaussianPolicy(a.num_hidden_layers * [a.net_size], env.obs_dim, env.act_dim, ctx=ctx)
    pol.set_flat_params(snap["policy"])
    demos = generate(pol, env, a.num_trajs, a.max_path_length)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "wb") as f:
        pickle.dump(demos, f)
