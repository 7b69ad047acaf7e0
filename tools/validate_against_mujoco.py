#!/usr/bin/env python
"""Pin or refute this repo's physics engines against MuJoCo in one command (VERDICT r2 "physics parity UNPINNED").

The reference steps MuJoCo 2.1 through gym 0.22 (`rlkit/envs/envs_dict.py:5-12`, requirements.txt:12); neither exists in the build
container, so the five device steppers (Hopper, Walker2d, HalfCheetah: planar engine, oracle/planar_env.py; Ant, Humanoid: 3-D engine,
oracle/spatial_env.py) have only ever been compared with this repo's own float64 oracle.  This tool makes the comparison a single
command for whoever has MuJoCo:

    python tools/validate_against_mujoco.py dump  --out triples.npz [--envs hopper walker2d halfcheetah ant humanoid] [--n 256]
        no MuJoCo needed: for every env, n states (q, v) reached by random-action rollouts of THIS engine (CPU oracle), the actions a
        applied there, and this engine's (q', v', reward, done) one env-step later.  Arrays only.
    python tools/validate_against_mujoco.py check --triples triples.npz
        needs `gym` (0.22) + `mujoco_py` (2.1) — or `gymnasium[mujoco]` with the -v2/-v4 XMLs: sets MuJoCo to every (q, v), applies a,
        steps once (frame_skip included) and reports, per env, max / median |dq'|, |dv'|, |d reward| and the fraction of agreeing
        `done` flags.  Exit code 0 if every env is within --tol (default 1e-3 on q', 1e-2 on v'), 1 otherwise, 2 if MuJoCo is absent.

State conventions: planar envs use MuJoCo's own qpos / qvel order (rootx, rootz, rooty, joints); the 3-D envs use qpos = position +
quaternion (w x y z) + hinge angles and qvel = world linear velocity + BODY-frame angular velocity + hinge rates, which is MuJoCo's free-
joint convention, so (q, v) are passed to `set_state` unchanged.
"""
import argparse
import sys

import numpy as np

ENVS = dict(hopper="Hopper-v2", walker2d="Walker2d-v2", halfcheetah="HalfCheetah-v2", ant="Ant-v2", humanoid="Humanoid-v2")


def _engine(name):
    """(oracle with reset(rng) -> (q, v) and step(q, v, a) -> (q', v', obs, r, done), act_dim) of this repo's CPU engine for `name`."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if name in ("hopper", "walker2d", "halfcheetah"):
        from ilswiss_amd.envs.models import MODELS
        from oracle.planar_env import PlanarOracle
        m = MODELS[name]()
        return PlanarOracle(m), int(sum(1 for g in m["gear"] if g != 0))
    from ilswiss_amd.envs.models3d import MODELS3D
    from oracle.spatial_env import SpatialOracle
    m = MODELS3D[name]()
    return SpatialOracle(m), len(m["act_links"])


def dump(args):
    rng = np.random.default_rng(args.seed)
    out = {}
    for name in args.envs:
        eng, adim = _engine(name)
        q, v = eng.reset(rng)
        Q, V, A, Q2, V2, R, D = [], [], [], [], [], [], []
        while len(Q) < args.n:
            a = rng.uniform(-1, 1, adim)
            q2, v2, _, r, done = eng.step(q, v, a)
            Q.append(q), V.append(v), A.append(a), Q2.append(q2), V2.append(v2), R.append(r), D.append(bool(done))
            q, v = (q2, v2) if not (done or rng.random() < 0.02) else eng.reset(rng)
        for k, val in (("q", Q), ("v", V), ("a", A), ("q2", Q2), ("v2", V2), ("r", R), ("done", D)):
            out[f"{name}_{k}"] = np.asarray(val)
        print(f"{name}: {len(Q)} triples, {int(np.sum(D))} terminal", file=sys.stderr)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out)
    return 0


def check(args):
    try:
        import gym
    except ImportError:
        try:
            import gymnasium as gym
        except ImportError:
            print("neither gym nor gymnasium is importable: MuJoCo is absent here (as in the build container); nothing checked", file=sys.stderr)
            return 2
    t = np.load(args.triples)
    bad = False
    for name in sorted({k.split("_")[0] for k in t.files}):
        env = gym.make(ENVS[name]).unwrapped
        env.reset()
        dq, dv, dr, agree = [], [], [], []
        for q, v, a, q2, v2, r, d in zip(*[t[f"{name}_{k}"] for k in ("q", "v", "a", "q2", "v2", "r", "done")]):
            env.set_state(q, v)
            step = env.step(a)
            done = bool(step[2]) if len(step) == 4 else bool(step[2] or step[3])
            sim = env.sim.data if hasattr(env, "sim") else env.data
            dq.append(np.abs(np.asarray(sim.qpos).ravel() - q2).max()), dv.append(np.abs(np.asarray(sim.qvel).ravel() - v2).max())
            dr.append(abs(float(step[1]) - r)), agree.append(done == bool(d))
        dq, dv, dr = np.asarray(dq), np.asarray(dv), np.asarray(dr)
        ok = dq.max() <= args.tol_q and dv.max() <= args.tol_v
        bad |= not ok
        print(f"{name:12s} |dq'| max {dq.max():.3e} med {np.median(dq):.3e}   |dv'| max {dv.max():.3e} med {np.median(dv):.3e}   "
              f"|dr| max {dr.max():.3e}   done agree {np.mean(agree):.3f}   {'PINNED' if ok else 'REFUTED at this tolerance'}")
    return 1 if bad else 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    d = sub.add_parser("dump")
    d.add_argument("--out", required=True)
    d.add_argument("--envs", nargs="+", default=list(ENVS))
    d.add_argument("--n", type=int, default=256)
    d.add_argument("--seed", type=int, default=0)
    c = sub.add_parser("check")
    c.add_argument("--triples", required=True)
    c.add_argument("--tol-q", type=float, default=1e-3)
    c.add_argument("--tol-v", type=float, default=1e-2)
    a = ap.parse_args()
    sys.exit(dump(a) if a.cmd == "dump" else check(a))
