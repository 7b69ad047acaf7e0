#!/usr/bin/env python
"""Do two builds of the library step the SAME states to the SAME bits?  (The acceptance check of a stepper rewrite that claims unchanged arithmetic.)

    ILSX_LIB=ilswiss_amd/libilsx_<old>.so python tools/stepper_bits.py dump /tmp/old.npz [steps]
    python tools/stepper_bits.py dump /tmp/new.npz
    python tools/stepper_bits.py compare /tmp/old.npz /tmp/new.npz

dump: every model, 256 envs (Ant / Humanoid: 64), 300 auto-reset rollout steps with random actions from the seeded device stream — contacts,
joint limits, terminations and resets on the way — then qpos / qvel of every env.  compare: exact equality, or the largest difference."""
import sys

import numpy as np

sys.path.insert(0, ".")


def dump(path, steps=300):
    import ilswiss_amd as ia
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    ctx = ia.Context(0, seed=11)
    out = {}
    for name, n in (("hopper", 256), ("walker", 256), ("halfcheetah", 256), ("ant", 64), ("humanoid", 64)):
        env = HipVectorEnv(name, n, seed=3, ctx=ctx)
        rb = ia.SimpleReplayBuffer(400 * n, env.obs_dim, env.act_dim, ctx=ctx)
        env.reset()
        for _ in range(steps):
            env.rollout_step(replay=rb, random_actions=True, max_path_length=1000)
        ctx.sync()
        q, v = env.get_state()
        out[name + "_q"], out[name + "_v"] = q, v
        env.close()
    np.savez(path, **out)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k], B[k])
        d = float(np.max(np.abs(A[k] - B[k])))
        print(f"{k:16s} {'identical' if same else 'DIFFERENT'}  max |diff| {d:.3e}  finite {bool(np.isfinite(A[k]).all() and np.isfinite(B[k]).all())}")
        bad += not same
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 300)
    else:
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
