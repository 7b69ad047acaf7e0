#!/usr/bin/env python
"""Clock ticks per stage of the planar stepper's dynamics evaluation (csrc/env2d_group.h, ILSX_EG_PROFILE), workgroup 0's wavefront.

    make -C ilswiss_amd/csrc VAR=egprof VARFLAGS=-DILSX_EG_PROFILE
    ILSX_LIB=ilswiss_amd/libilsx_egprof.so python tools/env2d_phases.py [hopper|walker|halfcheetah] [n_env]

The stamps (a full memory wait + s_memtime each) lengthen the step they measure; the split is what this is for.
"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.envs.vecenv import HipVectorEnv  # noqa: E402

NAMES = ["kinematics + forces", "mass matrix + rhs", "Cholesky", "L store, L^T read", "qacc0 solves", "contact / limit rows",
         "row read, z = L^-1 j, rhs", "Z store, A = Z Z^T", "Gauss-Seidel", "q.. assemble"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hopper"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    ctx = ia.Context()
    env = HipVectorEnv(name, n, seed=1, ctx=ctx)
    rb = ia.SimpleReplayBuffer(64 * n, env.obs_dim, env.act_dim, ctx=ctx)
    env.reset()
    for _ in range(60):   # random actions: after a few dozen steps most envs stand or lie on their contacts
        env.rollout_step(replay=rb, random_actions=True, max_path_length=1000)
    ctx.sync()
    out = (C.c_ulonglong * 16)()
    assert ctx.lib.ilsx_debug_eg_prof(out, 1) == 0
    K = 50
    t0 = time.perf_counter()
    for _ in range(K):
        env.rollout_step(replay=rb, random_actions=True, max_path_length=1000)
    ctx.sync()
    dt = time.perf_counter() - t0
    assert ctx.lib.ilsx_debug_eg_prof(out, 0) == 0
    steps, total = out[14], out[15]
    print(f"{name}, {n} envs: {dt / K * 1e6:.0f} us per vec step (with stamps); wavefront 0: {total / steps:.0f} ticks per env step")
    inside = sum(out[i] for i in range(10))
    for i, nm in enumerate(NAMES):
        print(f"  {nm:28s} {out[i] / steps:9.0f} ticks / env step  {100.0 * out[i] / total:5.1f} %")
    for nm, v in (("prologue (state, actions)", out[10]), ("integrator glue + stamps", out[11] - inside), ("epilogue (records, reset)", out[12])):
        print(f"  {nm:28s} {v / steps:9.0f} ticks / env step  {100.0 * v / total:5.1f} %")
    env.close()


if __name__ == "__main__":
    main()
