#!/usr/bin/env python
"""Cycles per stage of the wave-per-env 3-D stepper (see env3d_phases.hip).  Usage (GPU box):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC tools/ubench/env3d_phases.hip -o /tmp/libe3p.so
    python tools/ubench/env3d_phases.py [humanoid|ant] [n_env] [n_steps]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ilswiss_amd.envs.models3d import MODELS3D  # noqa: E402
from ilswiss_amd.envs.vecenv import spatial_struct  # noqa: E402

NAMES = ["kinematics", "link inertia+wrench", "subtree sums", "rhs + CRBA rows", "Cholesky", "qacc0 solves", "contacts + row table",
         "Jacobian rows", "row solves z=L^-1 j", "A = Z Z^T", "Gauss-Seidel", "qacc assemble", "integrator / outside"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
    n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    m = MODELS3D[name]()
    sm = spatial_struct(m)
    lib = C.CDLL(os.environ.get("E3P_LIB", "/tmp/libe3p.so"))
    rng = np.random.default_rng(0)
    q = np.tile(np.asarray(m["init_qpos"], float), (n_env, 1)) + rng.uniform(-0.01, 0.01, (n_env, m["nq"]))
    q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    v = rng.uniform(-0.01, 0.01, (n_env, m["nv"]))
    act = rng.uniform(-1, 1, (n_env, m["act_dim"])).astype(np.float32)
    out, qo, ms = np.zeros((n_env, 16)), np.zeros((n_env, 64)), C.c_float()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib.e3p_run(C.byref(sm), n_env, n_steps, p(q), p(v), p(act), p(out), p(qo), C.byref(ms))
    assert rc == 0, rc
    evals = n_steps * m["frame_skip"] * 4
    tot = out[:, 13].mean()
    print(f"{name}: {n_env} envs x {n_steps} steps: {ms.value:.3f} ms  ({ms.value / n_steps:.3f} ms per vec-env step, "
          f"{n_env * n_steps / ms.value * 1e3:.0f} env-steps/s); mean wave {tot / n_steps:.0f} clk per step; z after = {qo[:, 2].mean():.3f}")
    for k, nm in enumerate(NAMES):
        c = out[:, k].mean() / evals
        print(f"  {nm:24s} {c:9.0f} clk / eval  {100 * out[:, k].mean() / tot:5.1f} %")


if __name__ == "__main__":
    main()
