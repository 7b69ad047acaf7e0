"""CPU suite: the DEVICE code of the 3-D stepper (ilswiss_amd/csrc/env3d.h lane-per-env, env3d_wave.h wave-per-env), compiled for the
host by tests/harness/env3d_host.cpp, against oracle/spatial_env.py — tree recursions vs dense Jacobians + numpy solves.  The
wave-per-env form is emulated with every parallel loop run serially, once ascending and once descending: a dependence between
iterations of one parallel loop (a race on the GPU) makes the two orders disagree with the oracle.  The GPU suite (tests/test_env3d_hip.py) repeats
this through the C ABI; this file lets the recursions be checked where there is no GPU."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from ilswiss_amd.envs.models3d import ant, humanoid
from ilswiss_amd.envs.vecenv import spatial_struct
from oracle.spatial_env import SpatialOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    out = os.path.join(tempfile.mkdtemp(prefix="e3h_"), "libe3h.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "harness", "env3d_host.cpp"), "-o", out])
    lib = C.CDLL(out)
    lib.e3h_step.restype = C.c_int
    lib.e3h_qacc.restype = C.c_int
    return lib


def _step(harness, form, sm, q, v, act, obs, r, d):
    if form == "lane":
        return harness.e3h_step(C.byref(sm), _p(q), _p(v), _p(act), _p(obs), C.byref(r), C.byref(d))
    return harness.e3hw_step(C.byref(sm), WAVE_FLAGS[form], _p(q), _p(v), _p(act), _p(obs), C.byref(r), C.byref(d))


def _qacc(harness, form, sm, q, v, ctrl, out):
    if form == "lane":
        return harness.e3h_qacc(C.byref(sm), _p(q), _p(v), _p(ctrl), _p(out))
    return harness.e3hw_qacc(C.byref(sm), WAVE_FLAGS[form], _p(q), _p(v), _p(ctrl), _p(out))


# env3d.h ; env3d_wave.h with its parallel loops run in either order, dof count at run time or (the device's instantiations) compile time
WAVE_FLAGS = {"wave-ascending": 0, "wave-descending": 1, "wave-static-ascending": 2, "wave-static-descending": 3}
FORMS = ["lane"] + list(WAVE_FLAGS)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _spread(m, rng, low):
    q = np.asarray(m["init_qpos"], float).copy()
    q[2] += rng.uniform(-0.35, 0.3) if low else rng.uniform(0.5, 1.5)
    q[3:7] += rng.normal(0, 0.3, 4); q[3:7] /= np.linalg.norm(q[3:7])
    q[7:] += rng.uniform(-0.8, 0.8, m["nq"] - 7)
    v = rng.normal(0, 1.5, m["nv"])
    return q, v


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("mf", [ant, humanoid])
def test_device_dynamics_match_oracle(harness, mf, form):
    m = mf()
    P = SpatialOracle(m)
    sm = spatial_struct(m)
    assert harness.e3h_obs_dim(C.byref(sm)) == m["obs_dim"]
    rng = np.random.default_rng(5)
    n_rows = []
    for it in range(24):
        q, v = _spread(m, rng, low=it % 3 != 0)
        ctrl = rng.uniform(-1, 1, m["act_dim"]) * m["ctrl_range"]
        ref = P.dynamics(q, v, ctrl)
        got = np.empty(m["nv"])
        assert _qacc(harness, form, sm, q, v, ctrl, got) == 0
        np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-7 * max(1.0, np.abs(ref).max()), err_msg=f"{mf.__name__} it {it}")
        n_rows.append(not np.allclose(ref, P.dynamics(q + np.r_[0, 0, 10.0, np.zeros(m["nq"] - 3)], v, ctrl)))
    assert any(n_rows)    # some of the sampled states are in contact


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("mf", [ant, humanoid])
def test_device_step_matches_oracle(harness, mf, form):
    m = mf()
    P = SpatialOracle(m)
    sm = spatial_struct(m)
    rng = np.random.default_rng(11)
    dones = []
    for it in range(6):
        q, v = _spread(m, rng, low=it % 2 == 0)
        for k in range(3):
            act = rng.uniform(-1.3, 1.3, m["act_dim"]).astype(np.float32)
            qo, vo, oo, ro, do = P.step(q.copy(), v.copy(), act)
            obs, r, d = np.empty(m["obs_dim"]), C.c_double(), C.c_int()
            assert _step(harness, form, sm, q, v, act, obs, r, d) == 0
            np.testing.assert_allclose(q, qo, rtol=1e-8, atol=1e-8, err_msg=f"qpos {it}.{k}")
            np.testing.assert_allclose(v, vo, rtol=1e-7, atol=1e-6, err_msg=f"qvel {it}.{k}")
            np.testing.assert_allclose(obs, oo, rtol=1e-7, atol=1e-6, err_msg=f"obs {it}.{k}")
            np.testing.assert_allclose(r.value, ro, rtol=1e-7, atol=1e-6)
            assert bool(d.value) == bool(do)
            dones.append(bool(do))
            q, v = qo, vo
    assert not all(dones)
