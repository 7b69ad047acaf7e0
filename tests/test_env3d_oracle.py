"""CPU suite: the 3-D articulated-body oracle (oracle/spatial_env.py) and the Ant / Humanoid models (ilswiss_amd/envs/models3d.py).
MuJoCo is absent, so physics parity with the reference is UNPINNED; what is checked: the body masses MuJoCo reports for
Humanoid-v2 (public knowledge), invariants of the engine (energy / momentum in free flight, symmetric positive-definite mass
matrix), the in-tree task formulas (rlkit/envs/mujoco/humanoid.py:37-73, ant.py:11-43) and the observation layout."""
import copy

import numpy as np

from ilswiss_amd.envs.models3d import ant, humanoid
from oracle.spatial_env import SpatialOracle, integrate_pos


def test_humanoid_body_masses_are_mujocos():
    m = humanoid()
    # Humanoid-v2 model.body_mass (MuJoCo 2.1, gym 0.22 humanoid.xml); shin + foot are one link here (the foot body has no joint)
    ref = dict(torso=8.32207894, lwaist=2.03575204, pelvis=5.85278711, right_thigh=4.52555626, right_shin=2.63249442 + 1.76714587,
               left_thigh=4.52555626, left_shin=2.63249442 + 1.76714587, right_upper_arm=1.59405984, right_lower_arm=1.19834313,
               left_upper_arm=1.59405984, left_lower_arm=1.19834313)
    got = {n: ms for n, ms in zip(m["names"], m["mass"]) if ms > 0}
    assert set(got) == set(ref)
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-6, err_msg=k)
    assert m["nq"] == 24 and m["nv"] == 23 and m["act_dim"] == 17 and m["obs_dim"] == 376 and len(m["body_link"]) == 13
    a = ant()
    assert a["nq"] == 15 and a["nv"] == 14 and a["act_dim"] == 8 and a["obs_dim"] == 111 and len(a["body_link"]) == 13


def _free(model):
    m = copy.deepcopy(model)
    m["damping"] = [0.0] * m["n_link"]
    m["limited"] = [0] * m["n_link"]
    return m


def test_free_flight_conserves_energy_and_momentum():
    for mf in (ant, humanoid):
        m = _free(mf())
        P = SpatialOracle(m)
        rng = np.random.default_rng(1)
        q = np.asarray(m["init_qpos"], float)
        q[2] = 5.0
        q[7:] = rng.uniform(-0.3, 0.3, m["nq"] - 7)
        q[3:7] = rng.normal(0, 1, 4); q[3:7] /= np.linalg.norm(q[3:7])
        v = rng.normal(0, 1.0, m["nv"])
        E0 = P.energy(q, v)
        M, c, _ = P.mass_bias(q, v)
        assert np.allclose(M, M.T, atol=1e-12) and np.linalg.eigvalsh(M).min() > 0
        tot = sum(m["mass"])
        com0 = np.array([P.com_x(q)])
        R, o, w, Jw, Jo, al, ao, vo = P.kin(q, v)
        p0 = sum(m["mass"][l] * (vo[l] + np.cross(w[l], R[l] @ np.asarray(m["com"][l]))) for l in range(m["n_link"]))
        n = 40
        for _ in range(n):
            q, v = P.substep(q, v, np.zeros(m["act_dim"]))
        h = m["timestep"]
        assert abs(P.energy(q, v) - E0) < 2e-7 * max(1.0, abs(E0)), (mf.__name__, P.energy(q, v) - E0)
        np.testing.assert_allclose(P.com_x(q), com0[0] + p0[0] / tot * n * h, atol=2e-6)      # no horizontal force (RK4 truncation only)
        np.testing.assert_allclose(np.linalg.norm(q[3:7]), 1.0, atol=1e-12)


def test_quaternion_integration_matches_the_rotation_it_encodes():
    q = np.array([0, 0, 0, 1.0, 0, 0, 0])
    v = np.array([0, 0, 0, 0.3, -0.2, 0.5])
    q1 = integrate_pos(q, v, 0.1)
    ang = np.linalg.norm(v[3:]) * 0.1
    np.testing.assert_allclose(q1[3], np.cos(ang / 2), atol=1e-15)
    np.testing.assert_allclose(q1[4:7], np.sin(ang / 2) * v[3:] / np.linalg.norm(v[3:]), atol=1e-15)


def test_humanoid_task_formulas_and_observation_layout():
    m = humanoid()
    P = SpatialOracle(m)
    rng = np.random.default_rng(0)
    q, v = P.reset(rng)
    assert np.all(np.abs(q[:3] - [0, 0, 1.4]) <= 0.01 + 1e-12) and np.all(np.abs(v) <= 0.01) and abs(np.linalg.norm(q[3:7]) - 1) < 1e-12
    act = rng.uniform(-1.5, 1.5, 17)
    ctrl = np.clip(act * 0.4, -0.4, 0.4)          # NormalizedBoxEnv onto ctrlrange +-0.4, then clip
    x0 = P.com_x(q)
    q1, v1, ob, rew, done = P.step(q.copy(), v.copy(), act)
    np.testing.assert_allclose(rew, 0.25 * (P.com_x(q1) - x0) / 0.003 - 0.1 * np.sum(ctrl ** 2) + 5.0, rtol=1e-12)   # humanoid.py:42-47
    assert not done and ob.shape == (376,)
    np.testing.assert_allclose(ob[:22], q1[2:]); np.testing.assert_allclose(ob[22:45], v1)
    cin = ob[45:185].reshape(14, 10); cv = ob[185:269].reshape(14, 6); qf = ob[269:292]; cf = ob[292:]
    assert not cin[0].any() and not cv[0].any() and not cf.any() and not qf[:6].any()
    np.testing.assert_allclose(cin[1, 9], 8.32207894, rtol=2e-6)                      # torso mass
    np.testing.assert_allclose(cin[1:, 9].sum(), sum(m["mass"]), rtol=1e-12)         # (welded feet report 0: their mass is the shin's)
    np.testing.assert_allclose(cin[1:, 6:9].sum(0), 0.0, atol=1e-9)                   # sum m (c_b - com) = 0
    # qfrc_actuator: gear * ctrl on the actuated dofs, in DOF order (abdomen_z before abdomen_y) vs actuator order (y before z)
    np.testing.assert_allclose(qf[6:9], [100 * ctrl[1], 100 * ctrl[0], 100 * ctrl[2]])
    np.testing.assert_allclose(qf[6 + 5], 300 * ctrl[5]); np.testing.assert_allclose(qf[-1], 25 * ctrl[16])
    # termination (humanoid.py:49): z < 1.0 or z > 2.0
    qq = q.copy(); qq[2] = 0.9
    assert P.step(qq, v, np.zeros(17))[4]
    # standing still for a while: alive, torso stays up, reward ~ alive bonus
    q, v = np.asarray(m["init_qpos"], float), np.zeros(23)
    for _ in range(8):
        q, v, ob, rew, done = P.step(q, v, np.zeros(17))
        assert not done
    assert 1.25 < q[2] < 1.45 and abs(rew - 5.0) < 1.0


def test_ant_task_formulas_and_settling():
    m = ant()
    P = SpatialOracle(m)
    rng = np.random.default_rng(0)
    q, v = P.reset(rng)
    assert np.all(np.abs(q[:3] - [0, 0, 0.75]) <= 0.1 + 1e-12) and 0.03 < v.std() < 0.3
    act = rng.uniform(-1.5, 1.5, 8)
    q1, v1, ob, rew, done = P.step(q.copy(), v.copy(), act)
    a = np.clip(act, -1, 1)
    np.testing.assert_allclose(rew, (q1[0] - q[0]) / 0.05 - 0.5 * np.sum(a * a) + 1.0, rtol=1e-12)      # ant.py:11-20
    assert ob.shape == (111,) and not ob[27:].any()
    np.testing.assert_allclose(ob[:13], q1[2:]); np.testing.assert_allclose(ob[13:27], v1)
    assert P.step(np.where(np.arange(15) == 2, 1.2, q), v, np.zeros(8))[4]   # ant.py:22: healthy only for 0.2 <= z <= 1.0
    q, v = np.asarray(m["init_qpos"], float), np.zeros(14)
    for _ in range(30):
        q, v, ob, rew, done = P.step(q, v, np.zeros(8))
    assert not done and 0.25 < q[2] < 0.8 and np.abs(v).max() < 2.0      # dropped from 0.75 onto its legs, at rest above the floor
    assert np.all(q[7:][1::2] * np.array([1, -1, -1, 1]) > 0.3)          # the ankle limits (30..70 deg) have pushed the lower legs in range


def test_c_restatement_matches_the_numpy_oracle():
    """oracle/spatial_env.c (the compiled CPU baseline of bench.py's Humanoid env-steps/s) == oracle/spatial_env.py over
    chained steps with ground contacts, violated joint limits and terminations, for both 3-D models: state, observation (all 376 / 111
    entries, cinert / cvel / qfrc_actuator included), reward and done."""
    import ctypes as C
    import os
    import subprocess
    from ilswiss_amd.envs.vecenv import spatial_struct
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(root, "oracle", "_build", "liborc_spatial.so"))
    lib.orc_spatial_bench.restype = C.c_double
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    for make in (humanoid, ant):
        m = make()
        S, ms = SpatialOracle(m), spatial_struct(m)
        assert lib.orc_spatial_obs_dim(C.byref(ms)) == m["obs_dim"]
        rng = np.random.default_rng(11)
        nv, na = m["nv"], m["act_dim"]
        dones = []
        for trial in range(5):
            q, v = S.reset(rng)
            q[2] += rng.uniform(-0.25, 0.15)                       # some start in the ground (contacts)
            if trial == 4:                                         # the last trial starts outside the healthy band (done): Humanoid z < 1, Ant z > 1
                q[2] = 0.8 if make is humanoid else 1.3
            q[7:] += rng.uniform(-1.2, 1.2, m["nq"] - 7)           # push hinges past their ranges (limit rows)
            v += rng.normal(0, 1.0, nv)
            qc, vc = q.copy(), v.copy()
            for s in range(3):
                a = rng.uniform(-1.3, 1.3, na)
                q, v, ob, r, d = S.step(q, v, a)
                obc, rc, dc = np.empty(m["obs_dim"]), C.c_double(), C.c_int()
                assert lib.orc_spatial_step(C.byref(ms), p(qc), p(vc), p(a), p(obc), C.byref(rc), C.byref(dc)) == 0
                np.testing.assert_allclose(qc, q, rtol=1e-9, atol=1e-10, err_msg=f"{m['task']} qpos trial {trial} step {s}")
                np.testing.assert_allclose(vc, v, rtol=1e-8, atol=1e-8)
                np.testing.assert_allclose(obc, ob, rtol=1e-8, atol=1e-8)
                assert abs(rc.value - r) < 1e-6 * max(1.0, abs(r)) and bool(dc.value) == bool(d)
                dones.append(bool(d))
        assert any(dones) and not all(dones)
        cs = C.c_double()
        assert lib.orc_spatial_bench(C.byref(ms), 4, 5, 1000, 1, C.byref(cs)) > 0 and np.isfinite(cs.value)
