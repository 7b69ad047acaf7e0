"""GPU suite: the 3-D stepper (k_env3d_step / k_env3d_reset, Ant-v2 and Humanoid-v2) through the C ABI against
oracle/spatial_env.py (float64), plus the fused rollout step writing 376-wide Humanoid records into the replay ring."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(ctx, name, n, seed=3):
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    return HipVectorEnv(name, n, seed=seed, ctx=ctx)


def _spread(env, rng):
    m, n = env.model, env.env_num
    q, v = env.get_state()
    q[:, 2] += np.where(np.arange(n) % 3 == 0, rng.uniform(0.3, 1.0, n), rng.uniform(-0.35, 0.2, n))
    q[:, 3:7] += rng.normal(0, 0.3, (n, 4)); q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    q[:, 7:] += rng.uniform(-0.8, 0.8, (n, m["nq"] - 7))
    v += rng.normal(0, 1.5, v.shape)
    return q, v


@pytest.mark.parametrize("name", ["ant", "humanoid"])
def test_reset_and_step_match_oracle(ctx, name):
    from oracle.spatial_env import SpatialOracle
    n = 48
    env = _mk(ctx, name, n)
    m = env.model
    P = SpatialOracle(m)
    assert (env.obs_dim, env.act_dim, env.nq, env.nv) == (m["obs_dim"], m["act_dim"], m["nq"], m["nv"])
    rng = np.random.default_rng(7)
    obs0 = env.reset()
    q, v = env.get_state()
    # reset (humanoid.py:62-73, ant.py:36-43): init + U(+-c) on qpos, unit quaternion; qvel U(+-c) or 0.1 * randn
    nz = m["reset_noise"]
    init = np.asarray(m["init_qpos"])
    assert np.all(np.abs(q[:, :3] - init[:3]) <= nz) and np.all(np.abs(q[:, 7:] - init[7:]) <= nz)
    np.testing.assert_allclose(np.linalg.norm(q[:, 3:7], axis=1), 1.0, atol=1e-14)
    if m["reset_noise_vel_std"] > 0:
        assert 0.07 < v.std() < 0.13 and abs(v.mean()) < 0.02 and np.abs(v).max() > 0.2
    else:
        assert np.all(np.abs(v) <= nz) and v.std() > 0.3 * nz
    assert len(np.unique(q[:, 2])) == n
    for i in range(0, n, 7):
        np.testing.assert_allclose(obs0[i], P.obs(q[i], v[i], np.zeros(m["act_dim"])), rtol=1e-5, atol=1e-5)
    q, v = _spread(env, rng)
    env.set_state(q, v)
    dones = []
    for it in range(2):
        act = rng.uniform(-1.3, 1.3, (n, env.act_dim)).astype(np.float32)
        obs, rew, done, info = env.step(act)
        q1, v1 = env.get_state()
        for i in range(n):
            qo, vo, oo, ro, do = P.step(q[i].copy(), v[i].copy(), act[i])
            np.testing.assert_allclose(q1[i], qo, rtol=1e-8, atol=1e-8, err_msg=f"qpos env {i} it {it}")
            np.testing.assert_allclose(v1[i], vo, rtol=1e-7, atol=1e-6, err_msg=f"qvel env {i} it {it}")
            np.testing.assert_allclose(obs[i], oo, rtol=1e-5, atol=2e-5, err_msg=f"obs env {i} it {it}")
            np.testing.assert_allclose(rew[i], ro, rtol=1e-5, atol=1e-4)
            assert bool(done[i]) == bool(do), (i, it)
            dones.append(bool(do))
        q, v = q1, v1
    assert any(dones) and not all(dones)
    env.close()


@pytest.mark.parametrize("name", ["ant", "humanoid"])
def test_step3d_is_lane_independent(ctx, name):
    """Same states in different lanes / workgroups -> bit-identical next states (the [slot][env] scratch carries no cross-lane or
    stale dependence), over steps that enter and leave contact."""
    n_base, copies = 64, 5
    n = n_base * copies
    env = _mk(ctx, name, n, seed=11)
    rng = np.random.default_rng(5)
    env.reset()
    q, v = _spread(env, rng)
    qb, vb = q[:n_base].copy(), v[:n_base].copy()
    perm = [rng.permutation(n_base) for _ in range(copies)]
    idx = np.concatenate(perm)
    env.set_state(qb[idx], vb[idx])
    for it in range(4):
        ab = rng.uniform(-1.2, 1.2, (n_base, env.act_dim)).astype(np.float32)
        env.step(ab[idx])
        q1, v1 = env.get_state()
        assert np.isfinite(q1).all() and np.isfinite(v1).all()
        ref_q, ref_v = np.empty_like(qb), np.empty_like(vb)
        ref_q[perm[0]], ref_v[perm[0]] = q1[:n_base], v1[:n_base]
        np.testing.assert_array_equal(q1, ref_q[idx], err_msg=f"it {it}")
        np.testing.assert_array_equal(v1, ref_v[idx], err_msg=f"it {it}")
    env.close()


def test_humanoid_rollout_fills_replay_and_random_return(ctx):
    """The fused rollout on Humanoid: 376-wide records, auto-reset, episode statistics.  Uniform-random actions: the engine's
    known answer (return ~113 over ~22 steps per episode — alive bonus 5 per step dominates, as in MuJoCo)."""
    import ilswiss_amd as ia
    n = 512
    env = _mk(ctx, "humanoid", n, seed=5)
    rb = ia.SimpleReplayBuffer(8 * n, 376, 17, ctx=ctx)
    steps = 60
    for t in range(steps):
        env.rollout_step(policy=None, replay=rb, max_path_length=1000, random_actions=True)
    assert rb._size == 8 * n and rb._top == (steps * n) % (8 * n)
    episodes, ret_sum = env.rollout_stats()
    assert episodes >= n
    mean_ret = ret_sum / episodes
    assert 80.0 < mean_ret < 150.0, mean_ret
    b = rb._get_batch_using_indices(np.arange(8 * n))
    assert np.isfinite(b["observations"]).all() and np.isfinite(b["next_observations"]).all()
    term = b["terminals"].ravel().astype(bool)
    z = b["next_observations"][:, 0]
    assert np.array_equal(term, (z < 1.0) | (z > 2.0)) and 0 < term.sum() < term.size      # humanoid.py:47-48
    # 4.6 < reward < 5.4 + progress: alive bonus 5 minus ctrl cost 0.1 * |0.4 a|^2 <= 0.272, plus 0.25 * com velocity
    assert np.abs(b["rewards"] - 5.0).max() < 3.0
    # consecutive records of one env chain: next_obs(t) == obs(t+1) unless the episode ended
    o = b["observations"].reshape(8, n, 376); no = b["next_observations"].reshape(8, n, 376)
    top = rb._top // n
    for k in range(7):
        a, c = (top + k) % 8, (top + k + 1) % 8
        keep = ~term.reshape(8, n)[a]
        np.testing.assert_array_equal(no[a][keep], o[c][keep])
    env.close()


def test_humanoid_sac_trains_from_device_rollout(ctx):
    """Config 5's shapes end to end: obs 376 / act 17 / 256x256 SAC, fused rollout -> ring -> train_from_replay."""
    import ilswiss_amd as ia
    n = 256
    env = _mk(ctx, "humanoid", n, seed=2)
    rb = ia.SimpleReplayBuffer(64 * n, 376, 17, ctx=ctx)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([256, 256], 376, 17, ctx=ctx, seed=1)
    qf1, qf2 = ia.FlattenMlp([256, 256], 1, 393, ctx=ctx, seed=2), ia.FlattenMlp([256, 256], 1, 393, ctx=ctx, seed=3)
    tr = ia.SoftActorCritic(pol, qf1, qf2, reward_scale=1.0, env=env.single_env_view())
    for t in range(8):
        env.rollout_step(policy=None if t < 4 else pol, replay=rb, max_path_length=1000, random_actions=t < 4)
    tr.train_from_replay(rb, 20, 256)
    vals = [float(np.mean(v)) for v in tr.get_eval_statistics().values()]
    assert vals and np.isfinite(vals).all()
    obs = env.reset()
    a = pol.get_actions(obs.astype(np.float32), deterministic=True)
    assert a.shape == (n, 17) and np.isfinite(a).all() and np.abs(a).max() <= 1.0
    env.close()
