"""Hindsight Experience Replay: the relabelling buffer and the oracle against reference-generated vectors (CPU), the
device trainers against the same vectors and the loop on a stand-in goal env (GPU).  Fixtures: tools/make_golden.py her."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cat(g, s):
    b = {k: g[f"s{s}_{k}"] for k in ("observations", "actions", "rewards", "terminals", "next_observations", "desired_goals", "next_desired_goals")}
    return b, dict(b, observations=np.concatenate([b["observations"], b["desired_goals"]], 1),
                   next_observations=np.concatenate([b["next_observations"], b["next_desired_goals"]], 1))


def test_her_td3_oracle_matches_reference():
    from oracle.td3 import TD3Oracle
    g = np.load(os.path.join(G, "g20_her_td3.npz"))
    o, gd, a, B, steps, h1, h2 = (int(v) for v in g["dims"])
    orc = TD3Oracle(o + gd, a, [h1, h2], g["pi0"], g["q10"], g["q20"], policy_noise=float(g["sigma"]), policy_noise_clip=0.0, her=True,
                    reward_scale=1.0, discount=0.9, policy_lr=3e-4, qf_lr=3e-4, policy_and_target_update_period=2, soft_target_tau=0.005)
    assert np.allclose([orc.clip_l, orc.clip_r], g["clip"])
    for s in range(steps):
        res = orc.train_step(_cat(g, s)[1], g[f"s{s}_eps"])
        for k in ("qf1_loss", "qf2_loss", "policy_loss"):
            np.testing.assert_allclose(res[k], g[f"s{s}_{k}"], rtol=2e-4, atol=1e-5)
        for k in ("pi", "q1", "q2", "tpi", "tq1"):
            np.testing.assert_allclose(getattr(orc, k), g[f"s{s}_{k}"], atol=5e-5)


def test_hindsight_replay_buffer_golden():
    """Same scripted paths, same seeds -> the reference's relabelled batches, key for key (future / final / no relabelling)."""
    from ilswiss_amd.her import Box, DictSpace, HindsightReplayBuffer
    g = np.load(os.path.join(G, "g22_her_buffer.npz"))
    o, gd, a, cap = (int(v) for v in g["dims"])

    class Env:
        observation_space = DictSpace(observation=Box(-np.ones(o), np.ones(o)), desired_goal=Box(-np.ones(gd), np.ones(gd)),
                                      achieved_goal=Box(-np.ones(gd), np.ones(gd)))
        action_space = Box(-np.ones(a), np.ones(a))

        @staticmethod
        def compute_reward(ag, dg, info=None):
            return -(np.linalg.norm(ag - dg, axis=-1) > 0.5).astype(np.float32)
    for ci, (rtype, ratio) in enumerate((("future", 0.8), ("final", 0.8), ("future", 0.0))):
        rb = HindsightReplayBuffer(cap, Env, random_seed=77, relabel_type=rtype, her_ratio=ratio)
        for p in range(6):
            obs, dg, ag = g[f"p{p}_obs"], g[f"p{p}_dg"], g[f"p{p}_ag"]
            L = len(g[f"p{p}_act"])
            d = lambda i: dict(observation=obs[i], desired_goal=dg[i], achieved_goal=ag[i])   # noqa: E731
            for i in range(L):
                rb.add_sample(d(i), g[f"p{p}_act"][i], g[f"p{p}_rew"][i], bool(g[f"p{p}_term"][i]), d(i + 1))
            rb.terminate_episode()
        np.testing.assert_array_equal(np.array(sorted(rb._traj_endpoints.items())), g[f"c{ci}_endpoints"])
        np.random.seed(500 + ci)
        bt = rb.random_batch(12)
        for k in ("observations", "actions", "rewards", "terminals", "next_observations", "achieved_goals", "desired_goals",
                  "next_achieved_goals", "next_desired_goals"):
            np.testing.assert_array_equal(np.asarray(bt[k]), g[f"c{ci}_{k}"], err_msg=f"{rtype} {ratio} {k}")


def _nets(ia, ctx, g, o, gd, a, hid, two_heads):
    q1, q2 = ia.FlattenMlp(hid, 1, o + gd + a, ctx=ctx), ia.FlattenMlp(hid, 1, o + gd + a, ctx=ctx)
    q1.set_flat_params(g["q10"]), q2.set_flat_params(g["q20"])
    return q1, q2


@pytest.mark.gpu
def test_hip_her_td3_golden(ctx):
    import ilswiss_amd as ia
    from ilswiss_amd import her
    g = np.load(os.path.join(G, "g20_her_td3.npz"))
    o, gd, a, B, steps, h1, h2 = (int(v) for v in g["dims"])
    pol = her.MlpGaussianAndEpsilonPolicy([h1, h2], o, a, condition_dim=gd, max_sigma=float(g["sigma"]), min_sigma=float(g["sigma"]), output_activation="tanh", ctx=ctx)
    pol.set_flat_params(g["pi0"])
    q1, q2 = _nets(ia, ctx, g, o, gd, a, [h1, h2], False)
    tr = her.TD3(pol, q1, q2, reward_scale=1.0, discount=0.9, policy_lr=3e-4, qf_lr=3e-4, policy_and_target_update_period=2,
                 soft_target_tau=0.005, max_batch=B)
    np.testing.assert_allclose([tr.clip_return_l, tr.clip_return_r], g["clip"], rtol=1e-6)
    for s in range(steps):
        tr.eval_statistics = None
        tr.train_step(_cat(g, s)[0], g[f"s{s}_eps"])
        st = tr.get_eval_statistics()
        for k_ref, k in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss")):
            np.testing.assert_allclose(st[k_ref], g[f"s{s}_{k}"], rtol=5e-4, atol=2e-5, err_msg=f"step {s} {k}")
        np.testing.assert_allclose(st["Q Targets Mean"], g[f"s{s}_q_target_mean"], rtol=1e-4, atol=1e-4)
        for k, nm in (("policy", "pi"), ("qf1", "q1"), ("qf2", "q2"), ("target_policy", "tpi"), ("target_qf1", "tq1")):
            np.testing.assert_allclose(tr.get_flat_params(k), g[f"s{s}_{nm}"], atol=5e-5, err_msg=f"step {s} {k}")


@pytest.mark.gpu
def test_hip_her_sac_golden(ctx):
    import ilswiss_amd as ia
    from ilswiss_amd import her
    g = np.load(os.path.join(G, "g21_her_sac.npz"))
    o, gd, a, B, steps, h1, h2 = (int(v) for v in g["dims"])
    pol = ia.ReparamTanhMultivariateGaussianPolicy([h1, h2], o + gd, a, ctx=ctx)
    pol.set_flat_params(g["pi0"])
    q1, q2 = _nets(ia, ctx, g, o, gd, a, [h1, h2], True)
    tr = her.SAC(pol, q1, q2, reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, alpha_lr=3e-4, soft_target_tau=0.005, alpha=0.2,
                 train_alpha=True, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9, max_batch=B)
    assert tr.target_entropy == -a
    for s in range(steps):
        tr.eval_statistics = None
        tr.train_step(_cat(g, s)[0], g[f"s{s}_eps_next"], g[f"s{s}_eps_cur"])
        st = tr.get_eval_statistics()
        for k_ref, k in (("QF1 Loss", "qf1_loss"), ("Policy Loss", "policy_loss"), ("Alpha Loss", "alpha_loss")):
            np.testing.assert_allclose(st[k_ref], g[f"s{s}_{k}"], rtol=5e-4, atol=2e-5, err_msg=f"step {s} {k}")
        np.testing.assert_allclose(tr.get_params("policy"), g[f"s{s}_pi"], atol=5e-5)
        np.testing.assert_allclose(tr.get_params("qf1"), g[f"s{s}_q1"], atol=5e-5)
        np.testing.assert_allclose(tr.get_params("target_qf1"), g[f"s{s}_tq1"], atol=5e-5)


@pytest.mark.gpu
def test_her_loop_learns_to_reach_on_the_stand_in_env():
    """her.HER end to end (exploration policy -> HindsightReplayBuffer -> her.TD3 on the device) on PointReachEnv (NOT a reference env).
    Own context: the device's noise streams then do not depend on which tests ran before."""
    import ilswiss_amd as ia
    from ilswiss_amd import her
    ctx = ia.Context(0, seed=11)
    np.random.seed(3)
    env = her.PointReachEnv(seed=1)
    pol = her.MlpGaussianAndEpsilonPolicy([64, 64], 4, 2, action_space=env.action_space, condition_dim=2, output_activation="tanh", ctx=ctx, seed=5)
    q1, q2 = ia.FlattenMlp([64, 64], 1, 8, ctx=ctx, seed=6), ia.FlattenMlp([64, 64], 1, 8, ctx=ctx, seed=7)
    tr = her.TD3(pol, q1, q2, discount=0.95, policy_lr=1e-3, qf_lr=1e-3, max_batch=128)
    tr.eval_statistics = {}
    alg = her.HER(tr, env, pol, num_epochs=4, num_steps_per_epoch=1000, min_steps_before_training=500, max_path_length=25, batch_size=128,
                  replay_buffer_size=20000, num_steps_per_eval=500)
    first = alg.evaluate()
    hist = alg.train()
    assert alg._n_train_steps_total >= 3000 and max(hist) >= max(0.6, first + 0.3), (first, hist)
    ctx.close()


@pytest.mark.gpu
def test_device_hindsight_buffer_golden(ctx):
    """g22 on the device-resident buffer: the same scripted paths and seeds -> the reference's relabelled batches out of ONE gather kernel
    (ilsx_her_gather) over the HBM ring; the ring stores fp32 (the reference's host arrays are float64: 1e-6 relative on the values,
    rewards and terminals exact)."""
    from ilswiss_amd.her import Box, DeviceHindsightReplayBuffer, DictSpace
    g = np.load(os.path.join(G, "g22_her_buffer.npz"))
    o, gd, a, cap = (int(v) for v in g["dims"])

    class Env:
        observation_space = DictSpace(observation=Box(-np.ones(o), np.ones(o)), desired_goal=Box(-np.ones(gd), np.ones(gd)),
                                      achieved_goal=Box(-np.ones(gd), np.ones(gd)))
        action_space = Box(-np.ones(a), np.ones(a))
        reward_type, distance_threshold = "sparse", 0.5          # the rule tools/make_golden.py's compute_reward implements
    for ci, (rtype, ratio) in enumerate((("future", 0.8), ("final", 0.8), ("future", 0.0))):
        rb = DeviceHindsightReplayBuffer(cap, Env, random_seed=77, relabel_type=rtype, her_ratio=ratio, ctx=ctx)
        for p in range(6):
            obs, dg, ag = g[f"p{p}_obs"], g[f"p{p}_dg"], g[f"p{p}_ag"]
            L = len(g[f"p{p}_act"])
            d = lambda i: dict(observation=obs[i], desired_goal=dg[i], achieved_goal=ag[i])   # noqa: E731
            for i in range(L):
                rb.add_sample(d(i), g[f"p{p}_act"][i], g[f"p{p}_rew"][i], bool(g[f"p{p}_term"][i]), d(i + 1))
            rb.terminate_episode()
        np.testing.assert_array_equal(np.array(sorted(rb._traj_endpoints.items())), g[f"c{ci}_endpoints"])
        np.random.seed(500 + ci)
        dev = rb.random_batch(12)
        assert dev["_her_cat"] and dev["observations"].shape == (12, o + gd)      # device arrays, already observation | goal
        bt = rb.numpy_batch(dev)
        for k in ("observations", "actions", "next_observations", "desired_goals", "next_desired_goals"):
            np.testing.assert_allclose(bt[k], g[f"c{ci}_{k}"], rtol=1e-6, atol=1e-7, err_msg=f"{rtype} {ratio} {k}")
        np.testing.assert_array_equal(bt["terminals"], g[f"c{ci}_terminals"], err_msg=f"{rtype} {ratio}")
        np.testing.assert_allclose(bt["rewards"], g[f"c{ci}_rewards"], rtol=1e-6, atol=1e-7, err_msg=f"{rtype} {ratio} rewards")


@pytest.mark.gpu
def test_her_loop_uses_the_device_buffer_by_default():
    import ilswiss_amd as ia
    from ilswiss_amd import her
    c = ia.Context(0, seed=12)
    try:
        env = her.PointReachEnv(seed=2)
        pol = her.MlpGaussianAndEpsilonPolicy([64, 64], 4, 2, action_space=env.action_space, condition_dim=2, output_activation="tanh", ctx=c, seed=5)
        q1, q2 = ia.FlattenMlp([64, 64], 1, 8, ctx=c, seed=6), ia.FlattenMlp([64, 64], 1, 8, ctx=c, seed=7)
        tr = her.TD3(pol, q1, q2, discount=0.95, policy_lr=1e-3, qf_lr=1e-3, max_batch=64)
        alg = her.HER(tr, env, pol, num_epochs=1, num_steps_per_epoch=300, min_steps_before_training=100, max_path_length=25, batch_size=64,
                      replay_buffer_size=5000, num_steps_per_eval=50)
        assert isinstance(alg.replay_buffer, her.DeviceHindsightReplayBuffer)
        alg.train()
        assert alg._n_train_steps_total > 100 and np.isfinite(tr.get_flat_params("policy")).all()
        b = alg.replay_buffer.numpy_batch(alg.replay_buffer.random_batch(64))
        # relabelled rows (the first 80 %) mostly succeed by construction; every reward is the sparse rule of the env
        d = np.linalg.norm(b["next_observations"][:, :2] - b["desired_goals"], axis=1)
        np.testing.assert_array_equal(b["rewards"][:, 0], -(d > env.tol).astype(np.float32))
    finally:
        c.close()


def test_device_reward_rule_is_verified_against_the_envs_own_compute_reward():
    """the device-side relabel recomputes rewards with ONE rule (gym's Fetch rule on the goal distance); the reference calls
    env.compute_reward (relabel_replay_buffer.py:37-40).  The loop may only pick the device buffer when the two agree on a probe batch:
    an env that merely EXPOSES distance_threshold (gym's HandManipulate*: also a rotation threshold), has another reward_type, or
    throws, gets the host buffer; unknown reward types are refused by the device buffer itself, not mapped to 'dense'."""
    from ilswiss_amd import her

    class Fetchish(her.PointReachEnv):
        distance_threshold, reward_type = 0.05, "sparse"

        def compute_reward(self, ag, dg, info=None):
            return -(np.linalg.norm(np.asarray(ag) - np.asarray(dg), axis=-1) > self.distance_threshold).astype(np.float32)

    class Dense(Fetchish):
        reward_type = "dense"

        def compute_reward(self, ag, dg, info=None):
            return -np.linalg.norm(np.asarray(ag) - np.asarray(dg), axis=-1).astype(np.float32)

    class HandLike(Fetchish):          # distance AND "rotation" (here: the sign pattern) must match
        def compute_reward(self, ag, dg, info=None):
            ok = (np.linalg.norm(np.asarray(ag) - np.asarray(dg), axis=-1) <= self.distance_threshold) & (np.sign(ag[..., 0]) == np.sign(dg[..., 0]))
            return ok.astype(np.float32) - 1.0

    class Shaped(Fetchish):
        reward_type = "shaped"

    class Broken(Fetchish):
        def compute_reward(self, ag, dg, info=None):
            raise RuntimeError("needs the simulator")

    assert her.device_reward_rule_matches(her.PointReachEnv(), 2)          # the stand-in env: tol = 0.1, sparse
    assert her.device_reward_rule_matches(Fetchish(), 2) and her.device_reward_rule_matches(Dense(), 2)
    assert her.device_reward_rule(Dense()) == (1, 0.05) and her.device_reward_rule(Fetchish()) == (0, 0.05)
    assert not her.device_reward_rule_matches(HandLike(), 2)
    assert not her.device_reward_rule_matches(Shaped(), 2) and not her.device_reward_rule_matches(Broken(), 2)
    with pytest.raises(NotImplementedError):
        her.device_reward_rule(Shaped())

    class NoThr:
        reward_type = "sparse"
    with pytest.raises(NotImplementedError):
        her.device_reward_rule(NoThr())

    class Tr:          # a trainer without a ctx attribute: nothing to put a device buffer on either
        def end_epoch(self):
            pass
    loop = her.HER(Tr(), HandLike(), None)
    assert type(loop.replay_buffer) is her.HindsightReplayBuffer
