"""Behaviour cloning (bc/bc.py:81-106).  CPU: oracle vs the golden vectors produced by the reference; GPU: HIP path."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.bc import BCOracle


def _dims(g, mode):
    o, a, B, steps = [int(v) for v in g[f"{mode}_dims"][:4]]
    return o, a, B, steps, [int(v) for v in g[f"{mode}_dims"][4:]]


@pytest.mark.parametrize("mode", ["MLE", "MSE"])
def test_oracle_bc_golden(mode):
    g = load_golden("g13_bc")
    o, a, B, steps, hid = _dims(g, mode)
    orc = BCOracle(o, a, hid, g[f"{mode}_pi0"], mode=mode, lr=1e-3, momentum=0.5)
    for s in range(steps):
        res = orc.update(g[f"{mode}_s{s}_obs"], g[f"{mode}_s{s}_acts"], g[f"{mode}_s{s}_eps"])
        np.testing.assert_allclose(res["stat"], g[f"{mode}_s{s}_stat"], rtol=2e-4, atol=1e-5)
        ref = g[f"{mode}_s{s}_grad"]
        assert np.abs(res["grad"] - ref).max() <= 1e-4 * np.abs(ref).max()
        np.testing.assert_allclose(orc.pi, g[f"{mode}_s{s}_pi"], rtol=0, atol=5e-5)
    assert g["MLE_s0_acts"].max() > 0.99999          # the near-saturated expert action is in the fixture


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["MLE", "MSE"])
def test_hip_bc_golden(ctx, mode):
    import ilswiss_amd as ia
    from ilswiss_amd.bc import BC
    g = load_golden("g13_bc")
    o, a, B, steps, hid = _dims(g, mode)
    pol = ia.ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=1)
    pol.set_flat_params(g[f"{mode}_pi0"])
    tr = BC(mode, pol, batch_size=B, lr=1e-3, momentum=0.5)
    for s in range(steps):
        tr.end_epoch()
        tr.train_step(dict(observations=g[f"{mode}_s{s}_obs"], actions=g[f"{mode}_s{s}_acts"]), eps=g[f"{mode}_s{s}_eps"])
        key = "Log-Likelihood" if mode == "MLE" else "MSE"
        np.testing.assert_allclose(tr.get_eval_statistics()[key], g[f"{mode}_s{s}_stat"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(pol.get_flat_params(), g[f"{mode}_s{s}_pi"], rtol=0, atol=5e-5, err_msg=f"step {s}")


@pytest.mark.gpu
def test_hip_bc_clones_an_expert_from_the_replay(ctx):
    """_do_training on device-sampled expert batches: the clone's deterministic actions approach the expert's."""
    import ilswiss_amd as ia
    from ilswiss_amd.bc import BC
    from ilswiss_amd.replay import SimpleReplayBuffer
    rng = np.random.default_rng(3)
    o, a, N = 11, 3, 4096
    W = rng.normal(0, 0.5, (o, a)).astype(np.float32)
    obs = rng.normal(0, 1, (N, o)).astype(np.float32)
    act = np.tanh(obs @ W).astype(np.float32)
    rb = SimpleReplayBuffer(N, o, a, random_seed=1, ctx=ctx)
    rb.add_rows(obs, act, np.zeros(N, np.float32), np.zeros(N, bool), obs)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], o, a, ctx=ctx, seed=2)
    tr = BC("MLE", pol, expert_replay_buffer=rb, batch_size=256, lr=1e-3, num_updates_per_train_call=300)
    err0 = np.abs(pol.get_actions(obs[:512], deterministic=True) - act[:512]).mean()
    tr.train_from_replay()
    ll0 = tr.get_eval_statistics()["Log-Likelihood"]
    for _ in range(4):
        tr.end_epoch()
        tr.train_from_replay()
    err1 = np.abs(pol.get_actions(obs[:512], deterministic=True) - act[:512]).mean()
    assert tr.get_eval_statistics()["Log-Likelihood"] > ll0 + 1.0 and err1 < 0.35 * err0, (ll0, tr.get_eval_statistics(), err0, err1)


@pytest.mark.gpu
def test_hip_dagger_relabels_rollouts_with_the_expert(ctx):
    """dagger.py:27-71: demos copied into the replay buffer; the sampling loop stores the EXPERT's action for the observations
    the learner visits; after training on those labels the learner imitates the expert on its own state distribution."""
    import ilswiss_amd as ia
    from ilswiss_amd.bc import DAgger
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from ilswiss_amd.replay import SimpleReplayBuffer
    o, a = 11, 3
    expert = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], o, a, ctx=ctx, seed=11)
    flat = expert.get_flat_params()
    flat[-(2 * (64 * a + a)):-(64 * a + a)] *= 300.0          # a mean head with real structure
    expert.set_flat_params(flat)
    learner = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], o, a, ctx=ctx, seed=12)
    env = HipVectorEnv("hopper", 256, seed=4, ctx=ctx)
    rng = np.random.default_rng(0)
    demo_obs = rng.normal(0, 1, (500, o)).astype(np.float32)
    exp_rb = SimpleReplayBuffer(1000, o, a, random_seed=1, ctx=ctx)
    exp_rb.add_rows(demo_obs, expert.get_actions(demo_obs, deterministic=True), np.zeros(500, np.float32), np.zeros(500, bool), demo_obs)
    rb = SimpleReplayBuffer(100000, o, a, random_seed=2, ctx=ctx)
    tr = DAgger(ia.MakeDeterministic(expert), "MLE", learner, exp_rb, rb, num_initial_train_steps=50, batch_size=256, lr=1e-3,
                num_updates_per_train_call=100)
    assert rb.num_steps_can_sample() == 500                      # the demonstrations were copied in (dagger.py:27-35)
    for _ in range(20):
        env.rollout_step(policy=learner, replay=rb, max_path_length=200, label_policy=tr.expert_policy)
    assert rb.num_steps_can_sample() == 500 + 20 * 256
    batch = rb._gather(np.arange(500, 500 + 20 * 256))
    np.testing.assert_allclose(batch["actions"], expert.get_actions(batch["observations"], deterministic=True), rtol=1e-5, atol=1e-6)
    err0 = np.abs(learner.get_actions(batch["observations"], deterministic=True) - batch["actions"]).mean()
    for _ in range(6):
        tr.end_epoch()
        tr.train_from_replay(rb)
    err1 = np.abs(learner.get_actions(batch["observations"], deterministic=True) - batch["actions"]).mean()
    assert err1 < 0.4 * err0, (err0, err1)
