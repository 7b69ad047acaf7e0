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
        assert np.abs(res["grad"] - ref).max() <= 2e-3 * np.abs(ref).max()
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
