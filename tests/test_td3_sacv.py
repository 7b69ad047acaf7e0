"""TD3 (td3.py:72-124) and SAC with a V-function (sac/sac.py:70-179).
CPU: oracles vs the golden vectors produced by the reference; GPU: HIP path vs fixtures and oracle."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.sac_v import SacVOracle
from oracle.td3 import TD3Oracle

TD3_KW = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, policy_and_target_update_period=2,
              soft_target_tau=0.005)
SACV_KW = dict(reward_scale=1.0, discount=0.99, alpha=0.2, policy_lr=3e-4, qf_lr=3e-4, vf_lr=3e-4, soft_target_tau=0.005,
               policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9)
KEYS = ("observations", "actions", "rewards", "terminals", "next_observations")


def _batch(g, s):
    return {k: g[f"s{s}_{k}"] for k in KEYS}


def _dims(g):
    o, a, B, steps = [int(v) for v in g["dims"][:4]]
    return o, a, B, steps, [int(v) for v in g["dims"][4:]]


@pytest.mark.parametrize("tag,out_act", [("g6_td3", "tanh"), ("g6b_td3_identity", "identity")])
def test_oracle_td3_golden(tag, out_act):
    g = load_golden(tag)
    o, a, B, steps, hid = _dims(g)
    orc = TD3Oracle(o, a, hid, g["pi0"], g["q10"], g["q20"], policy_noise=0.2, policy_noise_clip=0.5, output_activation=out_act, **TD3_KW)
    for s in range(steps):
        res = orc.train_step(_batch(g, s), g[f"s{s}_eps"])
        for k in ("qf1_loss", "qf2_loss", "policy_loss"):
            np.testing.assert_allclose(res[k], g[f"s{s}_{k}"], rtol=2e-4, atol=1e-6)
        assert ("pi_grad" in res) == (s % 2 == 0) == (f"s{s}_grad_pi" in g)          # delayed update, td3.py:109
        for k in ("pi", "q1", "q2", "tpi", "tq1", "tq2"):
            np.testing.assert_allclose(getattr(orc, k), g[f"s{s}_{k}"], rtol=0, atol=5e-5)
    assert np.abs(0.2 * g["s0_eps"]).max() > 0.5               # the noise clip was exercised (policies.py:185)
    np.testing.assert_array_equal(g["s1_pi"], g["s0_pi"])     # the odd step leaves policy and targets alone
    np.testing.assert_array_equal(g["s1_tq1"], g["s0_tq1"])


def test_oracle_sac_v_golden():
    g = load_golden("g5_sac_v")
    o, a, B, steps, hid = _dims(g)
    orc = SacVOracle(o, a, hid, g["pi0"], g["q10"], g["q20"], g["vf0"], **SACV_KW)
    for s in range(steps):
        res = orc.train_step(_batch(g, s), g[f"s{s}_eps"])
        for k in ("qf1_loss", "qf2_loss", "vf_loss", "policy_loss"):
            np.testing.assert_allclose(res[k], g[f"s{s}_{k}"], rtol=2e-4, atol=1e-6)
        for k in ("pi", "q1", "q2", "vf", "tvf"):
            np.testing.assert_allclose(getattr(orc, k), g[f"s{s}_{k}"], rtol=0, atol=5e-5)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag,out_act", [("g6_td3", "tanh"), ("g6b_td3_identity", "identity")])
def test_hip_td3_golden(ctx, tag, out_act):
    """g6: the run script's tanh output; g6b: Mlp's default output activation (identity, networks.py:31) — both from the reference."""
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.td3 import TD3, MlpGaussianNoisePolicy
    g = load_golden(tag)
    o, a, B, steps, hid = _dims(g)
    pol = MlpGaussianNoisePolicy(hid, o, a, policy_noise=0.2, policy_noise_clip=0.5, output_activation=out_act, ctx=ctx, seed=1)
    q1, q2 = FlattenMlp(hid, 1, o + a, ctx=ctx, seed=2), FlattenMlp(hid, 1, o + a, ctx=ctx, seed=3)
    pol.set_flat_params(g["pi0"]); q1.set_flat_params(g["q10"]); q2.set_flat_params(g["q20"])
    tr = TD3(pol, q1, q2, max_batch=B, **TD3_KW)
    for s in range(steps):
        tr.eval_statistics = None
        tr.train_step(_batch(g, s), eps_target=g[f"s{s}_eps"])
        st = tr.get_eval_statistics()
        np.testing.assert_allclose(st["QF1 Loss"], g[f"s{s}_qf1_loss"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(st["QF2 Loss"], g[f"s{s}_qf2_loss"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(st["Policy Loss"], g[f"s{s}_policy_loss"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(st["Q Targets Mean"], g[f"s{s}_q_target_mean"], rtol=2e-4, atol=1e-6)
        for k in ("pi", "q1", "q2", "tpi", "tq1", "tq2"):
            np.testing.assert_allclose(tr.get_flat_params(k), g[f"s{s}_{k}"], rtol=0, atol=5e-5, err_msg=f"step {s} {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("out_act", ["tanh", "identity"])
def test_hip_td3_vs_oracle_h256(ctx, out_act):
    """BASELINE dims (H=256, B=256), 6 steps with Philox-free explicit noise, no statistics requested in between."""
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.td3 import TD3, MlpGaussianNoisePolicy
    from oracle import mlp as omlp
    rng = np.random.default_rng(61)
    o, a, hid, B = 17, 6, [256, 256], 256
    pi0, q10, q20 = omlp.init_mlp(rng, o, hid, a, init_w=1e-3), omlp.init_mlp(rng, o + a, hid, 1), omlp.init_mlp(rng, o + a, hid, 1)
    pi0[-(256 * a + a):] *= 100.0
    orc = TD3Oracle(o, a, hid, pi0, q10, q20, policy_noise=0.2, policy_noise_clip=0.5, output_activation=out_act, **TD3_KW)
    pol = MlpGaussianNoisePolicy(hid, o, a, policy_noise=0.2, policy_noise_clip=0.5, output_activation=out_act, ctx=ctx, seed=1)
    q1, q2 = FlattenMlp(hid, 1, o + a, ctx=ctx, seed=2), FlattenMlp(hid, 1, o + a, ctx=ctx, seed=3)
    pol.set_flat_params(pi0); q1.set_flat_params(q10); q2.set_flat_params(q20)
    tr = TD3(pol, q1, q2, max_batch=B, **TD3_KW)
    for s in range(6):
        batch = dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32),
                     actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                     rewards=rng.normal(0, 1, (B, 1)).astype(np.float32),
                     terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                     next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))
        eps = rng.normal(0, 1, (B, a)).astype(np.float32)
        orc.train_step(batch, eps)
        tr.train_step(batch, eps_target=eps)
    for k in ("pi", "q1", "q2", "tpi", "tq1", "tq2"):
        np.testing.assert_allclose(tr.get_flat_params(k), getattr(orc, k), rtol=0, atol=1e-4, err_msg=k)
    # exploration = the same noisy module (td3_exp_script.py:85): bounded by max_act + clip, noise really there
    obs = rng.normal(0, 1, (64, o)).astype(np.float32)
    det = pol.get_actions(obs, deterministic=True)
    np.testing.assert_allclose(det, orc.policy(orc.pi, obs)[0], rtol=1e-4, atol=2e-5)
    noisy = pol.get_actions(obs)
    assert 1e-3 < np.abs(noisy - det).max() <= 0.5 + 1e-6


@pytest.mark.gpu
def test_hip_sac_v_golden(ctx):
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.sac_v import SoftActorCriticV
    g = load_golden("g5_sac_v")
    o, a, B, steps, hid = _dims(g)
    pol = ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=1)
    q1, q2 = FlattenMlp(hid, 1, o + a, ctx=ctx, seed=2), FlattenMlp(hid, 1, o + a, ctx=ctx, seed=3)
    vf = FlattenMlp(hid, 1, o, ctx=ctx, seed=4)
    for net, k in ((pol, "pi0"), (q1, "q10"), (q2, "q20"), (vf, "vf0")):
        net.set_flat_params(g[k])
    tr = SoftActorCriticV(pol, q1, q2, vf, max_batch=B, **SACV_KW)
    for s in range(steps):
        tr.eval_statistics = None
        tr.train_step(_batch(g, s), eps=g[f"s{s}_eps"])
        st = tr.get_eval_statistics()
        for ref, k in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("VF Loss", "vf_loss"), ("Policy Loss", "policy_loss")):
            np.testing.assert_allclose(st[ref], g[f"s{s}_{k}"], rtol=2e-4, atol=1e-6, err_msg=f"step {s} {ref}")
        for k in ("pi", "q1", "q2", "vf", "tvf"):
            np.testing.assert_allclose(tr.get_flat_params(k), g[f"s{s}_{k}"], rtol=0, atol=5e-5, err_msg=f"step {s} {k}")


@pytest.mark.gpu
def test_hip_sac_v_large_batch_row_split_dw(ctx):
    """B = 2048 rows: row-split weight gradients with Adam (+ the Polyak write of target V) in k_dw_reduce."""
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.sac_v import SoftActorCriticV
    from oracle import mlp as omlp
    rng = np.random.default_rng(88)
    o, a, hid, B = 11, 3, [64, 64], 2048
    pi0 = omlp.init_mlp(rng, o, hid, a, init_w=1e-3, n_heads=2)
    q10, q20, vf0 = omlp.init_mlp(rng, o + a, hid, 1), omlp.init_mlp(rng, o + a, hid, 1), omlp.init_mlp(rng, o, hid, 1)
    orc = SacVOracle(o, a, hid, pi0, q10, q20, vf0, **SACV_KW)
    pol = ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=1)
    q1, q2, vf = FlattenMlp(hid, 1, o + a, ctx=ctx, seed=2), FlattenMlp(hid, 1, o + a, ctx=ctx, seed=3), FlattenMlp(hid, 1, o, ctx=ctx, seed=4)
    for net, p0 in ((pol, pi0), (q1, q10), (q2, q20), (vf, vf0)):
        net.set_flat_params(p0)
    tr = SoftActorCriticV(pol, q1, q2, vf, max_batch=B, **SACV_KW)
    for s in range(3):
        batch = dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32), actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                     rewards=rng.normal(0, 1, (B, 1)).astype(np.float32), terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                     next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))
        eps = rng.normal(0, 1, (B, a)).astype(np.float32)
        orc.train_step(batch, eps)
        tr.train_step(batch, eps=eps)
    for k in ("pi", "q1", "q2", "vf", "tvf"):
        np.testing.assert_allclose(tr.get_flat_params(k), getattr(orc, k), rtol=0, atol=5e-5, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("hid", [[128, 128], [256, 256]])
def test_hip_sac_v_column_split_widths_vs_oracle(ctx, hid):
    """H = 128 / 256 run on the column-split kernels (policy head finished in the consumer's prologue, target V riding
    in the same launch); 4 steps against the oracle, statistics requested on every step."""
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.sac_v import SoftActorCriticV
    from oracle import mlp as omlp
    rng = np.random.default_rng(hid[0])
    o, a, B = 17, 6, 96
    pi0 = omlp.init_mlp(rng, o, hid, a, init_w=1e-3, n_heads=2)
    pi0[-(2 * (hid[-1] * a + a)):] *= 100.0
    q10, q20, vf0 = omlp.init_mlp(rng, o + a, hid, 1), omlp.init_mlp(rng, o + a, hid, 1), omlp.init_mlp(rng, o, hid, 1)
    orc = SacVOracle(o, a, hid, pi0, q10, q20, vf0, **SACV_KW)
    pol = ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=1)
    q1, q2, vf = FlattenMlp(hid, 1, o + a, ctx=ctx, seed=2), FlattenMlp(hid, 1, o + a, ctx=ctx, seed=3), FlattenMlp(hid, 1, o, ctx=ctx, seed=4)
    for net, p0 in ((pol, pi0), (q1, q10), (q2, q20), (vf, vf0)):
        net.set_flat_params(p0)
    tr = SoftActorCriticV(pol, q1, q2, vf, max_batch=128, **SACV_KW)
    for s in range(4):
        batch = dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32), actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                     rewards=rng.normal(0, 1, (B, 1)).astype(np.float32), terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                     next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))
        eps = rng.normal(0, 1, (B, a)).astype(np.float32)
        res = orc.train_step(batch, eps)
        tr.eval_statistics = None
        tr.train_step(batch, eps=eps)
        st = tr.get_eval_statistics()
        for ref, k in (("QF1 Loss", "qf1_loss"), ("VF Loss", "vf_loss"), ("Policy Loss", "policy_loss")):
            np.testing.assert_allclose(st[ref], res[k], rtol=3e-4, atol=2e-6, err_msg=f"step {s} {ref}")
        np.testing.assert_allclose(st["Log Pis Mean"], res["log_pi"].mean(), rtol=1e-4, atol=1e-5)
    for k in ("pi", "q1", "q2", "vf", "tvf"):
        np.testing.assert_allclose(tr.get_flat_params(k), getattr(orc, k), rtol=0, atol=1e-4, err_msg=k)
