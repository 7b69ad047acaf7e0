"""CPU suite: the PPO oracle (GAE + clipped-surrogate minibatch updates) against the golden vectors produced
by the reference (ppo.py:57-170)."""
import numpy as np

from conftest import load_golden
from oracle.ppo import PPOOracle, gae_one_traj

KW = dict(reward_scale=1.0, discount=0.99, clip_eps=0.2, policy_lr=3e-4, value_lr=3e-4, gae_tau=0.95,
          value_l2_reg=1e-3, mini_batch_size=16, update_epoch=2)


def _trajs(g):
    return [dict(observations=g[f"t{i}_observations"], actions=g[f"t{i}_actions"], rewards=g[f"t{i}_rewards"])
            for i in range(len(g["lens"]))]


def test_gae_and_fixed_log_probs():
    g = load_golden("g7_ppo")
    o, a = int(g["dims"][0]), int(g["dims"][1])
    orc = PPOOracle(o, a, [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **KW)
    obs, act, R, A, V = orc.calc_adv(_trajs(g))
    np.testing.assert_allclose(V, g["values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(R, g["returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(A, g["advantages"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(orc.log_prob(obs, act)[0], g["fixed_log_probs"], rtol=1e-5, atol=1e-5)
    # per-trajectory standardisation with the unbiased std (ppo.py:86): each segment has mean 0, std(ddof=1) 1
    off = 0
    for L in g["lens"]:
        seg = A[off:off + L]
        assert abs(seg.mean()) < 1e-5 and abs(seg.std(ddof=1) - 1) < 1e-4
        off += L


def test_gae_zero_bootstrap_closed_form():
    v = np.array([[1.0], [2.0], [3.0]], np.float32)
    r = np.array([[0.5], [0.5], [0.5]], np.float32)
    R, _, adv = gae_one_traj(v, r, 0.9, 0.8)
    d2 = 0.5 + 0.9 * 0 - 3.0
    d1 = 0.5 + 0.9 * 3.0 - 2.0
    d0 = 0.5 + 0.9 * 2.0 - 1.0
    a2 = d2; a1 = d1 + 0.72 * a2; a0 = d0 + 0.72 * a1
    np.testing.assert_allclose(adv.ravel(), [a0, a1, a2], rtol=1e-6)
    np.testing.assert_allclose(R.ravel(), [1 + a0, 2 + a1, 3 + a2], rtol=1e-6)


import pytest


def _kw_of(g):
    """the ppo_params a fixture was generated with: KW + its epochs + whatever the generator overrode (kw_* arrays)"""
    kw = dict(KW, update_epoch=int(g["epochs"]))
    for k in g:
        if k.startswith("kw_"):
            v = np.asarray(g[k])
            kw[k[3:]] = bool(v) if v.dtype == bool else float(v)
    return kw


# g7b: clip_grad_norm_(20) active; g7c: use_value_clip=True (ppo.py:137-143), clip_eps 0.1, value_lr 3e-3 (rows cross the clip window)
@pytest.mark.parametrize("name", ["g7_ppo", "g7b_ppo_clip", "g7c_ppo_vclip", "g7d_ppo_condstd", "g7e_ppo_unequal"])
def test_train_step(name):
    g = load_golden(name)
    o, a = int(g["dims"][0]), int(g["dims"][1])
    orc = PPOOracle(o, a, [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **_kw_of(g))
    res = orc.train_step(_trajs(g), list(g["perms"]))
    assert (res["pi_grad_norm"] > 20) == (name == "g7b_ppo_clip")
    if name == "g7c_ppo_vclip":
        assert orc.use_value_clip and int(g["n_outside_clip"]) > 0
        # the clipped branch matters: the same run without it ends somewhere else
        plain = PPOOracle(o, a, [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **dict(_kw_of(g), use_value_clip=False))
        plain.train_step(_trajs(g), list(g["perms"]))
        assert np.abs(plain.vf - g["vf_final"]).max() > 1e-3
    np.testing.assert_allclose(orc.vf, g["vf_final"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(orc.pi, g["pi_final"], rtol=0, atol=5e-5)
    assert np.abs(orc.pi - g["pi0"]).max() > 1e-4  # the policy really moved


# ------------------------------------------------------------------------------------------------ GPU
def _hip_ppo(ctx, g, **over):
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
    o, a = int(g["dims"][0]), int(g["dims"][1])
    hid = [int(v) for v in g["dims"][2:]]
    kw = dict(_kw_of(g), **over)
    pol = ReparamMultivariateGaussianPolicy(hid, o, a, conditioned_std=bool(kw.pop("conditioned_std", False)), hidden_activation="tanh", ctx=ctx, seed=3)
    vf = FlattenMlp(hid, 1, o, hidden_activation="tanh", ctx=ctx, seed=4)
    tr = PPO(pol, vf, max_samples=4096, **kw)
    tr.set_flat_params(g["pi0"], g["vf0"])
    return tr


@pytest.mark.gpu
def test_hip_gae_and_fixed_log_probs_golden(ctx):
    g = load_golden("g7_ppo")
    tr = _hip_ppo(ctx, g)
    np.testing.assert_array_equal(tr.get_flat_params(0), g["pi0"])
    np.testing.assert_array_equal(tr.get_flat_params(1), g["vf0"])
    R, A, V, lp = tr.calc_adv(_trajs(g))
    np.testing.assert_allclose(V, g["values"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(R, g["returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(A, g["advantages"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(lp, g["fixed_log_probs"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g7_ppo", "g7b_ppo_clip", "g7c_ppo_vclip", "g7d_ppo_condstd", "g7e_ppo_unequal"])
def test_hip_train_step_golden(ctx, name):
    g = load_golden(name)
    tr = _hip_ppo(ctx, g)
    tr.train_step(_trajs(g), g["perms"])
    # g7c runs the value net at lr 3e-3 (10x the others): one Adam step moves a weight by up to 3e-3, the bound scales with it
    np.testing.assert_allclose(tr.get_flat_params(1), g["vf_final"], rtol=0, atol=2e-4 if name == "g7c_ppo_vclip" else 5e-5)
    np.testing.assert_allclose(tr.get_flat_params(0), g["pi_final"], rtol=0, atol=5e-5)
    assert np.abs(tr.get_flat_params(0) - g["pi0"]).max() > 1e-4
    if name == "g7d_ppo_condstd":   # conditioned_std=True: the fixed log-probs come from the clamped second head; deterministic action = the mean
        tr2 = _hip_ppo(ctx, g)
        _, _, _, lp = tr2.calc_adv(_trajs(g))
        np.testing.assert_allclose(lp, g["fixed_log_probs"], rtol=1e-5, atol=1e-5)
        obs = _trajs(g)[2]["observations"]
        from oracle.ppo import PPOOracle
        orc = PPOOracle(int(g["dims"][0]), int(g["dims"][1]), [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **_kw_of(g))
        np.testing.assert_allclose(tr2.policy_act(obs, True)[0], orc.pi_mean(obs)[0], rtol=1e-5, atol=1e-6)
        assert int(g["n_outside_clamp"]) > 0


@pytest.mark.gpu
def test_hip_train_step_vs_oracle_ragged(ctx):
    """Widths / sizes the goldens do not hold: H=128, 5 trajectories (one of length 2), N not a multiple of the
    minibatch (ragged last minibatch), 3 epochs."""
    from oracle import mlp as omlp
    rng = np.random.default_rng(99)
    o, a, hid = 17, 6, [128, 128]
    kw = dict(KW, mini_batch_size=48, update_epoch=3, gae_tau=0.9)
    vf0 = omlp.init_mlp(rng, o, hid, 1)
    pi0 = np.concatenate([omlp.init_mlp(rng, o, hid, a, init_w=1e-3, last_scale=(0.1, 0.0)),
                          rng.normal(-0.5, 0.2, a).astype(np.float32)])
    trajs = [dict(observations=rng.normal(0, 1, (L, o)).astype(np.float32),
                  actions=rng.normal(0, 0.7, (L, a)).astype(np.float32),
                  rewards=rng.normal(0.5, 1.0, (L, 1)).astype(np.float32)) for L in (2, 31, 100, 64, 9)]
    N = sum(t["rewards"].shape[0] for t in trajs)
    perms = np.stack([rng.permutation(N) for _ in range(3)])
    orc = PPOOracle(o, a, hid, pi0, vf0, **kw)
    g = dict(dims=np.array([o, a] + hid), epochs=3, pi0=pi0, vf0=vf0)
    tr = _hip_ppo(ctx, g, mini_batch_size=48, gae_tau=0.9)
    R, A, V, lp = tr.calc_adv(trajs)
    _, _, R0, A0, V0 = orc.calc_adv(trajs)
    np.testing.assert_allclose(V, V0, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(A, A0, rtol=2e-4, atol=2e-5)
    orc.train_step(trajs, list(perms))
    tr.train_step(trajs, perms)
    np.testing.assert_allclose(tr.get_flat_params(1), orc.vf, rtol=0, atol=5e-5)
    np.testing.assert_allclose(tr.get_flat_params(0), orc.pi, rtol=0, atol=5e-5)


@pytest.mark.gpu
def test_hip_gaussian_policy_act(ctx):
    g = load_golden("g7_ppo")
    tr = _hip_ppo(ctx, g)
    orc = PPOOracle(int(g["dims"][0]), int(g["dims"][1]), [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **KW)
    obs = g["t2_observations"]
    eps = np.random.default_rng(5).normal(0, 1, (obs.shape[0], 3)).astype(np.float32)
    mu, _ = orc.pi_mean(obs)
    act, lp = tr.policy_act(obs, eps=eps)
    np.testing.assert_allclose(act, mu + np.exp(g["pi0"][-3:]) * eps, rtol=1e-5, atol=1e-6)   # policies.py:409-417
    np.testing.assert_allclose(lp, orc.log_prob(obs, act)[0], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tr.policy_act(obs, deterministic=True)[0], mu, rtol=1e-5, atol=1e-6)
    a1, a2 = tr.policy_act(obs)[0], tr.policy_act(obs)[0]                                       # Philox draws move on
    assert np.abs(a1 - a2).max() > 1e-3 and np.isfinite(a1).all()


@pytest.mark.gpu
def test_hip_library_shuffle_is_a_permutation(ctx):
    import ctypes as C
    from ilswiss_amd import _lib
    g = load_golden("g7_ppo")
    tr = _hip_ppo(ctx, g)
    seen = []
    for n, key in ((1, 1), (2, 1), (72, 1), (72, 2), (4096, 7), (4095, 7), (3001, 9)):
        perm = np.empty(n, np.int32)
        _lib.check(ctx.lib.ilsx_ppo_debug_perm(tr.h, n, key, perm.ctypes.data_as(C.c_void_p)))
        np.testing.assert_array_equal(np.sort(perm), np.arange(n))
        seen.append(perm)
    assert (seen[2] != seen[3]).any()                       # a new key reshuffles
    assert np.abs(np.corrcoef(seen[4], np.arange(4096))[0, 1]) < 0.1   # and looks nothing like the identity
    # the NULL-perms train path runs and moves both networks
    p0, v0 = tr.get_flat_params(0), tr.get_flat_params(1)
    tr.train_step(_trajs(g))
    assert np.abs(tr.get_flat_params(0) - p0).max() > 1e-5 and np.abs(tr.get_flat_params(1) - v0).max() > 1e-5
    assert np.isfinite(tr.get_flat_params(0)).all()


@pytest.mark.gpu
def test_hip_gae_chunk_edges_and_long_trajectories(ctx):
    """Trajectory lengths around the 64-sample chunking and the 1024-sample register window of k_ppo_gae; a
    single-sample trajectory standardises to nan exactly like torch.std (ppo.py:86)."""
    rng = np.random.default_rng(17)
    g = load_golden("g7_ppo")
    o, a = int(g["dims"][0]), int(g["dims"][1])
    orc = PPOOracle(o, a, [int(v) for v in g["dims"][2:]], g["pi0"], g["vf0"], **dict(KW, reward_scale=0.7))
    tr = _hip_ppo(ctx, g, reward_scale=0.7)
    trajs = [dict(observations=rng.normal(0, 1, (L, o)).astype(np.float32), actions=rng.normal(0, 0.7, (L, a)).astype(np.float32),
                  rewards=rng.normal(0.5, 1.0, (L, 1)).astype(np.float32)) for L in (1500, 64, 65, 128, 1, 1024, 1025, 63, 2)]
    R, A, V, lp = tr.calc_adv(trajs)
    _, _, R0, A0, V0 = orc.calc_adv(trajs)
    np.testing.assert_allclose(V, V0, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(R, R0, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(A, A0, rtol=2e-4, atol=5e-5, equal_nan=True)
    assert np.isnan(A).sum() == 1


@pytest.mark.gpu
def test_hip_train_step_large_minibatch_row_split_dw(ctx):
    """Minibatches >= 1024 rows take the row-split weight-gradient path (slabs + k_dw_reduce, Adam in the reduce)."""
    from oracle import mlp as omlp
    rng = np.random.default_rng(123)
    o, a, hid = 11, 3, [64, 64]
    kw = dict(KW, mini_batch_size=1500, update_epoch=2)
    vf0 = omlp.init_mlp(rng, o, hid, 1)
    pi0 = np.concatenate([omlp.init_mlp(rng, o, hid, a, init_w=1e-3, last_scale=(0.1, 0.0)), rng.normal(-0.5, 0.2, a).astype(np.float32)])
    trajs = [dict(observations=rng.normal(0, 1, (L, o)).astype(np.float32), actions=rng.normal(0, 0.7, (L, a)).astype(np.float32),
                  rewards=rng.normal(0.5, 1.0, (L, 1)).astype(np.float32)) for L in (700, 900, 1000, 650)]
    N = sum(t["rewards"].shape[0] for t in trajs)     # 3250 = 1500 + 1500 + 250 (the last minibatch is NOT split)
    perms = np.stack([rng.permutation(N) for _ in range(2)])
    orc = PPOOracle(o, a, hid, pi0, vf0, **kw)
    tr = _hip_ppo(ctx, dict(dims=np.array([o, a] + hid), epochs=2, pi0=pi0, vf0=vf0), mini_batch_size=1500)
    orc.train_step(trajs, list(perms))
    tr.train_step(trajs, perms)
    np.testing.assert_allclose(tr.get_flat_params(1), orc.vf, rtol=0, atol=5e-5)
    np.testing.assert_allclose(tr.get_flat_params(0), orc.pi, rtol=0, atol=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("hid,mb", [([256, 256], 5003), ([64, 64], 4096), ([128, 128], 6000), ([64, 64], 10241)])   # 10241 = 20 x 512 + 1: whole-chunk row ranges would leave the last one empty
def test_hip_train_step_big_minibatch_block_dw(ctx, hid, mb):
    """Minibatches >= 4096 rows take the LDS-staged 128 x 128 block weight-gradient kernel (k_dw_big): 32-row chunks, row ranges that are
    not whole chunks (5003 rows), blocks wider than the matrix (H = 64, the 16-wide first layer, the 3-wide head) == the oracle."""
    from oracle import mlp as omlp
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
    rng = np.random.default_rng(321)
    o, a = 11, 3
    kw = dict(KW, mini_batch_size=mb, update_epoch=2)
    vf0 = omlp.init_mlp(rng, o, hid, 1)
    pi0 = np.concatenate([omlp.init_mlp(rng, o, hid, a, init_w=1e-3, last_scale=(0.1, 0.0)), rng.normal(-0.5, 0.2, a).astype(np.float32)])
    lens = (3000, 2500, 4000, 1900)
    trajs = [dict(observations=rng.normal(0, 1, (L, o)).astype(np.float32), actions=rng.normal(0, 0.7, (L, a)).astype(np.float32),
                  rewards=rng.normal(0.5, 1.0, (L, 1)).astype(np.float32)) for L in lens]
    N = sum(lens)   # 11400: two big minibatches and a ragged last one
    perms = np.stack([rng.permutation(N) for _ in range(2)])
    orc = PPOOracle(o, a, hid, pi0, vf0, **kw)
    pol = ReparamMultivariateGaussianPolicy(hid, o, a, conditioned_std=False, hidden_activation="tanh", ctx=ctx, seed=3)
    vf = FlattenMlp(hid, 1, o, hidden_activation="tanh", ctx=ctx, seed=4)
    tr = PPO(pol, vf, max_samples=12000, **kw)
    tr.set_flat_params(pi0, vf0)
    orc.train_step(trajs, list(perms))
    tr.train_step(trajs, perms)
    np.testing.assert_allclose(tr.get_flat_params(1), orc.vf, rtol=0, atol=5e-5)
    np.testing.assert_allclose(tr.get_flat_params(0), orc.pi, rtol=0, atol=5e-5)
    assert np.abs(tr.get_flat_params(0) - pi0).max() > 1e-4


def test_policy_ctor_defaults_are_the_references():
    """policies.py:131-140 (MlpGaussianNoisePolicy), :349-356 (ReparamMultivariateGaussianPolicy) over networks.py:24-32 (Mlp), typed in:
    a caller that relies on the class defaults gets the reference's network — or an explicit NotImplementedError, never another one."""
    import inspect
    from ilswiss_amd.ppo import ReparamMultivariateGaussianPolicy
    from ilswiss_amd.td3 import MlpGaussianNoisePolicy
    d = {n: p.default for n, p in inspect.signature(ReparamMultivariateGaussianPolicy.__init__).parameters.items()}
    assert (d["conditioned_std"], d["init_w"], d["hidden_activation"]) == (True, 1e-3, "relu")
    d = {n: p.default for n, p in inspect.signature(MlpGaussianNoisePolicy.__init__).parameters.items()}
    assert (d["init_w"], d["policy_noise"], d["policy_noise_clip"], d["max_act"], d["output_activation"]) == (1e-3, 0.1, 0.5, 1.0, "identity")
    with pytest.raises(NotImplementedError, match="output_activation"):
        MlpGaussianNoisePolicy([64, 64], 11, 3, output_activation="softmax", ctx=object())

