"""Discriminator step (adv_irl.py:133-216) + reward modes (adv_irl.py:277-298):
CPU: oracle vs the golden vectors produced by the reference; GPU: HIP path vs fixtures and oracle."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import mlp as omlp
from oracle.disc import RELU, TANH, DiscOracle, disc_reward

KW = dict(disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=True, grad_pen_weight=8.0)
CASES = (("tanh", TANH), ("relu", RELU), ("tanh_sat", TANH))


def test_oracle_disc_steps_golden():
    g = load_golden("g8_g9_disc")
    for tag, act in CASES:
        D, Hd, B, steps, _ = [int(v) for v in g[f"{tag}_dims"]]
        orc = DiscOracle(D, Hd, g[f"{tag}_params0"], act=act, **KW)
        for s in range(steps):
            res = orc.train_step(g[f"{tag}_s{s}_x_exp"], g[f"{tag}_s{s}_x_pol"], g[f"{tag}_s{s}_eps"])
            np.testing.assert_allclose(res["ce_loss"], g[f"{tag}_s{s}_ce"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(res["grad_pen_loss"], 8.0 * g[f"{tag}_s{s}_gp"], rtol=2e-3, atol=1e-5)
            np.testing.assert_allclose(res["accuracy"], g[f"{tag}_s{s}_acc"])
            ref = g[f"{tag}_s{s}_grad"]
            assert np.abs(res["grad"] - ref).max() <= 5e-3 * np.abs(ref).max()
            np.testing.assert_allclose(orc.p, g[f"{tag}_s{s}_params"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(orc.logits(g[f"{tag}_probe"]), g[f"{tag}_probe_logits"], rtol=1e-4, atol=1e-5)
    assert np.abs(g["tanh_sat_probe_logits"]).max() == 10.0  # the saturated case really exercises the clamp
    # ... also on the gradient-penalty rows: clamped interpolates have dD/dx == 0, count (0-1)^2 in the penalty and give no
    # gradient (torch's norm backward is 0 at 0); everything stays finite
    orc = DiscOracle(*[int(v) for v in g["tanh_sat_dims"][:2]], g["tanh_sat_params0"], act=TANH, **KW)
    res = orc.train_step(g["tanh_sat_s0_x_exp"], g["tanh_sat_s0_x_pol"], g["tanh_sat_s0_eps"])
    assert (res["grad_norm"] == 0).sum() >= 1 and np.isfinite(res["grad"]).all() and np.isfinite(g["tanh_sat_s0_grad"]).all()


def test_oracle_reward_modes_golden():
    g = load_golden("g8_g9_disc")
    for mode in ("airl", "gail", "gail2", "fairl"):
        np.testing.assert_allclose(disc_reward(g["rew_grid"], mode), g[f"rew_{mode}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(disc_reward(g["rew_grid"], "gail2", rew_clip_min=-5.0, rew_clip_max=-0.5), g["rew_gail2_clip"], rtol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag,act", CASES)
def test_hip_disc_steps_golden(ctx, tag, act):
    from ilswiss_amd.adv_irl import MLPDisc
    g = load_golden("g8_g9_disc")
    D, Hd, B, steps, o = [int(v) for v in g[f"{tag}_dims"]]
    disc = MLPDisc(o, D - o, hid_dim=Hd, hid_act=tag.split("_")[0], max_batch=B, ctx=ctx,
                   **KW)
    disc.set_flat_params(g[f"{tag}_params0"])
    np.testing.assert_array_equal(disc.get_flat_params(), g[f"{tag}_params0"])
    for s in range(steps):
        xe, xp = g[f"{tag}_s{s}_x_exp"], g[f"{tag}_s{s}_x_pol"]
        st = disc.train_step(xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:], eps=g[f"{tag}_s{s}_eps"])
        np.testing.assert_allclose(st["Disc CE Loss"], g[f"{tag}_s{s}_ce"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Grad Pen"], g[f"{tag}_s{s}_gp"], rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(st["Disc Acc"], g[f"{tag}_s{s}_acc"])
        ref = g[f"{tag}_s{s}_grad"]
        got = disc.get_flat_grads()
        assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max(), (s, np.abs(got - ref).max(), np.abs(ref).max())
        np.testing.assert_allclose(disc.get_flat_params(), g[f"{tag}_s{s}_params"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(disc(g[f"{tag}_probe"]), g[f"{tag}_probe_logits"], rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("o,a,Hd,B,act,gp", [(17, 6, 128, 256, "tanh", True), (11, 3, 256, 37, "relu", True),
                                               (17, 6, 64, 48, "tanh", False), (11, 3, 128, 16, "tanh", True)])
def test_hip_disc_vs_oracle(ctx, o, a, Hd, B, act, gp):
    from ilswiss_amd.adv_irl import MLPDisc
    rng = np.random.default_rng(o + Hd + B)
    D = o + a
    flat = omlp.init_mlp(rng, D, [Hd, Hd], 1, init_w=0.2, b_init=0.02)
    kw = dict(KW, use_grad_pen=gp)
    disc = MLPDisc(o, a, hid_dim=Hd, hid_act=act, max_batch=B, ctx=ctx, **kw)
    disc.set_flat_params(flat)
    orc = DiscOracle(D, Hd, flat, act=TANH if act == "tanh" else RELU, **kw)
    for s in range(3):
        xe = rng.normal(0, 1, (B, D)).astype(np.float32)
        xp = (rng.normal(0.2, 1.3, (B, D))).astype(np.float32)
        eps = rng.random((B, 1)).astype(np.float32)
        st = disc.train_step(xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:], eps=eps)
        res = orc.train_step(xe, xp, eps)
        np.testing.assert_allclose(st["Disc CE Loss"], res["ce_loss"], rtol=1e-4, atol=1e-6)
        if gp:
            np.testing.assert_allclose(st["Grad Pen"] * 8.0, res["grad_pen_loss"], rtol=1e-3, atol=1e-5)
        got, ref = disc.get_flat_grads(), res["grad"]
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, (s, np.abs(got - ref).max(), np.abs(ref).max())
        np.testing.assert_allclose(disc.get_flat_params(), orc.p, rtol=0, atol=5e-5)


@pytest.mark.gpu
def test_hip_reward_modes_golden(ctx):
    """Reward modes on a logits grid: a discriminator whose output IS its first input (clamped)."""
    from ilswiss_amd.adv_irl import MLPDisc
    g = load_golden("g8_g9_disc")
    o, a, Hd = 2, 1, 64
    disc = MLPDisc(o, a, hid_dim=Hd, hid_act="relu", ctx=ctx, max_batch=64, **KW)
    W0 = np.zeros((Hd, 3), np.float32); W0[0, 0], W0[1, 0] = 1, -1
    W1 = np.zeros((Hd, Hd), np.float32); W1[0, 0] = W1[1, 1] = 1
    W2 = np.zeros((1, Hd), np.float32); W2[0, 0], W2[0, 1] = 1, -1
    z = np.zeros(Hd, np.float32)
    disc.set_flat_params(omlp.pack([(W0, z), (W1, z), (W2, np.zeros(1, np.float32))]))
    grid = g["rew_grid"].ravel()
    obs = np.stack([grid, np.zeros_like(grid)], 1)
    act = np.zeros((len(grid), 1), np.float32)
    for mode in ("airl", "gail", "gail2", "fairl"):
        r, lg = disc.rewards(obs, act, mode)
        ref = disc_reward(np.clip(g["rew_grid"], -10, 10), mode)   # MLPDisc clamps its output to +-10 first
        np.testing.assert_allclose(lg, np.clip(g["rew_grid"], -10, 10), rtol=0, atol=1e-6)
        np.testing.assert_allclose(r, ref, rtol=2e-5, atol=1e-6)
        inside = np.abs(grid) <= 10
        np.testing.assert_allclose(r[inside], g[f"rew_{mode}"][inside], rtol=2e-5, atol=1e-6)
    r, _ = disc.rewards(obs, act, "gail2", rew_clip_min=-5.0, rew_clip_max=-0.5)
    assert r.min() >= -5.0 and r.max() <= -0.5


@pytest.mark.gpu
def test_adv_irl_loop_separates_expert_from_policy(ctx):
    """k disc steps + m relabelled SAC steps per loop from two HBM buffers: the discriminator learns to
    tell the two data sources apart and the relabelled rewards follow the mode's sign."""
    import ilswiss_amd as ia
    from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
    rng = np.random.default_rng(4)
    o, a, n = 17, 6, 4000
    exp_rb, rb = ia.SimpleReplayBuffer(n, o, a, ctx=ctx), ia.SimpleReplayBuffer(n, o, a, ctx=ctx)
    eo = rng.normal(0.5, 1, (n, o)).astype(np.float32); ea = np.tanh(rng.normal(0.5, 1, (n, a))).astype(np.float32)
    po = rng.normal(-0.5, 1, (n, o)).astype(np.float32); pa = np.tanh(rng.normal(-0.5, 1, (n, a))).astype(np.float32)
    z = np.zeros(n, np.float32)
    exp_rb.add_rows(eo, ea, z, z.astype(np.uint8), eo)
    rb.add_rows(po, pa, z, z.astype(np.uint8), po)
    disc = MLPDisc(o, a, hid_dim=128, hid_act="tanh", max_batch=256, ctx=ctx, seed=1, **KW)
    H = [64, 64]
    pol = ia.ReparamTanhMultivariateGaussianPolicy(H, o, a, ctx=ctx, seed=2)
    q1, q2 = ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=3), ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=4)
    sac = ia.SoftActorCritic(pol, q1, q2, reward_scale=2.0, beta_1=0.25, max_batch=256)   # gail_walker.yaml:12,74
    irl = AdvIRLTrainer("gail2", disc, sac, exp_rb, rb, num_update_loops_per_train_call=150)
    irl.train()
    st0 = irl.get_eval_statistics()
    irl.end_epoch()
    irl.train()
    st = irl.get_eval_statistics()
    for k in ("Disc CE Loss", "Disc Acc", "Grad Pen", "Disc Rew Mean", "QF1 Loss", "Policy Loss"):
        assert k in st, k
    assert st["Disc Acc"] > 0.8 and st["Disc CE Loss"] < st0["Disc CE Loss"]
    assert st["Disc Rew Max"] <= 0.0   # gail2: log D <= 0
    r_e, _ = disc.rewards(eo[:256], ea[:256], "gail2")
    r_p, _ = disc.rewards(po[:256], pa[:256], "gail2")
    assert r_e.mean() > r_p.mean()
