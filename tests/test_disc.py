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
            np.testing.assert_allclose(res["grad_pen_loss"], 8.0 * g[f"{tag}_s{s}_gp"], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(res["accuracy"], g[f"{tag}_s{s}_acc"])
            ref = g[f"{tag}_s{s}_grad"]
            assert np.abs(res["grad"] - ref).max() <= 1e-4 * np.abs(ref).max()
            np.testing.assert_allclose(orc.p, g[f"{tag}_s{s}_params"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(orc.logits(g[f"{tag}_probe"]), g[f"{tag}_probe_logits"], rtol=1e-4, atol=1e-5)
    assert np.abs(g["tanh_sat_probe_logits"]).max() == 10.0  # the saturated case really exercises the clamp
    # ... also on the gradient-penalty rows: clamped interpolates have dD/dx == 0, count (0-1)^2 in the penalty and give no
    # gradient (torch's norm backward is 0 at 0); everything stays finite
    orc = DiscOracle(*[int(v) for v in g["tanh_sat_dims"][:2]], g["tanh_sat_params0"], act=TANH, **KW)
    res = orc.train_step(g["tanh_sat_s0_x_exp"], g["tanh_sat_s0_x_pol"], g["tanh_sat_s0_eps"])
    assert (res["grad_norm"] == 0).sum() >= 1 and np.isfinite(res["grad"]).all() and np.isfinite(g["tanh_sat_s0_grad"]).all()


BLOCK_CASES = [("tanh1", TANH), ("tanh3", TANH), ("relu3", RELU), ("relu1", RELU)]


@pytest.mark.parametrize("tag,act", BLOCK_CASES)
def test_oracle_disc_blocks_golden(tag, act):
    """g25: MLPDisc(num_layer_blocks = 1 / 3) through the reference's autograd double backward vs the general-depth restatement."""
    g = load_golden("g25_disc_blocks")
    D, Hd, B, steps, _, L = [int(v) for v in g[f"{tag}_dims"]]
    orc = DiscOracle(D, Hd, g[f"{tag}_params0"], act=act, num_layer_blocks=L, **KW)
    for s in range(steps):
        res = orc.train_step_blocks(g[f"{tag}_s{s}_x_exp"], g[f"{tag}_s{s}_x_pol"], g[f"{tag}_s{s}_eps"])
        np.testing.assert_allclose(res["ce_loss"], g[f"{tag}_s{s}_ce"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(res["grad_pen_loss"], 8.0 * g[f"{tag}_s{s}_gp"], rtol=1e-4, atol=1e-6)
        if s == 0:
            ref = g[f"{tag}_s0_grad"]
            assert np.abs(res["grad"] - ref).max() <= 1e-4 * np.abs(ref).max()
    np.testing.assert_allclose(orc.p, g[f"{tag}_params_final"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(orc.forward_blocks(g[f"{tag}_probe"])[0], g[f"{tag}_probe_logits"], rtol=1e-4, atol=1e-5)


def test_oracle_blocks_routine_is_the_two_block_routine():
    """train_step_blocks at L = 2 == the hand-written two-block train_step (the one pinned by g8)."""
    from oracle import mlp as omlp
    rng = np.random.default_rng(0)
    D, Hd, B = 9, 16, 12
    for act in (TANH, RELU):
        flat = omlp.init_mlp(rng, D, [Hd, Hd], 1, init_w=0.3)
        a, b = DiscOracle(D, Hd, flat, act=act), DiscOracle(D, Hd, flat, act=act, num_layer_blocks=2)
        xe, xp, e = (rng.normal(0, 1, (B, D)).astype(np.float32), rng.normal(0, 1, (B, D)).astype(np.float32), rng.random((B, 1)).astype(np.float32))
        ra, rb = a.train_step(xe, xp, e), b.train_step_blocks(xe, xp, e)
        assert np.abs(ra["grad"] - rb["grad"]).max() <= 1e-6 * np.abs(ra["grad"]).max()
        assert ra["grad_pen_loss"] == rb["grad_pen_loss"]


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
    disc = MLPDisc(D, hid_dim=Hd, hid_act=tag.split("_")[0], use_bn=False, ctx=ctx).bind(o, max_batch=B, **KW)
    disc.set_flat_params(g[f"{tag}_params0"])
    np.testing.assert_array_equal(disc.get_flat_params(), g[f"{tag}_params0"])
    for s in range(steps):
        xe, xp = g[f"{tag}_s{s}_x_exp"], g[f"{tag}_s{s}_x_pol"]
        st = disc.train_step(xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:], eps=g[f"{tag}_s{s}_eps"])
        np.testing.assert_allclose(st["Disc CE Loss"], g[f"{tag}_s{s}_ce"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Grad Pen"], g[f"{tag}_s{s}_gp"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Disc Acc"], g[f"{tag}_s{s}_acc"])
        ref = g[f"{tag}_s{s}_grad"]
        got = disc.get_flat_grads()
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (s, np.abs(got - ref).max(), np.abs(ref).max())
        np.testing.assert_allclose(disc.get_flat_params(), g[f"{tag}_s{s}_params"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(disc(g[f"{tag}_probe"]), g[f"{tag}_probe_logits"], rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,act", BLOCK_CASES)
def test_hip_disc_blocks_golden(ctx, tag, act):
    """num_layer_blocks = 1 and 3 on the HIP path (the per-layer launch chain, ilsx_disc.hip disc_step_blocks) vs the reference's numbers."""
    from ilswiss_amd.adv_irl import MLPDisc
    g = load_golden("g25_disc_blocks")
    D, Hd, B, steps, o, L = [int(v) for v in g[f"{tag}_dims"]]
    disc = MLPDisc(D, num_layer_blocks=L, hid_dim=Hd, hid_act=tag[:4], use_bn=False, ctx=ctx).bind(o, max_batch=B, **KW)
    disc.set_flat_params(g[f"{tag}_params0"])
    np.testing.assert_array_equal(disc.get_flat_params(), g[f"{tag}_params0"])
    for s in range(steps):
        xe, xp = g[f"{tag}_s{s}_x_exp"], g[f"{tag}_s{s}_x_pol"]
        st = disc.train_step(xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:], eps=g[f"{tag}_s{s}_eps"])
        np.testing.assert_allclose(st["Disc CE Loss"], g[f"{tag}_s{s}_ce"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Grad Pen"], g[f"{tag}_s{s}_gp"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Disc Acc"], g[f"{tag}_s{s}_acc"])
        if s == 0:
            ref, got = g[f"{tag}_s0_grad"], disc.get_flat_grads()
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())
    np.testing.assert_allclose(disc.get_flat_params(), g[f"{tag}_params_final"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(disc(g[f"{tag}_probe"]), g[f"{tag}_probe_logits"], rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("o,a,Hd,B,act,gp", [(17, 6, 128, 256, "tanh", True), (11, 3, 256, 37, "relu", True),
                                               (17, 6, 64, 48, "tanh", False), (11, 3, 128, 16, "tanh", True),
                                               (17, 6, 100, 64, "relu", True), (11, 3, 100, 32, "tanh", True)])   # 100: the reference default, zero-padded to 128
def test_hip_disc_vs_oracle(ctx, o, a, Hd, B, act, gp):
    from ilswiss_amd.adv_irl import MLPDisc
    rng = np.random.default_rng(o + Hd + B)
    D = o + a
    flat = omlp.init_mlp(rng, D, [Hd, Hd], 1, init_w=0.2, b_init=0.02)
    kw = dict(KW, use_grad_pen=gp)
    disc = MLPDisc(D, hid_dim=Hd, hid_act=act, use_bn=False, ctx=ctx).bind(o, max_batch=B, **kw)
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
            np.testing.assert_allclose(st["Grad Pen"] * 8.0, res["grad_pen_loss"], rtol=1e-4, atol=1e-6)
        got, ref = disc.get_flat_grads(), res["grad"]
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7, (s, np.abs(got - ref).max(), np.abs(ref).max())
        np.testing.assert_allclose(disc.get_flat_params(), orc.p, rtol=0, atol=5e-5)


@pytest.mark.gpu
def test_hip_reward_modes_golden(ctx):
    """Reward modes on a logits grid: a discriminator whose output IS its first input (clamped)."""
    from ilswiss_amd.adv_irl import MLPDisc
    g = load_golden("g8_g9_disc")
    o, a, Hd = 2, 1, 64
    disc = MLPDisc(o + a, hid_dim=Hd, hid_act="relu", use_bn=False, ctx=ctx).bind(o, max_batch=64, **KW)
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
    disc = MLPDisc(o + a, hid_dim=128, hid_act="tanh", use_bn=False, ctx=ctx, seed=1)
    H = [64, 64]
    pol = ia.ReparamTanhMultivariateGaussianPolicy(H, o, a, ctx=ctx, seed=2)
    q1, q2 = ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=3), ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=4)
    sac = ia.SoftActorCritic(pol, q1, q2, reward_scale=2.0, beta_1=0.25, max_batch=256)   # gail_walker.yaml:12,74
    irl = AdvIRLTrainer("gail2", disc, sac, exp_rb, replay_buffer=rb, num_update_loops_per_train_call=150, disc_optim_batch_size=256,
                        policy_optim_batch_size=256, num_disc_updates_per_loop_iter=1, num_policy_updates_per_loop_iter=1, **KW)
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


# ------------------------------------------------------------------------------------------------ branches the YAMLs leave off (g24)
def _b(g, st, tag):
    return {k: g[f"s{st}_{tag}_{k}"] for k in ("observations", "actions", "rewards", "terminals", "next_observations")}


def test_oracle_state_only_and_expert_rows_golden():
    g = load_golden("g24_disc_branches")
    o, a, Hd, B, steps = [int(v) for v in g["dims"]]
    orc = DiscOracle(2 * o, Hd, g["params0"], act=TANH, disc_lr=1e-3, disc_momentum=0.0, use_grad_pen=True, grad_pen_weight=10.0)
    for st in range(steps):
        be, bp = _b(g, st, "exp"), _b(g, st, "pol")
        res = orc.train_step(np.concatenate([be["observations"], be["next_observations"]], 1),
                             np.concatenate([bp["observations"], bp["next_observations"]], 1), g[f"s{st}_eps"])
        np.testing.assert_allclose(res["ce_loss"], g[f"s{st}_ce"], rtol=1e-4, atol=1e-6)
        assert np.abs(res["grad"] - g[f"s{st}_grad"]).max() <= 1e-4 * np.abs(g[f"s{st}_grad"]).max()
        np.testing.assert_allclose(orc.p, g[f"s{st}_params"], rtol=0, atol=5e-5)
    pol = {k[7:]: v for k, v in g.items() if k.startswith("pt_pol_")}
    exp = {k[7:]: v for k, v in g.items() if k.startswith("pt_exp_")}
    x = np.concatenate([np.concatenate([pol["observations"], exp["observations"]]), np.concatenate([pol["next_observations"], exp["next_observations"]])], 1)
    np.testing.assert_allclose(disc_reward(orc.logits(x), "gail2", rew_clip_min=float(g["pt_clip"][0]), rew_clip_max=float(g["pt_clip"][1])),
                               g["pt_rewards"], rtol=1e-4, atol=1e-5)


def test_ctor_signatures_are_the_references():
    """simple_disc_models.py:9-17 and adv_irl.py:34-54, typed in as the config schema they are: same names, order and defaults."""
    import inspect
    from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
    sig = inspect.signature(MLPDisc.__init__)
    pos = [(n, p.default) for n, p in sig.parameters.items() if p.kind == p.POSITIONAL_OR_KEYWORD][1:]
    assert pos == [("input_dim", inspect.Parameter.empty), ("num_layer_blocks", 2), ("hid_dim", 100), ("hid_act", "relu"), ("use_bn", True),
                   ("clamp_magnitude", 10.0)]
    sig = inspect.signature(AdvIRLTrainer.__init__)
    pos = [(n, p.default) for n, p in sig.parameters.items() if p.kind == p.POSITIONAL_OR_KEYWORD][1:20]
    assert pos == [("mode", inspect.Parameter.empty), ("discriminator", inspect.Parameter.empty), ("policy_trainer", inspect.Parameter.empty),
                   ("expert_replay_buffer", inspect.Parameter.empty), ("state_only", False), ("disc_optim_batch_size", 1024),
                   ("policy_optim_batch_size", 1024), ("policy_optim_batch_size_from_expert", 0), ("num_update_loops_per_train_call", 1),
                   ("num_disc_updates_per_loop_iter", 100), ("num_policy_updates_per_loop_iter", 100), ("disc_lr", 1e-3), ("disc_momentum", 0.0),
                   ("disc_optimizer_class", None), ("use_grad_pen", True), ("grad_pen_weight", 10), ("rew_clip_min", None), ("rew_clip_max", None),
                   ("replay_buffer", None)]


def test_unimplemented_disc_options_fail_loudly():
    from ilswiss_amd.adv_irl import MLPDisc
    d = MLPDisc(23, ctx=object())                      # the reference's default network HAS BatchNorm (simple_disc_models.py:15) and is built as such
    assert d.use_bn and d.hid_dim == 100 and d.num_params == (23 * 100 + 3 * 100) + (100 * 100 + 3 * 100) + 100 + 1
    with pytest.raises(NotImplementedError, match="num_layer_blocks"):
        MLPDisc(23, num_layer_blocks=4, use_bn=False, ctx=object())


@pytest.mark.gpu
def test_hip_state_only_disc_steps_golden(ctx):
    from ilswiss_amd.adv_irl import MLPDisc
    g = load_golden("g24_disc_branches")
    o, a, Hd, B, steps = [int(v) for v in g["dims"]]
    disc = MLPDisc(2 * o, hid_dim=Hd, hid_act="tanh", use_bn=False, ctx=ctx).bind(o, o, state_only=True, max_batch=B)   # AdvIRL's default optimiser
    disc.set_flat_params(g["params0"])
    for st in range(steps):
        be, bp = _b(g, st, "exp"), _b(g, st, "pol")
        stt = disc.train_step(be["observations"], be["next_observations"], bp["observations"], bp["next_observations"], eps=g[f"s{st}_eps"])
        np.testing.assert_allclose(stt["Disc CE Loss"], g[f"s{st}_ce"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(stt["Grad Pen"], g[f"s{st}_gp"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(stt["Disc Acc"], g[f"s{st}_acc"])
        ref, got = g[f"s{st}_grad"], disc.get_flat_grads()
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
        np.testing.assert_allclose(disc.get_flat_params(), g[f"s{st}_params"], rtol=0, atol=5e-5)


@pytest.mark.gpu
def test_hip_policy_batch_from_expert_and_state_only_relabel_golden(ctx):
    """adv_irl.py:239-255 + :269: the policy batch is cat([policy-buffer rows, expert-buffer rows]); the whole batch is relabelled by the
    (state-only) discriminator and handed to the policy trainer.  The two rings hold exactly the fixture's rows, so whatever the draw, the
    rows that reach the SAC step must be policy rows first, expert rows last, with the reference's rewards for those rows."""
    import ctypes as C
    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
    g = load_golden("g24_disc_branches")
    o, a, Hd, _, _ = [int(v) for v in g["dims"]]
    Bp, nfe = [int(v) for v in g["pt_dims"]]
    pol = {k[7:]: v for k, v in g.items() if k.startswith("pt_pol_")}
    exp = {k[7:]: v for k, v in g.items() if k.startswith("pt_exp_")}
    rb, erb = ia.SimpleReplayBuffer(Bp - nfe, o, a, ctx=ctx), ia.SimpleReplayBuffer(nfe, o, a, ctx=ctx)
    for buf, b in ((rb, pol), (erb, exp)):
        buf.add_rows(b["observations"], b["actions"], b["rewards"][:, 0], b["terminals"][:, 0].astype(np.uint8), b["next_observations"])
    disc = MLPDisc(2 * o, hid_dim=Hd, hid_act="tanh", use_bn=False, ctx=ctx)
    disc.set_flat_params(g["s1_params"])      # the discriminator the reference relabelled with (after its two steps)
    H = [64, 64]
    sac = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy(H, o, a, ctx=ctx, seed=2), ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=3),
                             ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=4), max_batch=Bp)
    irl = AdvIRLTrainer("gail2", disc, sac, erb, state_only=True, disc_optim_batch_size=nfe, policy_optim_batch_size=Bp,
                        policy_optim_batch_size_from_expert=nfe, num_disc_updates_per_loop_iter=0, num_policy_updates_per_loop_iter=1,
                        rew_clip_min=float(g["pt_clip"][0]), rew_clip_max=float(g["pt_clip"][1]), replay_buffer=rb)
    # reference rewards by row content (draws are with replacement: look every used row up in the fixture)
    allobs = np.concatenate([pol["observations"], exp["observations"]])
    for path in ("fused", "python"):
        if path == "fused":
            irl.train(1)                               # ilsx_advirl_train
        else:
            irl._do_policy_training()
        bufs = [ctx.empty((Bp, o)), ctx.empty((Bp, a)), ctx.empty((Bp,)), ctx.empty((Bp,)), ctx.empty((Bp, o))]
        _lib.check(ctx.lib.ilsx_sac_debug_last_batch(sac.h, Bp, *[b.ptr for b in bufs], None))
        obs, act, rew, done, nobs = [b.numpy() for b in bufs]
        src = np.array([int(np.flatnonzero((allobs == r).all(1))[0]) for r in obs])
        assert (src[: Bp - nfe] < Bp - nfe).all() and (src[Bp - nfe:] >= Bp - nfe).all(), (path, src)   # policy rows first, expert rows last
        np.testing.assert_allclose(rew, g["pt_rewards"][src, 0], rtol=1e-4, atol=2e-5)
        np.testing.assert_array_equal(nobs, np.concatenate([pol["next_observations"], exp["next_observations"]])[src])


BN_CASES = ["tanh2", "relu2", "tanh3", "relu1"]


def _bn_check_final(g, tag, disc, steps, lr):
    dead = g[f"{tag}_dead_bias_mask"].astype(bool)
    d = np.abs(disc.get_flat_params() - g[f"{tag}_params_final"])
    # Linear biases under a BatchNorm have gradient exactly 0; what any implementation holds there is rounding noise that Adam turns into
    # +-lr steps (they do not influence the function).  Every other parameter: 5e-5 after the chained steps
    assert d[~dead].max() < 5e-5 and d[dead].max() <= steps * 2.02 * lr, (tag, d[~dead].max(), d[dead].max())
    rm, rv = disc.get_bn_stats()
    np.testing.assert_allclose(rv, g[f"{tag}_running_var"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rm, g[f"{tag}_running_mean"], rtol=0, atol=steps * 2.02 * lr + 1e-5)
    np.testing.assert_allclose(disc(g[f"{tag}_probe"]), g[f"{tag}_probe_logits_eval"], rtol=1e-3, atol=3e-3)   # eval mode: running statistics


@pytest.mark.gpu
@pytest.mark.parametrize("tag", BN_CASES)
def test_hip_disc_bn_golden(ctx, tag):
    """MLPDisc(use_bn=True) — the reference constructor's default — on the HIP path vs the reference's own numbers (g26: AdvIRL._do_reward_training
    with the module in train mode, then the eval-mode logits of _do_policy_training): chained steps, tanh / relu, 1-3 blocks, hid_dim 100."""
    from ilswiss_amd.adv_irl import MLPDisc
    g = load_golden("g26_disc_bn")
    D, Hd, B, steps, o, L = [int(v) for v in g[f"{tag}_dims"]]
    disc = MLPDisc(D, num_layer_blocks=L, hid_dim=Hd, hid_act=tag[:4], use_bn=True, ctx=ctx).bind(o, max_batch=max(B, 20), **KW)
    assert disc.num_params == g[f"{tag}_params0"].size
    disc.set_flat_params(g[f"{tag}_params0"])
    np.testing.assert_array_equal(disc.get_flat_params(), g[f"{tag}_params0"])
    for s in range(steps):
        xe, xp = g[f"{tag}_s{s}_x_exp"], g[f"{tag}_s{s}_x_pol"]
        st = disc.train_step(xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:], eps=g[f"{tag}_s{s}_eps"])
        np.testing.assert_allclose(st["Disc CE Loss"], g[f"{tag}_s{s}_ce"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Grad Pen"], g[f"{tag}_s{s}_gp"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Disc Acc"], g[f"{tag}_s{s}_acc"])
        if s == 0:
            ref, got, live = g[f"{tag}_s0_grad"], disc.get_flat_grads(), ~g[f"{tag}_dead_bias_mask"].astype(bool)
            assert np.abs(got - ref)[live].max() <= 1e-4 * np.abs(ref).max(), (np.abs(got - ref)[live].max(), np.abs(ref).max())
    _bn_check_final(g, tag, disc, steps, KW["disc_lr"])


@pytest.mark.gpu
@pytest.mark.parametrize("o,a,Hd,B,act,L,gp", [(17, 6, 128, 256, "tanh", 2, True), (11, 3, 100, 64, "relu", 2, True), (17, 6, 64, 48, "tanh", 3, False),
                                             (11, 3, 64, 300, "tanh", 2, True)])   # 600 CE rows: a column is more than one 512-row batch (disc_bn.h DBN_U)
def test_hip_disc_bn_vs_oracle(ctx, o, a, Hd, B, act, L, gp):
    """the same at the configs' sizes (GAIL Walker2d: 23 -> 128 -> 128 -> 1, B = 256) against oracle.DiscBNOracle, three chained steps,
    with the relabelled rewards of the eval-mode forward (adv_irl.py:268-301) after them"""
    from ilswiss_amd.adv_irl import MLPDisc
    from oracle.disc import DiscBNOracle, disc_reward
    rng = np.random.default_rng(o + Hd + B)
    D = o + a
    flat = DiscBNOracle.init(rng, D, Hd, L)
    kw = dict(KW, use_grad_pen=gp)
    disc = MLPDisc(D, num_layer_blocks=L, hid_dim=Hd, hid_act=act, use_bn=True, ctx=ctx).bind(o, max_batch=B, **kw)
    disc.set_flat_params(flat)
    orc = DiscBNOracle(D, Hd, flat, act=TANH if act == "tanh" else RELU, num_layer_blocks=L, **kw)
    live = ~orc.dead_bias_mask()
    for s in range(3):
        xe = rng.normal(0, 1, (B, D)).astype(np.float32)
        xp = (rng.normal(0, 1, (B, D)) * 1.5 + 0.3).astype(np.float32)
        eps = rng.random((B, 1)).astype(np.float32)
        st = disc.train_step(xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:], eps=eps)
        res = orc.train_step(xe, xp, eps)
        np.testing.assert_allclose(st["Disc CE Loss"], res["ce_loss"], rtol=1e-4, atol=1e-6)
        if gp:
            np.testing.assert_allclose(st["Grad Pen"] * kw["grad_pen_weight"], res["grad_pen_loss"], rtol=2e-4, atol=1e-6)
        got, ref = disc.get_flat_grads(), res["grad"]
        assert np.abs(got - ref)[live].max() <= 1e-4 * np.abs(ref).max(), (s, np.abs(got - ref)[live].max() / np.abs(ref).max())
    d = np.abs(disc.get_flat_params() - orc.p)
    assert d[live].max() < 5e-5, d[live].max()
    x = (rng.normal(0, 1, (B, D)) * 1.2).astype(np.float32)
    rew, lg = disc.rewards(x[:, :o], x[:, o:], "gail2", rew_clip_min=-8.0)
    np.testing.assert_allclose(lg, orc.logits(x), rtol=1e-3, atol=3e-3)
    np.testing.assert_allclose(rew, disc_reward(orc.logits(x), "gail2", rew_clip_min=-8.0), rtol=1e-3, atol=3e-3)


@pytest.mark.gpu
def test_hip_gail_loop_with_the_default_bn_discriminator(ctx):
    """AdvIRL around MLPDisc's DEFAULT constructor arguments (use_bn=True, hid_dim=100, relu): the device loop (ilsx_advirl_train) runs,
    statistics are finite, the discriminator learns to separate two shifted clouds, and a snapshot round trip (parameters, Adam state,
    running statistics) continues bit-exactly."""
    import ilswiss_amd as ia
    from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
    o, a, B = 11, 3, 64
    rng = np.random.default_rng(5)

    def ring(n, shift, seed):
        rb = ia.SimpleReplayBuffer(n, o, a, random_seed=seed, ctx=ctx)
        rb.add_rows(rng.normal(shift, 1, (n, o)).astype(np.float32), np.tanh(rng.normal(shift, 1, (n, a))).astype(np.float32),
                    rng.normal(0, 1, n).astype(np.float32), np.zeros(n, np.uint8), rng.normal(shift, 1, (n, o)).astype(np.float32))
        return rb
    erb, prb = ring(2000, 0.8, 1), ring(4000, -0.8, 2)

    def build(seed):
        pol = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], o, a, ctx=ctx, seed=seed)
        q1, q2 = ia.FlattenMlp([64, 64], 1, o + a, ctx=ctx, seed=seed + 1), ia.FlattenMlp([64, 64], 1, o + a, ctx=ctx, seed=seed + 2)
        sac = ia.SoftActorCritic(pol, q1, q2, max_batch=B, policy_lr=3e-4, qf_lr=3e-4)
        disc = MLPDisc(o + a, ctx=ctx, seed=seed + 3)                  # every default: 2 blocks of 100, relu, batch norm, clamp 10
        return AdvIRLTrainer("gail2", disc, sac, erb, disc_optim_batch_size=B, policy_optim_batch_size=B, num_update_loops_per_train_call=5,
                             num_disc_updates_per_loop_iter=1, num_policy_updates_per_loop_iter=1, replay_buffer=prb, disc_lr=3e-3, disc_momentum=0.9,
                             use_grad_pen=True, grad_pen_weight=4.0)
    irl = build(10)
    assert irl.disc.use_bn and irl.disc.hid_dim == 100
    irl.train(1)
    st0 = dict(irl.get_eval_statistics())
    for _ in range(30):
        irl.train(1)
    irl.end_epoch()
    irl.train(1)
    st1 = dict(irl.get_eval_statistics())
    for st in (st0, st1):
        assert all(np.isfinite(v) for v in st.values()), st
    assert st1["Disc Acc"] > 0.9 and st1["Disc CE Loss"] < st0["Disc CE Loss"], (st0, st1)
    rm, rv = irl.disc.get_bn_stats()
    assert np.isfinite(rm).all() and (rv > 0).all() and np.abs(rm).max() > 1e-3
    snap = irl.get_snapshot()
    irl2 = build(99)                     # different initial weights
    irl2.load_snapshot(snap)
    np.testing.assert_array_equal(irl2.disc.get_flat_params(), irl.disc.get_flat_params())
    np.testing.assert_array_equal(irl2.disc.get_bn_stats()[1], rv)
