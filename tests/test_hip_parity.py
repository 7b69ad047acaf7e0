"""GPU suite (-m gpu): the HIP path, called through the C ABI (ctypes adapters in ilswiss_amd/),
against (1) the golden vectors produced by the reference and (2) the oracle on seeded inputs.
Tolerances: fp32, rtol 1e-5 / atol 1e-6 for a single op; 5e-5 absolute on parameters after chained
optimiser steps; log-prob rows use the sensitivity-aware tolerance of oracle.tanh_gaussian."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

SAC_KW = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, alpha_lr=3e-4, soft_target_tau=0.005,
              alpha=0.2, train_alpha=True, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9)
SAC_KW_WALKER = dict(SAC_KW, reward_scale=2.0, beta_1=0.25, target_entropy=-4.0)


def _oracle():
    from oracle import mlp as omlp
    from oracle import tanh_gaussian as otg
    return omlp, otg


# ------------------------------------------------------------------------------------------- MLP
@pytest.mark.parametrize("tag", ["relu", "tanh"])
def test_mlp_forward_golden(ctx, tag):
    import ilswiss_amd as ia
    g = load_golden("g2_mlp")
    # golden nets are 32 wide; libilsx widths are 64/128/256 -> embed the 32-wide net in a 64-wide one
    omlp, _ = _oracle()
    lay = omlp.unpack(g[f"{tag}_params"], 14, [32, 32], 1)
    H = 64
    W0 = np.zeros((H, 14), np.float32); W0[:32] = lay[0][0]
    b0 = np.zeros(H, np.float32); b0[:32] = lay[0][1]
    W1 = np.zeros((H, H), np.float32); W1[:32, :32] = lay[1][0]
    b1 = np.zeros(H, np.float32); b1[:32] = lay[1][1]
    W2 = np.zeros((1, H), np.float32); W2[:, :32] = lay[2][0]
    flat = omlp.pack([(W0, b0), (W1, b1), (W2, lay[2][1])])
    net = ia.FlattenMlp([H, H], 1, 14, hidden_activation=tag, ctx=ctx)
    net.set_flat_params(flat)
    np.testing.assert_array_equal(net.get_flat_params(), flat)  # layout round trip
    y = net(g[f"{tag}_obs"], g[f"{tag}_act"])
    np.testing.assert_allclose(y, g[f"{tag}_y"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("H,nhid,in_dim,out_dim,rows,act", [
    (64, 1, 5, 2, 7, "relu"), (128, 2, 23, 1, 512, "tanh"), (256, 2, 14, 1, 256, "relu"),
    (256, 3, 393, 1, 100, "relu"), (256, 2, 376, 17, 33, "tanh"), (64, 2, 16, 64, 16, "relu")])
def test_mlp_forward_vs_oracle(ctx, H, nhid, in_dim, out_dim, rows, act):
    import ilswiss_amd as ia
    omlp, _ = _oracle()
    rng = np.random.default_rng(H + nhid + in_dim)
    flat = omlp.init_mlp(rng, in_dim, [H] * nhid, out_dim, init_w=0.1)
    net = ia.Mlp([H] * nhid, out_dim, in_dim, hidden_activation=act, ctx=ctx)
    assert net.num_params == flat.size
    net.set_flat_params(flat)
    x = rng.normal(0, 1, (rows, in_dim)).astype(np.float32)
    outs, _ = omlp.forward(flat, x, in_dim, [H] * nhid, out_dim, act=omlp.RELU if act == "relu" else omlp.TANH)
    np.testing.assert_allclose(net(x), outs[0], rtol=2e-5, atol=2e-5)


def test_net_init_rule(ctx):
    import ilswiss_amd as ia
    omlp, _ = _oracle()
    net = ia.FlattenMlp([256, 256], 1, 14, ctx=ctx, seed=3)
    lay = omlp.unpack(net.get_flat_params(), 14, [256, 256], 1)
    for l in (0, 1):  # networks.py:57-62 + pytorch_util.py:20-29: U(+-1/sqrt(out_features)), b = 0.1
        assert 0.9 / 16 < np.abs(lay[l][0]).max() <= 1 / 16 and np.allclose(lay[l][1], 0.1)
        assert abs(lay[l][0].mean()) < 2e-3
    assert np.abs(lay[2][0]).max() <= 3e-3 and np.abs(lay[2][1]).max() <= 3e-3


# ------------------------------------------------------------------------------------------- policy head
def _policy_with_identity_heads(ia, ctx, A, mu, ls_raw):
    """A policy whose heads output given (mu, ls_raw) for a one-hot style input: obs = [mu|ls_raw] (2A dims),
    hidden layer passes |x| through two relu units per input."""
    omlp, _ = _oracle()
    H, o = 64, 2 * A
    assert 2 * o <= H
    W0 = np.zeros((H, o), np.float32); b0 = np.zeros(H, np.float32)
    for i in range(o):
        W0[2 * i, i], W0[2 * i + 1, i] = 1.0, -1.0  # relu(x), relu(-x)
    Wm = np.zeros((A, H), np.float32); Ws = np.zeros((A, H), np.float32)
    for j in range(A):
        Wm[j, 2 * j], Wm[j, 2 * j + 1] = 1.0, -1.0
        Ws[j, 2 * (A + j)], Ws[j, 2 * (A + j) + 1] = 1.0, -1.0
    z = np.zeros(A, np.float32)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([H], o, A, ctx=ctx)
    pol.set_flat_params(omlp.pack([(W0, b0), (Wm, z), (Ws, z)]))
    return pol, np.concatenate([mu, ls_raw], 1).astype(np.float32)


def test_tanh_gaussian_head_golden(ctx):
    import ilswiss_amd as ia
    _, otg = _oracle()
    g = load_golden("g1_tanh_gaussian_head")
    A = g["mu"].shape[1]
    pol, obs = _policy_with_identity_heads(ia, ctx, A, g["mu"], g["log_std_raw"])
    act, mean, log_std, logp, *_ = pol.forward(obs, return_log_prob=True, eps=g["eps"])
    np.testing.assert_allclose(mean, g["mu"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(log_std, g["log_std_f32"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(act, g["action_f32"], rtol=1e-5, atol=2e-6)
    tol = otg.logp_fp32_tolerance(g["action_f32"], ulps=8.0)
    assert np.all(np.abs(logp - g["log_prob_f32"]) <= tol), np.abs(logp - g["log_prob_f32"]).max()
    # deterministic action = tanh(mean) (policies.py:274-275)
    det = pol.get_actions(obs, deterministic=True)
    np.testing.assert_allclose(det, np.tanh(g["mu"]), rtol=1e-5, atol=2e-6)
    # inverse path (policies.py:329-345) on non-saturated actions
    a_in = np.clip(g["action_f32"], -0.999, 0.999)
    lp = pol.get_log_prob(obs, a_in)
    ref = otg.log_prob_of_action(g["mu"], g["log_std_raw"], a_in, dtype=np.float64)
    ok = g["log_std_f32"].min(axis=1) > -5.0  # sigma = e^-20 rows: (mu-z)^2/sigma^2 ~ 1e7 is fp32-meaningless
    np.testing.assert_allclose(lp[ok], ref[ok], rtol=2e-4, atol=2e-3)


def test_policy_philox_noise_is_standard_normal(ctx):
    import ilswiss_amd as ia
    A, n = 6, 4096
    mu = np.zeros((n, A), np.float32)
    pol, obs = _policy_with_identity_heads(ia, ctx, A, mu, np.zeros((n, A), np.float32))  # std = 1
    a1 = pol.get_actions(obs)
    a2 = pol.get_actions(obs)
    z = np.arctanh(np.clip(a1, -0.9999999, 0.9999999))
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
    assert np.abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.05
    assert not np.allclose(a1, a2)  # fresh counter each call


# ------------------------------------------------------------------------------------------- SAC
def _make_agent(ia, ctx, o, a, hidden, pi0, q10, q20, kw, max_batch):
    pol = ia.ReparamTanhMultivariateGaussianPolicy(hidden, o, a, ctx=ctx)
    q1 = ia.FlattenMlp(hidden, 1, o + a, ctx=ctx)
    q2 = ia.FlattenMlp(hidden, 1, o + a, ctx=ctx)
    pol.set_flat_params(pi0), q1.set_flat_params(q10), q2.set_flat_params(q20)
    return ia.SoftActorCritic(pol, q1, q2, max_batch=max_batch, **kw), pol, q1, q2


def _rand_batch(rng, B, o, a):
    return dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32),
                actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                rewards=rng.normal(0, 1, (B, 1)).astype(np.float32),
                terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))


@pytest.mark.parametrize("o,a,H,nhid,B,kw", [
    (11, 3, 64, 2, 32, SAC_KW), (17, 6, 128, 2, 37, SAC_KW_WALKER), (11, 3, 256, 2, 256, SAC_KW),
    (376, 17, 256, 2, 64, dict(SAC_KW, target_entropy=-4.0)), (5, 2, 64, 1, 16, SAC_KW), (8, 2, 64, 3, 48, SAC_KW)])
def test_sac_steps_vs_oracle(ctx, o, a, H, nhid, B, kw):
    """5 chained SoftActorCritic.train_step calls: every intermediate the reference exposes."""
    import ilswiss_amd as ia
    from oracle.sac_alpha import SacAlphaOracle
    omlp, _ = _oracle()
    rng = np.random.default_rng(o * 100 + H + B)
    hidden = [H] * nhid
    pi0 = omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + a, hidden, 1), omlp.init_mlp(rng, o + a, hidden, 1)
    tr, pol, _, _ = _make_agent(ia, ctx, o, a, hidden, pi0, q10, q20, kw, B)
    orc = SacAlphaOracle(o, a, hidden, pi0, q10, q20, **kw)
    for s in range(5):
        batch = _rand_batch(rng, B, o, a)
        e1, e2 = rng.normal(0, 1, (B, a)).astype(np.float32), rng.normal(0, 1, (B, a)).astype(np.float32)
        tr.end_epoch()
        tr.train_step(batch, e1, e2)
        res = orc.train_step(batch, e1, e2)
        st = tr.get_eval_statistics()
        for k_ref, k_or in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss"),
                            ("Alpha Loss", "alpha_loss")):
            np.testing.assert_allclose(st[k_ref], res[k_or], rtol=2e-4, atol=2e-6, err_msg=f"{k_ref} step {s}")
        np.testing.assert_allclose(st["Q1 Predictions Mean"], res["q1_pred"].mean(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st["Log Pis Mean"], res["log_pi"].mean(), rtol=1e-4, atol=1e-5)
        for name, arr in (("Q1 Predictions", res["q1_pred"]), ("Q2 Predictions", res["q2_pred"]), ("Log Pis", res["log_pi"]),
                          ("Policy mu", res["policy_mean"]), ("Policy log std", res["policy_log_std"])):   # create_stats_ordered_dict
            np.testing.assert_allclose(st[name + " Std"], arr.std(), rtol=1e-4, atol=1e-6, err_msg=name)
            np.testing.assert_allclose(st[name + " Max"], arr.max(), rtol=1e-4, atol=1e-5, err_msg=name)
            np.testing.assert_allclose(st[name + " Min"], arr.min(), rtol=1e-4, atol=1e-5, err_msg=name)
        for nm, key in (("qf1", "q1_grad"), ("qf2", "q2_grad"), ("policy", "pi_grad")):
            got, ref = tr.get_grads(nm), res[key]
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (s, nm, np.abs(got - ref).max(), np.abs(ref).max())
        np.testing.assert_allclose(tr.log_alpha, orc.log_alpha[0], rtol=0, atol=1e-6)
        for nm, ov in (("policy", orc.pi), ("qf1", orc.q1), ("qf2", orc.q2), ("target_qf1", orc.tq1), ("target_qf2", orc.tq2)):
            np.testing.assert_allclose(tr.get_params(nm), ov, rtol=0, atol=5e-5, err_msg=f"{nm} step {s}")
    # the policy handle shares the agent's arena: acting uses the trained weights
    np.testing.assert_array_equal(pol.get_flat_params(), tr.get_params("policy"))


def _embed32(omlp, flat, in_dim, out_dim, n_heads, H=64):
    lay = omlp.unpack(flat, in_dim, [32, 32], out_dim, n_heads)
    W0 = np.zeros((H, in_dim), np.float32); W0[:32] = lay[0][0]
    b0 = np.zeros(H, np.float32); b0[:32] = lay[0][1]
    W1 = np.zeros((H, H), np.float32); W1[:32, :32] = lay[1][0]
    b1 = np.zeros(H, np.float32); b1[:32] = lay[1][1]
    heads = []
    for W, b in lay[2:]:
        Wh = np.zeros((out_dim, H), np.float32); Wh[:, :32] = W
        heads.append((Wh, b))
    return omlp.pack([(W0, b0), (W1, b1)] + heads)


def _extract32(omlp, flat, in_dim, out_dim, n_heads, H=64):
    lay = omlp.unpack(flat, in_dim, [H, H], out_dim, n_heads)
    out = [(lay[0][0][:32], lay[0][1][:32]), (lay[1][0][:32, :32], lay[1][1][:32])]
    out += [(W[:, :32], b) for W, b in lay[2:]]
    return omlp.pack(out)


@pytest.mark.parametrize("name,kw", [("g4_sac_alpha_small", SAC_KW), ("g4_sac_alpha_walker", SAC_KW_WALKER)])
def test_sac_golden_small(ctx, name, kw):
    """The reference's own numbers (32-wide nets embedded in 64-wide ones: the padding units have zero
    weights AND zero bias, so they stay exactly dead under relu + Adam)."""
    import ilswiss_amd as ia
    omlp, _ = _oracle()
    g = load_golden(name)
    o, a, B, steps = [int(v) for v in g["dims"][:4]]
    tr, *_ = _make_agent(ia, ctx, o, a, [64, 64], _embed32(omlp, g["pi0"], o, a, 2),
                         _embed32(omlp, g["q10"], o + a, 1, 1), _embed32(omlp, g["q20"], o + a, 1, 1), kw, B)
    for s in range(steps):
        batch = {k: g[f"s{s}_{k}"] for k in ("observations", "actions", "rewards", "terminals", "next_observations")}
        tr.end_epoch()
        tr.train_step(batch, g[f"s{s}_eps_next"], g[f"s{s}_eps_cur"])
        st = tr.get_eval_statistics()
        for k_ref, k_g in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss"),
                           ("Alpha Loss", "alpha_loss"), ("Q1 Predictions Mean", "q1_mean"),
                           ("Log Pis Mean", "log_pi_mean"), ("Policy mu Mean", "mu_mean"),
                           ("Policy log std Mean", "log_std_mean")):
            np.testing.assert_allclose(st[k_ref], g[k_g][s], rtol=2e-4, atol=2e-6, err_msg=f"{k_ref} step {s}")
        np.testing.assert_allclose(tr.log_alpha, g["log_alpha"][s], rtol=0, atol=1e-6)
        for nm, gk, ind, od, nh in (("qf1", "q1", o + a, 1, 1), ("qf2", "q2", o + a, 1, 1), ("policy", "pi", o, a, 2)):
            # gradients against the FLOAT64 run of the reference (tools/make_golden.py `tr64`): its fp32 autograd carries
            # cancellation noise on the log-prob path (Appendix A.1; up to 3e-5 of the largest entry in these fixtures), the float64
            # run does not, so one bound serves all three networks: 1e-4 of the largest entry (§8c "1e-4 chained")
            got = _extract32(omlp, tr.get_grads(nm), ind, od, nh)
            ref = g[f"s{s}_grad_{gk}_f64"]
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (s, nm, np.abs(got - ref).max() / np.abs(ref).max())
            ref32 = g[f"s{s}_grad_{gk}"]       # and the reference's own fp32 numbers at the same bound
            assert np.abs(got - ref32).max() <= 1e-4 * np.abs(ref32).max(), (s, nm)
        for nm, gk, ind, od, nh in (("qf1", "q1", o + a, 1, 1), ("qf2", "q2", o + a, 1, 1), ("policy", "pi", o, a, 2),
                                    ("target_qf1", "tq1", o + a, 1, 1), ("target_qf2", "tq2", o + a, 1, 1)):
            got = _extract32(omlp, tr.get_params(nm), ind, od, nh)
            np.testing.assert_allclose(got, g[f"s{s}_{gk}"], rtol=0, atol=5e-5, err_msg=f"{nm} step {s}")


def test_sac_golden_h256_b256(ctx):
    """BASELINE dims (Hopper, H=256, B=256): the reference's per-step losses and weight checksums."""
    import ilswiss_amd as ia
    omlp, _ = _oracle()
    g = load_golden("g4_sac_alpha_h256")
    o, a, B, steps = [int(v) for v in g["dims"][:4]]
    rng = np.random.default_rng(int(g["seed"]))
    hidden = [256, 256]
    pi0 = omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + a, hidden, 1), omlp.init_mlp(rng, o + a, hidden, 1)
    np.testing.assert_array_equal(pi0, g["pi0"])
    tr, *_ = _make_agent(ia, ctx, o, a, hidden, g["pi0"], g["q10"], g["q20"], SAC_KW, B)
    for s in range(steps):
        batch = _rand_batch(rng, B, o, a)
        e1, e2 = rng.normal(0, 1, (B, a)).astype(np.float32), rng.normal(0, 1, (B, a)).astype(np.float32)
        tr.end_epoch()
        tr.train_step(batch, e1, e2)
        st = tr.get_eval_statistics()
        for k_ref, k_g in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss"),
                           ("Alpha Loss", "alpha_loss")):
            np.testing.assert_allclose(st[k_ref], g[k_g][s], rtol=2e-4, atol=2e-6, err_msg=f"{k_ref} step {s}")
    for nm, gk in (("policy", "pi"), ("qf1", "q1"), ("qf2", "q2"), ("target_qf1", "tq1"), ("target_qf2", "tq2")):
        v = tr.get_params(nm)
        np.testing.assert_allclose(v[::97], g[f"final_{gk}_sample"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(v.astype(np.float64).sum(), g[f"final_{gk}_sum"], rtol=1e-4, atol=1e-2)


def test_sac_split_phases_equal_train_step(ctx):
    """critic_backward ; critic_update ; actor_backward ; actor_update == train_step (bitwise)."""
    import ilswiss_amd as ia
    omlp, _ = _oracle()
    rng = np.random.default_rng(5)
    o, a, hidden, B = 11, 3, [64, 64], 48
    pi0 = omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + a, hidden, 1), omlp.init_mlp(rng, o + a, hidden, 1)
    t1, *_ = _make_agent(ia, ctx, o, a, hidden, pi0, q10, q20, SAC_KW, B)
    t2, *_ = _make_agent(ia, ctx, o, a, hidden, pi0, q10, q20, SAC_KW, B)
    for _ in range(3):
        batch = _rand_batch(rng, B, o, a)
        e1, e2 = rng.normal(0, 1, (B, a)).astype(np.float32), rng.normal(0, 1, (B, a)).astype(np.float32)
        t1.train_step(batch, e1, e2)
        t2.set_batch(batch, e1, e2)
        t2.critic_backward(); t2.critic_update(); t2.actor_backward(); t2.actor_update()
    for nm in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
        np.testing.assert_array_equal(t1.get_params(nm), t2.get_params(nm))
    assert t1.log_alpha == t2.log_alpha


def test_sac_two_way_batch_split_matches_single(ctx):
    """SURVEY §8e parity oracle for the split-run mode: two half-batch agents with grad_world=2 whose
    gradient arenas are summed == one agent on the full batch (up to fp32 summation order)."""
    import ctypes as C

    import ilswiss_amd as ia
    omlp, _ = _oracle()
    rng = np.random.default_rng(6)
    o, a, hidden, B = 11, 3, [64, 64], 64
    pi0 = omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + a, hidden, 1), omlp.init_mlp(rng, o + a, hidden, 1)
    full, *_ = _make_agent(ia, ctx, o, a, hidden, pi0, q10, q20, SAC_KW, B)
    halves = [_make_agent(ia, ctx, o, a, hidden, pi0, q10, q20, dict(SAC_KW, grad_world=2), B // 2)[0] for _ in range(2)]
    batch = _rand_batch(rng, B, o, a)
    e1, e2 = rng.normal(0, 1, (B, a)).astype(np.float32), rng.normal(0, 1, (B, a)).astype(np.float32)
    full.train_step(batch, e1, e2)

    def allreduce(seg):  # host stand-in for ncclAllReduce(sum) over the two "ranks"
        views = [h.grads_view(seg) for h in halves]
        n = views[0].shape[0]
        bufs = []
        for v in views:
            host = np.empty(n, np.float32)
            ia._lib.check(ctx.lib.ilsx_memcpy_d2h(ctx.h, host.ctypes.data_as(C.c_void_p), C.c_void_p(v.ptr), host.nbytes))
            bufs.append(host)
        tot = bufs[0] + bufs[1]
        for v in views:
            ia._lib.check(ctx.lib.ilsx_memcpy_h2d(ctx.h, C.c_void_p(v.ptr), tot.ctypes.data_as(C.c_void_p), tot.nbytes))

    for r, h in enumerate(halves):
        sl = slice(r * B // 2, (r + 1) * B // 2)
        h.set_batch({k: v[sl] for k, v in batch.items()}, e1[sl], e2[sl])
        h.critic_backward()
    allreduce(0)
    for h in halves:
        h.critic_update()
        h.actor_backward()
    allreduce(1)
    for h in halves:
        h.actor_update()
    for nm in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
        np.testing.assert_allclose(halves[0].get_params(nm), full.get_params(nm), rtol=0, atol=2e-6)
        np.testing.assert_array_equal(halves[0].get_params(nm), halves[1].get_params(nm))
    np.testing.assert_allclose(halves[0].log_alpha, full.log_alpha, rtol=0, atol=1e-7)


def test_snapshot_roundtrip(ctx):
    import ilswiss_amd as ia
    omlp, _ = _oracle()
    rng = np.random.default_rng(8)
    o, a, hidden, B = 11, 3, [64, 64], 32
    pi0 = omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + a, hidden, 1), omlp.init_mlp(rng, o + a, hidden, 1)
    t1, *_ = _make_agent(ia, ctx, o, a, hidden, pi0, q10, q20, SAC_KW, B)
    b = [(_rand_batch(rng, B, o, a), rng.normal(0, 1, (B, a)).astype(np.float32), rng.normal(0, 1, (B, a)).astype(np.float32))
         for _ in range(4)]
    for x in b[:2]:
        t1.train_step(*x)
    snap = t1.get_snapshot()
    t2, *_ = _make_agent(ia, ctx, o, a, hidden, pi0 * 0, q10 * 0, q20 * 0, SAC_KW, B)
    t2.load_snapshot(snap)
    for x in b[2:]:
        t1.train_step(*x)
        t2.train_step(*x)
    for nm in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
        np.testing.assert_array_equal(t1.get_params(nm), t2.get_params(nm))
    assert t1.log_alpha == t2.log_alpha


# ------------------------------------------------------------------------------------------- replay
def test_replay_ring_semantics_golden(ctx):
    import ilswiss_amd as ia
    g = load_golden("g10_replay")
    rb = ia.SimpleReplayBuffer(int(g["cap"]), int(g["o"]), int(g["a"]), random_seed=1995, ctx=ctx)
    i = 0
    for k, n_after in enumerate(g["snap_n"]):
        while i < n_after:
            rb.add_sample(g["obs"][i], g["act"][i], g["rew"][i], int(g["term"][i]), g["next_obs"][i])
            i += 1
        rb.terminate_episode()
        assert rb._top == g["snap_top"][k] and rb._size == g["snap_size"][k]
        ends = np.array(sorted(rb._traj_endpoints.items()), dtype=np.int64).reshape(-1, 2)
        np.testing.assert_array_equal(ends, g[f"snap{k}_ends"])
    np.testing.assert_array_equal(list(rb._traj_endpoints.keys()), g["final_traj_starts"])
    np.testing.assert_array_equal(list(rb._traj_endpoints.values()), g["final_traj_ends"])
    b = rb._get_batch_using_indices(g["idx"])
    np.testing.assert_allclose(b["observations"], g["gather_obs"], atol=1e-6)
    np.testing.assert_allclose(b["actions"], g["gather_act"], atol=1e-6)
    np.testing.assert_allclose(b["rewards"], g["gather_rew"], atol=1e-6)
    np.testing.assert_array_equal(b["terminals"], g["gather_term"])
    np.testing.assert_allclose(b["next_observations"], g["gather_next_obs"], atol=1e-6)
    trajs = rb.sample_all_trajs()
    np.testing.assert_array_equal([len(t["rewards"]) for t in trajs], g["traj_lens"])
    np.testing.assert_allclose(np.concatenate([t["rewards"].ravel() for t in trajs]), g["traj_rew_concat"], atol=1e-6)
    # bursts with ep_end flags reproduce the same cursors
    rb2 = ia.SimpleReplayBuffer(int(g["cap"]), int(g["o"]), int(g["a"]), ctx=ctx)
    rb2.add_rows(g["obs"][:17], g["act"][:17], g["rew"][:17], g["term"][:17], g["next_obs"][:17], g["ep_end"][:17])
    for lo, hi in ((17, 37), (37, len(g["rew"]))):  # a burst may not exceed the capacity (23)
        rb2.add_rows(g["obs"][lo:hi], g["act"][lo:hi], g["rew"][lo:hi], g["term"][lo:hi], g["next_obs"][lo:hi],
                     g["ep_end"][lo:hi])
    assert rb2._traj_endpoints == rb._traj_endpoints and rb2._top == rb._top
    b2 = rb2._get_batch_using_indices(g["idx"])
    np.testing.assert_array_equal(b2["observations"], b["observations"])
    rb.clear()
    assert rb._size == 0 and rb._top == 0 and rb._traj_endpoints == {}


def test_replay_trajectory_sampling_golden(ctx, tmp_path):
    """g23 through the adapter: sample_trajs / samples_per_traj / get_all draw and gather what the reference does; save_data's layout."""
    import pickle
    import ilswiss_amd as ia
    g, t = load_golden("g10_replay"), load_golden("g23_replay_trajs")
    rb = ia.SimpleReplayBuffer(int(g["cap"]), int(g["o"]), int(g["a"]), random_seed=int(t["seed"]), ctx=ctx)
    rb.add_rows(g["obs"][:17], g["act"][:17], g["rew"][:17], g["term"][:17], g["next_obs"][:17], g["ep_end"][:17])
    for lo, hi in ((17, 37), (37, len(g["rew"]))):
        rb.add_rows(g["obs"][lo:hi], g["act"][lo:hi], g["rew"][lo:hi], g["term"][lo:hi], g["next_obs"][lo:hi], g["ep_end"][lo:hi])
    calls = [("sample_trajs", dict(num_trajs=3)), ("sample_trajs", dict(num_trajs=2, samples_per_traj=4)),
             ("sample_trajs", dict(num_trajs=9, samples_per_traj=12)), ("sample_all_trajs", dict(samples_per_traj=3)), ("get_all", {})]
    for ci, (fn, kw) in enumerate(calls):
        res = getattr(rb, fn)(**kw)
        res = res if isinstance(res, list) else [res]
        np.testing.assert_array_equal([len(x["rewards"]) for x in res], t[f"c{ci}_lens"])
        np.testing.assert_allclose(np.concatenate([x["observations"] for x in res]), t[f"c{ci}_obs"], atol=1e-6)
        np.testing.assert_allclose(np.concatenate([x["rewards"].ravel() for x in res]), t[f"c{ci}_rew"], atol=1e-6)
    rb.save_data(str(tmp_path / "buf.pkl"))
    d = pickle.load(open(tmp_path / "buf.pkl", "rb"))
    assert set(d) == {"observations", "actions", "next_observations", "terminals", "timeouts", "rewards", "agent_infos", "env_infos"}
    assert len(d["rewards"]) == rb._top == len(d["agent_infos"])


def test_replay_random_batch_index_stream_and_uniformity(ctx):
    import ilswiss_amd as ia
    from oracle.replay import ReplayOracle
    rng = np.random.default_rng(11)
    cap, o, a, n = 5000, 11, 3, 3000
    rb = ia.SimpleReplayBuffer(cap, o, a, random_seed=77, ctx=ctx)
    orc = ReplayOracle(cap, o, a, random_seed=77)
    obs = rng.normal(0, 1, (n, o)).astype(np.float32); nobs = rng.normal(0, 1, (n, o)).astype(np.float32)
    act = rng.normal(0, 1, (n, a)).astype(np.float32); rew = np.arange(n, dtype=np.float32)
    term = (rng.random(n) < 0.01).astype(np.uint8)
    rb.add_rows(obs, act, rew, term, nobs)
    orc.add_rows(obs, act, rew, term, nobs)
    for _ in range(3):  # same RandomState(seed).randint stream as the reference -> identical batches
        b, ob = rb.random_batch(256), orc.gather(orc.draw_indices(256))
        for k in ("observations", "actions", "rewards", "terminals", "next_observations"):
            np.testing.assert_array_equal(b[k], ob[k])
    # on-device Philox draw: uniform over [0,size), with replacement, fresh per call
    C = __import__("ctypes")
    B = 4096
    outs = [ctx.empty(s) for s in ((B, o), (B, a), (B,), (B,), (B, o))]
    idx = ctx.empty((B,), np.int64)
    draws = []
    for _ in range(2):
        ia._lib.check(ctx.lib.ilsx_replay_sample(rb.h, B, None, *[x.ptr for x in outs], idx.ptr))
        i = idx.numpy()
        np.testing.assert_array_equal(outs[2].numpy(), rew[i])       # gathered row == drawn index
        np.testing.assert_array_equal(outs[0].numpy(), obs[i])
        assert i.min() >= 0 and i.max() < n and len(np.unique(i)) < B  # with replacement
        hist = np.bincount(i * 10 // n, minlength=10)
        assert np.all(np.abs(hist - B / 10) < 5 * np.sqrt(B / 10))
        draws.append(i)
    assert not np.array_equal(draws[0], draws[1])


def test_replay_sample_many_records(ctx):
    import ilswiss_amd as ia
    rng = np.random.default_rng(12)
    cap, o, a = 2048, 11, 3
    rb = ia.SimpleReplayBuffer(cap, o, a, ctx=ctx)
    n = cap
    obs = rng.normal(0, 1, (n, o)).astype(np.float32)
    rew = np.arange(n, dtype=np.float32)
    rb.add_rows(obs, rng.normal(0, 1, (n, a)).astype(np.float32), rew, np.zeros(n, np.uint8), obs + 1)
    import ctypes as C
    rec = C.c_int()
    ia._lib.check(ctx.lib.ilsx_replay_record_floats(rb.h, C.byref(rec)))
    assert rec.value == 32  # Hopper: 27 floats -> one 128-byte record
    out = ctx.empty((8 * 256, rec.value))
    ia._lib.check(ctx.lib.ilsx_replay_sample_many(rb.h, 8, 256, out.ptr))
    r = out.numpy()
    rows = r[:, o + a].astype(np.int64)  # the reward column holds the row id
    np.testing.assert_array_equal(r[:, :o], obs[rows])
    np.testing.assert_allclose(r[:, o + a + 2: 2 * o + a + 2], obs[rows] + 1, rtol=0, atol=0)
    assert len(np.unique(rows)) > 0.5 * len(rows) * (1 - np.exp(-1))


def test_replay_absorbing_add_path_golden(ctx):
    """G19 (reference-generated): add_path(path, absorbing=True, env) on the HBM ring — rows, the absorbing flags, cursors,
    trajectory end points, wrap-around (simple_replay_buffer.py:163-213)."""
    import ilswiss_amd as ia
    g = load_golden("g19_absorbing")
    cap, o, a = int(g["cap"]), int(g["o"]), int(g["a"])
    rb = ia.SimpleReplayBuffer(cap, o, a, random_seed=7, ctx=ctx)
    stream = iter(g["acts_stream"])
    env = type("E", (), {})()
    env.action_space = type("S", (), dict(sample=lambda self: next(stream)))()
    for i in range(int(g["n_paths"])):
        rb.add_path({k: g[f"p{i}_{k}"] for k in ("observations", "actions", "rewards", "next_observations", "terminals")},
                    absorbing=True, env=env)
    assert rb._top == int(g["top"]) and rb._size == int(g["size"]) and rb.get_traj_num() == int(g["n_paths"])
    assert rb._traj_endpoints == dict(zip(g["traj_starts"].tolist(), g["traj_ends"].tolist()))
    b = rb._get_batch_using_indices(np.arange(cap))
    for k, gk in (("observations", "ring_obs"), ("actions", "ring_act"), ("rewards", "ring_rew"), ("terminals", "ring_term"),
                  ("next_observations", "ring_next_obs"), ("absorbing", "ring_absorbing")):
        np.testing.assert_allclose(np.asarray(b[k], np.float64), np.asarray(g[gk], np.float64), rtol=0, atol=1e-6, err_msg=k)
    # rows added without the keyword (and rows the fused rollout writes) carry [0, 0], also over old flags
    rb.add_rows(np.zeros((cap, o), np.float32), np.zeros((cap, a), np.float32), np.zeros(cap, np.float32), np.zeros(cap, np.uint8),
                np.zeros((cap, o), np.float32))
    assert not rb._get_batch_using_indices(np.arange(cap))["absorbing"].any()


# ------------------------------------------------------------------------------------------- fused loop
def test_train_from_replay_graph_equals_direct_and_learns(ctx, monkeypatch):
    """The captured-graph loop (on-device sampling) is deterministic given the seed, and fits a toy target."""
    import ilswiss_amd as ia
    omlp, _ = _oracle()
    rng = np.random.default_rng(13)
    o, a, hidden, B, n = 11, 3, [64, 64], 128, 4096
    obs = rng.normal(0, 1, (n, o)).astype(np.float32)
    act = np.tanh(rng.normal(0, 1, (n, a))).astype(np.float32)
    rew = (obs[:, 0] + act[:, 0]).astype(np.float32)  # learnable reward, terminal everywhere -> Q regresses r
    pi0 = omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + a, hidden, 1), omlp.init_mlp(rng, o + a, hidden, 1)
    finals = []
    for _ in range(2):
        c2 = ia.Context(0, seed=99)
        rb = ia.SimpleReplayBuffer(n, o, a, random_seed=5, ctx=c2)
        rb.add_rows(obs, act, rew, np.ones(n, np.uint8), obs)
        tr, *_ = _make_agent(ia, c2, o, a, hidden, pi0, q10, q20, dict(SAC_KW, qf_lr=3e-3), B)
        tr.train_from_replay(rb, 1, B)
        first = tr.get_eval_statistics()["QF1 Loss"]
        tr.end_epoch()
        tr.train_from_replay(rb, 300, B)
        tr.end_epoch()
        tr.train_from_replay(rb, 1, B)
        last = tr.get_eval_statistics()["QF1 Loss"]
        assert last < 0.2 * first, (first, last)
        finals.append(tr.get_params("qf1"))
        c2.close()
    np.testing.assert_array_equal(finals[0], finals[1])


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{}, {"ILSX_GRP_MT": "2", "ILSX_DW_GRP_STRIP": "1"}, {"ILSX_GRP_MT": "4"}, {"ILSX_DW_GRP_STRIP": "1"},
                                   {"ILSX_GRP_LATE": "1"}, {"ILSX_GRP_LATE": "0"}, {"ILSX_DW_TILE_GRP": "12"},
                                   {"ILSX_DW_TILE_GRP": "24", "ILSX_DW_GRP_LOW": "1"}, {"ILSX_DW_TILE_GRP": "24", "ILSX_DW_GRP_LOW": "0"},
                                   {"ILSX_GRP_LATE": "1", "_hid": "128"}, {"_hid": "128"}],
                         ids=["default", "mt2_strip", "mt4", "strip", "late", "early", "dw_tile12", "dw_tile24_low", "dw_tile24_high", "late_h128", "h128"])
@pytest.mark.parametrize("o,a,B", [(11, 3, 256), (111, 8, 256), (17, 6, 100)])   # narrow: one-launch forward ; Ant widths: the two-phase forward (kernels.h PH 1 / 2) ; a ragged batch
def test_sac_group_lockstep_is_bitwise_the_independent_runs(ctx, o, a, B, knobs, monkeypatch):
    """K=3 co-resident seeds stepped by ONE launch per stage (ilsx_sac_group) == each agent stepped alone with
    ilsx_sac_train_from_replay: same per-agent Philox streams, and per 16-row tile / per output element the same chain of MFMAs
    whatever the launch shape — macro tiles of 2 or 4 row tiles per workgroup on the XCD-stable 1-D grid (fwd_split_tile.inc MT,
    GrpSwizzle), weight gradients as one-wavefront strips (k_dw_strip) or as 8-wave tiles (the 32 x 64 tile in its 104-register
    and its 63-register, eight-waves-per-SIMD instance: k_mlp_bwd_dw_low) — so every parameter is bit-identical."""
    import ilswiss_amd as ia
    from ilswiss_amd.replay import SimpleReplayBuffer
    for k_, v_ in knobs.items():
        if not k_.startswith("_"):
            monkeypatch.setenv(k_, v_)
    hid, K, n = ([128, 128] if knobs.get("_hid") == "128" else [256, 256]), 3, 7
    rng = np.random.default_rng(5)
    N = 5000
    data = [(rng.normal(0, 1, (N, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (N, a))).astype(np.float32),
             rng.normal(0, 1, N).astype(np.float32), rng.random(N) < 0.01, rng.normal(0, 1, (N, o)).astype(np.float32)) for _ in range(K)]

    def make(k, ctx_k):
        rb = SimpleReplayBuffer(8192, o, a, random_seed=k, ctx=ctx_k)
        rb.add_rows(*data[k])
        pol = ia.ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx_k, seed=10 + k)
        q1, q2 = ia.FlattenMlp(hid, 1, o + a, ctx=ctx_k, seed=20 + k), ia.FlattenMlp(hid, 1, o + a, ctx=ctx_k, seed=30 + k)
        tr = ia.SoftActorCritic(pol, q1, q2, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
        return rb, tr
    # the per-object Philox stream ids come from the ctx's allocation counter: build both worlds in fresh, identically
    # ordered contexts so that agent k gets the same streams in both
    c1, c2 = ia.Context(0, seed=77), ia.Context(0, seed=77)
    solo, grouped = [make(k, c1) for k in range(K)], [make(k, c2) for k in range(K)]
    for rb, tr in solo:
        tr.eval_statistics = {}
        tr.train_from_replay(rb, n, B)
    grp = ia.SoftActorCriticGroup([tr for _, tr in grouped])
    for _, tr in grouped:
        tr.eval_statistics = {}
    grp.train_from_replay([rb for rb, _ in grouped], n - 1, B)
    for _, tr in grouped:
        tr.eval_statistics = None          # ask for statistics on the last step
    grp.train_from_replay([rb for rb, _ in grouped], 1, B)
    for (_, t1), (_, t2) in zip(solo, grouped):
        for name in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
            np.testing.assert_array_equal(t1.get_params(name), t2.get_params(name), err_msg=name)
        assert t1.log_alpha == t2.log_alpha
        assert np.isfinite(t2.eval_statistics["QF1 Loss"]) and t2.eval_statistics["QF1 Loss"] > 0
    assert np.abs(solo[0][1].get_params("policy") - solo[1][1].get_params("policy")).max() > 1e-4   # the seeds differ
    grp.close(); c1.close(); c2.close()


_DEFER_SCRIPT = r'''
import hashlib, json, sys
import numpy as np
sys.path.insert(0, ".")
import ilswiss_amd as ia
from ilswiss_amd.replay import SimpleReplayBuffer
o, a, hid, B, N = 11, 3, [256, 256], 256, 5000
rng = np.random.default_rng(5)
ctx = ia.Context(0, seed=77)
rb = SimpleReplayBuffer(8192, o, a, random_seed=3, ctx=ctx)
rb.add_rows(rng.normal(0, 1, (N, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (N, a))).astype(np.float32),
            rng.normal(0, 1, N).astype(np.float32), rng.random(N) < 0.01, rng.normal(0, 1, (N, o)).astype(np.float32))
pol = ia.ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=10)
q1, q2 = ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=20), ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=30)
tr = ia.SoftActorCritic(pol, q1, q2, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rb, 3, B)            # a call boundary in the middle: the pending tail is flushed and picked up again
tr.train_from_replay(rb, 1, B)
tr.eval_statistics = None                  # statistics of the last step
tr.train_from_replay(rb, 4, B)
h = hashlib.sha256()
for name in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
    h.update(np.ascontiguousarray(tr.get_params(name)).tobytes())
st = tr.eval_statistics
print(json.dumps(dict(params=h.hexdigest(), log_alpha=repr(tr.log_alpha), qf1=repr(float(st["QF1 Loss"])),
                      pl=repr(float(st["Policy Loss"])), lp=repr(float(st["Log Pis Mean"])))))
'''


def test_deferred_tail_is_bitwise_the_tail_launch():
    """Inside train_from_replay the step's tail (alpha Adam, counters, Adam scalars) runs one step late in an extra workgroup of
    the next step's first launch (TailLite); ILSX_NO_DEFER_TAIL=1 restores the tail launch.  Parameters, log-alpha and the
    statistics of the last step must agree bit for bit, with and without the hipGraph."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, extra in (("defer", {}), ("plain", {"ILSX_NO_DEFER_TAIL": "1"}), ("defer_nograph", {"ILSX_NO_GRAPH": "1"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _DEFER_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs["defer"] == outs["plain"], outs
    assert outs["defer"] == outs["defer_nograph"], outs


_CT_SCRIPT = r'''
import ctypes as C, hashlib, json, sys
import numpy as np
sys.path.insert(0, ".")
import ilswiss_amd as ia
from ilswiss_amd import _lib
from ilswiss_amd.replay import SimpleReplayBuffer
o, a, hid, B, N = 11, 3, [256, 256], 256, 5000
ctx = ia.Context(0, seed=77)
rbs = []
for k in range(2):
    rng = np.random.default_rng(5 + k)
    rb = SimpleReplayBuffer(8192, o, a, random_seed=3 + k, ctx=ctx)
    rb.add_rows(rng.normal(k, 1, (N, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (N, a))).astype(np.float32),
                rng.normal(k, 1, N).astype(np.float32), rng.random(N) < 0.01, rng.normal(k, 1, (N, o)).astype(np.float32))
    rbs.append(rb)
pol = ia.ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=10)
q1, q2 = ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=20), ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=30)
tr = ia.SoftActorCritic(pol, q1, q2, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rbs[0], 3, B)            # captured graph, ring 0: its phase launches read this agent's constant-memory slots
_lib.check(ctx.lib.ilsx_prof_enable(ctx.h, 1))
tr.train_from_replay(rbs[1], 2, B)            # direct launches (kernel timing on), ring 1: another state key, ITS blocks go into the same slots
_lib.check(ctx.lib.ilsx_prof_enable(ctx.h, 0))
tr.train_from_replay(rbs[0], 3, B)            # the cached graph again: the slots must hold ring 0's blocks when it runs
h = hashlib.sha256()
for name in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
    h.update(np.ascontiguousarray(tr.get_params(name)).tobytes())
fb = tr.phase_state()
print(json.dumps(dict(params=h.hexdigest(), log_alpha=repr(tr.log_alpha), on_phase=fb["last_window_on_phase"], fallbacks=fb["fallbacks"])))
'''


@pytest.mark.gpu
def test_phase_kernels_constant_slots_are_reprimed_before_a_cached_graph_replays():
    """The merged phase kernels read their descriptor blocks from per-agent constant-memory slots (kernels.h g_phase_a_tab / g_phase_c_tab).  A cached
    step graph is replayed across calls; in between, another step form of the SAME agent (here: direct launches on another ring) rewrites the slots —
    ilsx_sac_train_from_replay therefore re-primes them before every replay.  Same parameters as with the blocks in the argument segment
    (ILSX_PHASE_CT=0), bit for bit; and without the re-prime (ILSX_PHASE_CT_NO_REPRIME=1, a test aid) the result differs — the check has teeth."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, extra in (("ct", {}), ("args", {"ILSX_PHASE_CT": "0"}), ("stale", {"ILSX_PHASE_CT_NO_REPRIME": "1"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _CT_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs["ct"]["on_phase"] and outs["ct"]["fallbacks"] == 0, outs
    assert outs["ct"] == outs["args"], outs
    assert outs["stale"]["params"] != outs["ct"]["params"], outs


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["relu_200_100", "tanh_200_100", "relu_48_160_96", "tanh_100"])
def test_mlp_unequal_widths_forward_golden(ctx, tag):
    """networks.py:23-60 takes ANY hidden_sizes; the kernels run at 64 / 128 / 256 and the library embeds a narrower or unequal layer as
    structural zeros (include/ilsx.h ilsx_mlp_cfg::hidden_sizes).  The reference's own forward of [200, 100], [48, 160, 96] and [100] nets
    (g2b, reference-generated) must come out of FlattenMlp(hidden_sizes=...) with the LOGICAL parameter vector set and read back unchanged."""
    import ilswiss_amd as ia
    g = load_golden("g2b_mlp_unequal")
    Hh = [int(v) for v in g[f"{tag}_hidden"]]
    net = ia.FlattenMlp(Hh, 1, 14, hidden_activation=tag.split("_")[0], ctx=ctx)
    assert net.num_params == g[f"{tag}_params"].size and net.kernel_width == (256 if max(Hh) > 128 else 128)
    net.set_flat_params(g[f"{tag}_params"])
    np.testing.assert_array_equal(net.get_flat_params(), g[f"{tag}_params"])
    np.testing.assert_allclose(net(g[f"{tag}_obs"], g[f"{tag}_act"]), g[f"{tag}_y"], rtol=1e-5, atol=2e-6)
    fresh = ia.FlattenMlp(Hh, 1, 14, ctx=ctx, seed=4)       # the init rule on LOGICAL sizes: fanin bound 1 / sqrt(out features), biases 0.1
    omlp, _ = _oracle()
    lay = omlp.unpack(fresh.get_flat_params(), 14, Hh, 1)
    for l, h in enumerate(Hh):
        assert lay[l][0].shape == (h, 14 if l == 0 else Hh[l - 1]) and 0.85 / np.sqrt(h) < np.abs(lay[l][0]).max() <= 1 / np.sqrt(h) and np.allclose(lay[l][1], 0.1)


@pytest.mark.gpu
def test_sac_steps_with_unequal_widths_vs_oracle(ctx):
    """SAC on [200, 100] networks (policy and critics): five chained train steps land on the oracle's parameters — the structural zeros stay
    zero through Adam and the Polyak update (a unit that moved would show up as a parameter error), gradients within 1e-4 of the oracle's."""
    import ilswiss_amd as ia
    from oracle.sac_alpha import SacAlphaOracle
    omlp, _ = _oracle()
    o, a, Hh, B = 11, 3, [200, 100], 64
    rng = np.random.default_rng(71)
    pi0 = omlp.init_mlp(rng, o, Hh, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + a, Hh, 1), omlp.init_mlp(rng, o + a, Hh, 1)
    kw = dict(policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005)
    pol = ia.ReparamTanhMultivariateGaussianPolicy(Hh, o, a, ctx=ctx)
    q1, q2 = ia.FlattenMlp(Hh, 1, o + a, ctx=ctx), ia.FlattenMlp(Hh, 1, o + a, ctx=ctx)
    pol.set_flat_params(pi0), q1.set_flat_params(q10), q2.set_flat_params(q20)
    tr = ia.SoftActorCritic(pol, q1, q2, max_batch=B, **kw)
    orc = SacAlphaOracle(o, a, Hh, pi0, q10, q20, **kw)
    for s in range(5):
        batch = dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32), actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                     rewards=rng.normal(0, 1, (B, 1)).astype(np.float32), terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                     next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))
        e1, e2 = rng.normal(0, 1, (B, a)).astype(np.float32), rng.normal(0, 1, (B, a)).astype(np.float32)
        tr.eval_statistics = None
        tr.train_step(batch, e1, e2)
        res = orc.train_step(batch, e1, e2)
        for nm, key in (("qf1", "q1_grad"), ("qf2", "q2_grad"), ("policy", "pi_grad")):
            got, ref = tr.get_grads(nm), res[key]
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (s, nm, np.abs(got - ref).max() / np.abs(ref).max())
    st = tr.get_eval_statistics()
    np.testing.assert_allclose(st["QF1 Loss"], res["qf1_loss"], rtol=2e-4, atol=2e-6)
    for nm, ov in (("policy", orc.pi), ("qf1", orc.q1), ("qf2", orc.q2), ("target_qf1", orc.tq1), ("target_qf2", orc.tq2)):
        np.testing.assert_allclose(tr.get_params(nm), ov, rtol=0, atol=5e-5, err_msg=nm)
    snap = tr.get_snapshot()                          # optimiser state crosses the ABI in logical sizes too
    assert snap["qf1_optimizer"]["exp_avg"].size == q10.size


@pytest.mark.gpu
@pytest.mark.parametrize("o,a,B,nb", [(11, 3, 256, 8), (11, 3, 100, 3), (17, 6, 256, 5), (17, 6, 37, 7), (111, 8, 64, 3), (376, 17, 50, 2)],
                         ids=["hopper", "hopper_ragged", "walker", "walker_ragged", "ant_generic", "humanoid_generic"])
def test_replay_sample_many_draws_what_the_sample_kernel_draws(o, a, B, nb):
    """The bandwidth form (one index draw per row, shared across the row's 16-byte pieces through lane exchanges; 8- and 16-piece records as
    unrolled instances, any other width one record at a time) returns, batch by batch, exactly the rows ilsx_replay_sample draws call by call —
    same Philox counters — and whole records, also when the row count is not a multiple of a wavefront's 64."""
    import ctypes as C

    import ilswiss_amd as ia
    rng = np.random.default_rng(3)
    cap = 3000
    data = (rng.normal(0, 1, (cap, o)).astype(np.float32), rng.normal(0, 1, (cap, a)).astype(np.float32), np.arange(cap, dtype=np.float32),
            (rng.random(cap) < 0.1).astype(np.uint8), rng.normal(0, 1, (cap, o)).astype(np.float32))
    c1, c2 = ia.Context(0, seed=5), ia.Context(0, seed=5)
    rb1, rb2 = ia.SimpleReplayBuffer(cap, o, a, random_seed=9, ctx=c1), ia.SimpleReplayBuffer(cap, o, a, random_seed=9, ctx=c2)
    rb1.add_rows(*data), rb2.add_rows(*data)
    rec = C.c_int()
    ia._lib.check(c1.lib.ilsx_replay_record_floats(rb1.h, C.byref(rec)))
    bufs = [c1.empty((B, o)), c1.empty((B, a)), c1.empty((B,)), c1.empty((B,)), c1.empty((B, o)), c1.empty((B,), np.int64)]
    idx = []
    for _ in range(nb):
        ia._lib.check(c1.lib.ilsx_replay_sample(rb1.h, B, None, *[b.ptr for b in bufs]))
        idx.append(bufs[5].numpy().copy())
    idx = np.concatenate(idx)
    out = c2.empty((nb * B + 5, rec.value))
    out.copy_from(np.full((nb * B + 5, rec.value), -7.0, np.float32))
    ia._lib.check(c2.lib.ilsx_replay_sample_many(rb2.h, nb, B, out.ptr))
    r = out.numpy()
    np.testing.assert_array_equal(r[:nb * B, o + a].astype(np.int64), idx)                 # the reward column holds the row id
    np.testing.assert_array_equal(r[:nb * B, :o], data[0][idx])
    np.testing.assert_array_equal(r[:nb * B, o:o + a], data[1][idx])
    np.testing.assert_array_equal(r[:nb * B, o + a + 1], data[3][idx].astype(np.float32))
    np.testing.assert_array_equal(r[:nb * B, o + a + 2:2 * o + a + 2], data[4][idx])
    assert (r[nb * B:] == -7.0).all()                                                       # nothing written past the last row
    c1.close(); c2.close()
