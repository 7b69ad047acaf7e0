"""CPU suite: the oracle (our restatement) against the golden vectors produced by running the
reference (tools/make_golden.py).  This is what pins the oracle; the -m gpu tests then compare the
HIP path with the oracle and with the same fixtures."""
import numpy as np

from conftest import load_golden
from oracle import mlp as omlp
from oracle import optim
from oracle import tanh_gaussian as otg
from oracle.replay import ReplayOracle
from oracle.sac_alpha import SacAlphaOracle

SAC_KW = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, alpha_lr=3e-4, soft_target_tau=0.005,
              alpha=0.2, train_alpha=True, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9)
SAC_KW_WALKER = dict(SAC_KW, reward_scale=2.0, beta_1=0.25, target_entropy=-4.0)


def test_head_forward_and_inverse():
    g = load_golden("g1_tanh_gaussian_head")
    fw = otg.head_forward(g["mu"], g["log_std_raw"], g["eps"])
    tol = otg.logp_fp32_tolerance(g["action_f32"])
    assert np.all(np.abs(fw["log_prob"] - g["log_prob_f32"]) <= tol)
    np.testing.assert_allclose(fw["action"], g["action_f32"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fw["log_std"], g["log_std_f32"], rtol=0, atol=0)
    fw64 = otg.head_forward(g["mu"], g["log_std_raw"], g["eps"], dtype=np.float64)
    np.testing.assert_allclose(fw64["log_prob"], g["log_prob_f64"], rtol=1e-9, atol=1e-8)
    lp = otg.log_prob_of_action(g["mu"], g["log_std_raw"], g["action_f64"], dtype=np.float64)
    np.testing.assert_allclose(lp, g["log_prob_of_action_f64"], rtol=1e-9, atol=1e-7)


def test_head_backward_float64():
    g = load_golden("g1_tanh_gaussian_head")
    fw = otg.head_forward(g["mu"], g["log_std_raw"], g["eps"], dtype=np.float64)
    dmu, dls = otg.head_backward(fw, g["eps"], g["log_std_raw"], g["g_action"], g["g_logp"], dtype=np.float64)
    np.testing.assert_allclose(dmu, g["d_mu_f64"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dls, g["d_ls_raw_f64"], rtol=1e-6, atol=1e-6)
    # clamp gate: row 0 has raw log-std {-25, -20, 2, 2.5, 0, -19.999} -> gradient only inside [-20, 2]
    assert dls[0, 0] == 0 and dls[0, 3] == 0 and dls[0, 1] != 0 and dls[0, 2] != 0


def test_mlp_forward_backward():
    g = load_golden("g2_mlp")
    for tag, act in (("relu", omlp.RELU), ("tanh", omlp.TANH)):
        x = np.concatenate([g[f"{tag}_obs"], g[f"{tag}_act"]], 1)
        outs, hs = omlp.forward(g[f"{tag}_params"], x, 14, [32, 32], 1, act=act)
        np.testing.assert_allclose(outs[0], g[f"{tag}_y"], rtol=1e-5, atol=1e-6)
        grad, dx = omlp.backward(g[f"{tag}_params"], hs, [g[f"{tag}_gy"]], 14, [32, 32], 1, act=act)
        np.testing.assert_allclose(grad, g[f"{tag}_grad"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dx, g[f"{tag}_dx"], rtol=1e-4, atol=1e-6)


def test_mlp_unequal_widths_forward_backward():
    """g2b: the reference's Mlp with hidden_sizes the kernels have no width for ([200, 100], [48, 160, 96], [100]; networks.py:23-60)"""
    g = load_golden("g2b_mlp_unequal")
    for tag in ("relu_200_100", "tanh_200_100", "relu_48_160_96", "tanh_100"):
        act, Hh = (omlp.RELU if tag.startswith("relu") else omlp.TANH), [int(v) for v in g[f"{tag}_hidden"]]
        x = np.concatenate([g[f"{tag}_obs"], g[f"{tag}_act"]], 1)
        outs, hs = omlp.forward(g[f"{tag}_params"], x, 14, Hh, 1, act=act)
        np.testing.assert_allclose(outs[0], g[f"{tag}_y"], rtol=1e-5, atol=1e-6)
        grad, dx = omlp.backward(g[f"{tag}_params"], hs, [g[f"{tag}_gy"]], 14, Hh, 1, act=act)
        np.testing.assert_allclose(grad, g[f"{tag}_grad"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dx, g[f"{tag}_dx"], rtol=1e-4, atol=1e-6)


def test_init_rule_bounds():
    g = load_golden("g2_mlp")
    # reference: hidden W ~ U(+-1/sqrt(out_features)) = 1/16 for H=256 whatever the fan-in; b = 0.1
    assert 0.9 / 16 < g["init_fc0_w_absmax"] <= 1 / 16 and 0.9 / 16 < g["init_fc1_w_absmax"] <= 1 / 16
    assert np.allclose(g["init_fc0_b"], 0.1) and g["init_last_w_absmax"] <= 3e-3
    flat = omlp.init_mlp(np.random.default_rng(0), 14, [256, 256], 1)
    lay = omlp.unpack(flat, 14, [256, 256], 1)
    assert 0.9 / 16 < np.abs(lay[0][0]).max() <= 1 / 16 and np.allclose(lay[0][1], 0.1)
    assert np.abs(lay[2][0]).max() <= 3e-3


def _run_sac_case(name, kw, full):
    g = load_golden(name)
    o, a, B, steps = [int(v) for v in g["dims"][:4]]
    hidden = [int(v) for v in g["dims"][4:]]
    rng = np.random.default_rng(int(g["seed"]))
    # regenerate the exact input stream of tools/make_golden.py::_sac_case
    pi0 = omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2)
    q10 = omlp.init_mlp(rng, o + a, hidden, 1)
    q20 = omlp.init_mlp(rng, o + a, hidden, 1)
    np.testing.assert_array_equal(pi0, g["pi0"])
    orc = SacAlphaOracle(o, a, hidden, g["pi0"], g["q10"], g["q20"], **kw)
    for s in range(steps):
        batch = dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32),
                     actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                     rewards=rng.normal(0, 1, (B, 1)).astype(np.float32),
                     terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                     next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))
        e1 = rng.normal(0, 1, (B, a)).astype(np.float32)
        e2 = rng.normal(0, 1, (B, a)).astype(np.float32)
        if full:
            np.testing.assert_array_equal(batch["observations"], g[f"s{s}_observations"])
            np.testing.assert_array_equal(e2, g[f"s{s}_eps_cur"])
        res = orc.train_step(batch, e1, e2)
        for k in ("qf1_loss", "qf2_loss", "policy_loss", "alpha_loss"):
            np.testing.assert_allclose(res[k], g[k][s], rtol=2e-4, atol=1e-6, err_msg=f"{k} step {s}")
        np.testing.assert_allclose(orc.log_alpha[0], g["log_alpha"][s], rtol=0, atol=1e-6)
        np.testing.assert_allclose(res["q1_pred"].mean(), g["q1_mean"][s], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(res["log_pi"].mean(), g["log_pi_mean"][s], rtol=1e-4, atol=1e-5)
        if full:
            for nm, key in (("q1", "q1_grad"), ("q2", "q2_grad"), ("pi", "pi_grad")):
                ref = g[f"s{s}_grad_{nm}"]
                assert np.abs(res[key] - ref).max() <= 1e-4 * np.abs(ref).max(), (s, nm)
                ref64 = g[f"s{s}_grad_{nm}_f64"]    # the float64 run of the reference: the oracle sits at ~1e-7 of it
                assert np.abs(res[key] - ref64).max() <= 2e-6 * np.abs(ref64).max(), (s, nm)
            for nm in ("pi", "q1", "q2", "tq1", "tq2"):
                np.testing.assert_allclose(getattr(orc, nm), g[f"s{s}_{nm}"], rtol=0, atol=5e-5, err_msg=f"{nm} step {s}")
    for nm in ("pi", "q1", "q2", "tq1", "tq2"):
        v = getattr(orc, nm)
        if full:
            np.testing.assert_allclose(v, g["final_" + nm], rtol=0, atol=5e-5)
        else:
            np.testing.assert_allclose(v[::97], g[f"final_{nm}_sample"], rtol=0, atol=5e-5)
            np.testing.assert_allclose(v.astype(np.float64).sum(), g[f"final_{nm}_sum"], rtol=1e-4, atol=1e-2)


def test_sac_alpha_small():
    _run_sac_case("g4_sac_alpha_small", SAC_KW, True)


def test_sac_alpha_walker_gail_hparams():
    _run_sac_case("g4_sac_alpha_walker", SAC_KW_WALKER, True)


def test_sac_alpha_h256_b256():
    _run_sac_case("g4_sac_alpha_h256", SAC_KW, False)


def test_adam_matches_closed_form_first_step():
    p = np.array([1.0, -2.0], np.float32)
    st = optim.AdamState(2)
    optim.adam_step(p, np.array([0.5, -0.25], np.float32), st, lr=1e-3)
    # first Adam step moves every coordinate by lr*sign(g) (up to eps)
    np.testing.assert_allclose(p, [1.0 - 1e-3, -2.0 + 1e-3], rtol=0, atol=1e-7)


def replay_script(rb_add, rb_term, g):
    i = 0
    for k, n_after in enumerate(g["snap_n"]):
        while i < n_after:
            rb_add(i)
            i += 1
        rb_term()
        yield k


def test_replay_ring_semantics():
    g = load_golden("g10_replay")
    orc = ReplayOracle(int(g["cap"]), int(g["o"]), int(g["a"]))
    add = lambda i: orc.add_sample(g["obs"][i], g["act"][i], g["rew"][i], int(g["term"][i]), g["next_obs"][i])  # noqa
    for k in replay_script(add, orc.terminate_episode, g):
        assert orc.top == g["snap_top"][k] and orc.size == g["snap_size"][k]
        ends = np.array(sorted(orc.traj_endpoints.items()), dtype=np.int64).reshape(-1, 2)
        np.testing.assert_array_equal(ends, g[f"snap{k}_ends"])
    np.testing.assert_array_equal(list(orc.traj_endpoints.keys()), g["final_traj_starts"])
    np.testing.assert_array_equal(list(orc.traj_endpoints.values()), g["final_traj_ends"])
    b = orc.gather(g["idx"])
    np.testing.assert_allclose(b["observations"], g["gather_obs"], atol=1e-6)
    np.testing.assert_array_equal(b["terminals"], g["gather_term"])
    trajs = orc.sample_all_trajs()
    np.testing.assert_array_equal([len(t["rewards"]) for t in trajs], g["traj_lens"])
    np.testing.assert_allclose(np.concatenate([t["rewards"].ravel() for t in trajs]), g["traj_rew_concat"], atol=1e-6)
    # the oracle's index stream is numpy's legacy RandomState, like the reference (simple_replay_buffer.py:20,242)
    np.testing.assert_array_equal(np.random.RandomState(1995).randint(0, 1000, 8), g["randint_1995"])
    # add_rows == the same sequence issued as bursts with ep_end flags
    orc2 = ReplayOracle(int(g["cap"]), int(g["o"]), int(g["a"]))
    orc2.add_rows(g["obs"], g["act"], g["rew"], g["term"], g["next_obs"], g["ep_end"])
    assert orc2.traj_endpoints == orc.traj_endpoints and orc2.top == orc.top


def test_replay_trajectory_sampling_golden():
    """g23: sample_trajs / sample_all_trajs(samples_per_traj) / get_all — same draws from RandomState(seed) as the reference's."""
    g, t = load_golden("g10_replay"), load_golden("g23_replay_trajs")
    orc = ReplayOracle(int(g["cap"]), int(g["o"]), int(g["a"]), random_seed=int(t["seed"]))
    for i in range(len(g["rew"])):
        orc.add_sample(g["obs"][i], g["act"][i], g["rew"][i], int(g["term"][i]), g["next_obs"][i])
        if g["ep_end"][i]:
            orc.terminate_episode()
    calls = [("sample_trajs", dict(num_trajs=3)), ("sample_trajs", dict(num_trajs=2, samples_per_traj=4)),
             ("sample_trajs", dict(num_trajs=9, samples_per_traj=12)), ("sample_all_trajs", dict(samples_per_traj=3)), ("get_all", {})]
    for ci, (fn, kw) in enumerate(calls):
        res = getattr(orc, fn)(**kw)
        res = res if isinstance(res, list) else [res]
        np.testing.assert_array_equal([len(x["rewards"]) for x in res], t[f"c{ci}_lens"])
        np.testing.assert_allclose(np.concatenate([x["observations"] for x in res]), t[f"c{ci}_obs"], atol=1e-6)


def test_running_mean_std_and_action_map():
    from oracle.envnorm import RunningMeanStd, action_map, normalize_obs
    g = load_golden("g11_g12_rms_actionmap")
    rms = RunningMeanStd()
    for k in range(4):
        rms.update(g[f"x{k}"])
        np.testing.assert_allclose(rms.mean, g["means"][k], rtol=1e-12)
        np.testing.assert_allclose(rms.var, g["vars"][k], rtol=1e-12)
        assert rms.count == g["counts"][k]
    np.testing.assert_allclose(normalize_obs(g["probe"], rms), g["normed"], rtol=1e-12)
    np.testing.assert_allclose(action_map(g["acts"], g["lb"], g["ub"]), g["scaled"], rtol=1e-12)


def test_oracle_absorbing_add_path_golden():
    """G19: the wrap_absorbing branch of SimpleReplayBuffer.add_path (simple_replay_buffer.py:163-213), ring + cursors."""
    from oracle.replay import ReplayOracle
    g = load_golden("g19_absorbing")
    orc = ReplayOracle(int(g["cap"]), int(g["o"]), int(g["a"]))
    it = iter(g["acts_stream"])
    for i in range(int(g["n_paths"])):
        orc.add_path({k: g[f"p{i}_{k}"] for k in ("observations", "actions", "rewards", "next_observations", "terminals")},
                     absorbing=True, sample_action=lambda: next(it))
    assert orc.top == int(g["top"]) and orc.size == int(g["size"])
    assert list(orc.traj_endpoints.keys()) == list(g["traj_starts"]) and list(orc.traj_endpoints.values()) == list(g["traj_ends"])
    b = orc.gather(np.arange(int(g["cap"])))
    for k, gk in (("observations", "ring_obs"), ("actions", "ring_act"), ("rewards", "ring_rew"), ("terminals", "ring_term"),
                  ("next_observations", "ring_next_obs"), ("absorbing", "ring_absorbing")):
        np.testing.assert_allclose(np.asarray(b[k], np.float64), np.asarray(g[gk], np.float64), rtol=0, atol=1e-6, err_msg=k)
    assert g["ring_absorbing"].sum() > 0 and not g["ring_term"].any()


def test_torch_cpu_restatement_matches_the_reference_vectors():
    """oracle/sac_alpha_torch.py (the PyTorch-CPU restatement bench.py times) on the reference's own g4 vectors."""
    import torch
    from oracle.sac_alpha_torch import SacAlphaTorch
    torch.set_num_threads(1)
    for name, kw in (("g4_sac_alpha_small", SAC_KW), ("g4_sac_alpha_walker", SAC_KW_WALKER)):
        g = load_golden(name)
        o, a, B, steps = [int(v) for v in g["dims"][:4]]
        hidden = [int(v) for v in g["dims"][4:]]
        ag = SacAlphaTorch(o, a, hidden, g["pi0"], g["q10"], g["q20"], **kw)
        for s in range(steps):
            batch = {k: g[f"s{s}_{k}"] for k in ("observations", "actions", "rewards", "terminals", "next_observations")}
            res = ag.train_step(batch, g[f"s{s}_eps_next"], g[f"s{s}_eps_cur"])
            for k in ("qf1_loss", "qf2_loss", "policy_loss", "alpha_loss"):
                np.testing.assert_allclose(res[k], g[k][s], rtol=2e-4, atol=1e-6, err_msg=f"{name} {k} step {s}")
            np.testing.assert_allclose(float(ag.log_alpha), g["log_alpha"][s], rtol=0, atol=1e-6)
            for nm in ("pi", "q1", "q2", "tq1", "tq2"):
                np.testing.assert_allclose(ag.flat(nm), g[f"s{s}_{nm}"], rtol=0, atol=5e-5, err_msg=f"{name} {nm} step {s}")
