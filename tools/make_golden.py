#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (survey container only).

    python tools/make_golden.py            # all fixtures
    python tools/make_golden.py sac_alpha  # one group

The reference checkout (/root/reference, read-only) is imported through tools/_ref_harness.py;
inputs are drawn from seeded numpy generators, noise is injected (torch.randn wrapped), and only
plain arrays (inputs + the reference's outputs) are written.  While generating, every fixture is
also cross-checked against our restatement in oracle/ so a broken oracle is caught here.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_harness as H  # noqa: E402

H.install()
from oracle import mlp as omlp  # noqa: E402
from oracle import tanh_gaussian as otg  # noqa: E402

# ILSX_GOLDEN_OUT=<dir>: write somewhere else (tests/test_make_golden.py regenerates every fixture into a temp dir and compares it with
# tests/golden/ array for array)
GOLD = os.environ.get("ILSX_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(1)


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def n(x):
    return x.detach().cpu().numpy().copy()


def set_flat(module, flat):
    """Load a flat fp32 vector (oracle/mlp.py layout == parameters() order) into a torch module."""
    off = 0
    with torch.no_grad():
        for p in module.parameters():
            k = p.numel()
            p.copy_(t(flat[off:off + k]).view_as(p))
            off += k
    assert off == flat.size


def get_flat(module):
    return np.concatenate([n(p).ravel() for p in module.parameters()]).astype(np.float32)


def get_flat_grad(module):
    return np.concatenate([n(p.grad).ravel() for p in module.parameters()]).astype(np.float32)


# --------------------------------------------------------------------------------------------------
def gen_head():
    """G1: tanh-Gaussian head forward + float64 gradients + get_log_prob inverse path."""
    from rlkit.torch.common.distributions import ReparamTanhMultivariateNormal
    rng = np.random.default_rng(101)
    B, A = 64, 6
    mu = rng.normal(0, 1.0, (B, A)).astype(np.float32)
    ls_raw = rng.normal(-1.0, 1.5, (B, A)).astype(np.float32)
    ls_raw[0, :] = [-25.0, -20.0, 2.0, 2.5, 0.0, -19.999]  # clamp edges (policies.py:15-16,267)
    eps = rng.normal(0, 1, (B, A)).astype(np.float32)
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = t(mu).to(dt).requires_grad_(True)
        lr = t(ls_raw).to(dt).requires_grad_(True)
        ls = torch.clamp(lr, -20, 2)
        with H.NoiseInjector() as inj:
            inj.push(t(eps).to(dt))
            dist = ReparamTanhMultivariateNormal(m, ls)
            a, z = dist.sample(return_pretanh_value=True)
        lp = dist.log_prob(a, pre_tanh_value=z)
        g_a = t(rng.normal(0, 1, (B, A)).astype(np.float32)).to(dt) if tag == "f32" else g_a64
        if tag == "f32":
            g_a64 = g_a.double()
            g_lp32 = t(rng.normal(0, 1, (B, 1)).astype(np.float32))
        g_lp = g_lp32.to(dt)
        ((a * g_a).sum() + (lp * g_lp).sum()).backward()
        out.update({f"action_{tag}": n(a), f"z_{tag}": n(z), f"log_std_{tag}": n(ls),
                    f"log_prob_{tag}": n(lp), f"d_mu_{tag}": n(m.grad), f"d_ls_raw_{tag}": n(lr.grad)})
        # inverse path (policies.py:329-345): log_prob of given actions
        dist2 = ReparamTanhMultivariateNormal(t(mu).to(dt), torch.clamp(t(ls_raw).to(dt), -20, 2))
        out[f"log_prob_of_action_{tag}"] = n(dist2.log_prob(a.detach()))
    out.update(mu=mu, log_std_raw=ls_raw, eps=eps, g_action=n(g_a64).astype(np.float32), g_logp=n(g_lp32))
    # oracle cross-check
    fw = otg.head_forward(mu, ls_raw, eps)
    # fp32 log(1 - a^2 + 1e-6) amplifies a 1-ulp tanh difference by 2|a|/(1-a^2+1e-6): per-row tolerance
    tol = otg.logp_fp32_tolerance(out["action_f32"])
    assert np.all(np.abs(fw["log_prob"] - out["log_prob_f32"]) <= tol), "head fwd mismatch"
    fw64 = otg.head_forward(mu, ls_raw, eps, dtype=np.float64)
    dmu, dls = otg.head_backward(fw64, eps, ls_raw, out["g_action"], out["g_logp"], dtype=np.float64)
    # autograd keeps the two cancelling +-eps/sigma terms (sigma down to e^-20): 1e-7 residual even in f64
    assert np.allclose(dmu, out["d_mu_f64"], rtol=1e-6, atol=1e-6), np.abs(dmu - out["d_mu_f64"]).max()
    assert np.allclose(dls, out["d_ls_raw_f64"], rtol=1e-6, atol=1e-6)
    save("g1_tanh_gaussian_head", **out)


def gen_mlp():
    """G2: Mlp / FlattenMlp forward + backward, relu and tanh."""
    import torch.nn.functional as F
    from rlkit.torch.common.networks import FlattenMlp
    rng = np.random.default_rng(202)
    out = {}
    for tag, act, fn in (("relu", omlp.RELU, F.relu), ("tanh", omlp.TANH, torch.tanh)):
        o, a, Hh, B = 11, 3, [32, 32], 24
        flat = omlp.init_mlp(rng, o + a, Hh, 1)
        net = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1, hidden_activation=fn)
        set_flat(net, flat)
        obs = rng.normal(0, 1, (B, o)).astype(np.float32)
        actn = np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32)
        to, ta = t(obs).requires_grad_(True), t(actn).requires_grad_(True)
        y = net(to, ta)
        gy = rng.normal(0, 1, (B, 1)).astype(np.float32)
        (y * t(gy)).sum().backward()
        out.update({f"{tag}_params": flat, f"{tag}_obs": obs, f"{tag}_act": actn, f"{tag}_y": n(y),
                    f"{tag}_gy": gy, f"{tag}_grad": get_flat_grad(net),
                    f"{tag}_dx": np.concatenate([n(to.grad), n(ta.grad)], 1)})
        x = np.concatenate([obs, actn], 1)
        outs, hs = omlp.forward(flat, x, o + a, Hh, 1, act=act)
        assert np.allclose(outs[0], out[f"{tag}_y"], rtol=1e-5, atol=1e-6)
        g, dx = omlp.backward(flat, hs, [gy], o + a, Hh, 1, act=act)
        assert np.allclose(g, out[f"{tag}_grad"], rtol=1e-4, atol=1e-6)
        assert np.allclose(dx, out[f"{tag}_dx"], rtol=1e-4, atol=1e-6)
    # G3 init statistics of the reference rule (bounds, not bit values); torch's generator is seeded so that the file reproduces
    torch.manual_seed(20240)
    net = FlattenMlp(hidden_sizes=[256, 256], input_size=14, output_size=1)
    ps = [n(p) for p in net.parameters()]
    out["init_fc0_w_absmax"] = np.abs(ps[0]).max()
    out["init_fc1_w_absmax"] = np.abs(ps[2]).max()
    out["init_fc0_b"] = ps[1][:4]
    out["init_last_w_absmax"] = np.abs(ps[4]).max()
    save("g2_mlp", **out)


def gen_mlp_unequal():
    """G2b: Mlp / FlattenMlp with hidden_sizes the kernels have no width for (networks.py:23-60 takes any list): [200, 100] and
    [48, 160, 96]: forward, parameter gradients, input gradients, relu and tanh — what the library's structural-zero embedding has to match."""
    import torch.nn.functional as F
    from rlkit.torch.common.networks import FlattenMlp
    rng = np.random.default_rng(2020)
    out = {}
    for tag, act, fn, Hh in (("relu_200_100", omlp.RELU, F.relu, [200, 100]), ("tanh_200_100", omlp.TANH, torch.tanh, [200, 100]),
                             ("relu_48_160_96", omlp.RELU, F.relu, [48, 160, 96]), ("tanh_100", omlp.TANH, torch.tanh, [100])):
        o, a, B = 11, 3, 24
        flat = omlp.init_mlp(rng, o + a, Hh, 1, init_w=0.3)
        net = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1, hidden_activation=fn)
        set_flat(net, flat)
        obs = rng.normal(0, 1, (B, o)).astype(np.float32)
        actn = np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32)
        to, ta = t(obs).requires_grad_(True), t(actn).requires_grad_(True)
        y = net(to, ta)
        gy = rng.normal(0, 1, (B, 1)).astype(np.float32)
        (y * t(gy)).sum().backward()
        out.update({f"{tag}_params": flat, f"{tag}_obs": obs, f"{tag}_act": actn, f"{tag}_y": n(y), f"{tag}_gy": gy, f"{tag}_grad": get_flat_grad(net),
                    f"{tag}_dx": np.concatenate([n(to.grad), n(ta.grad)], 1), f"{tag}_hidden": np.array(Hh)})
        outs, hs = omlp.forward(flat, np.concatenate([obs, actn], 1), o + a, Hh, 1, act=act)
        assert np.allclose(outs[0], out[f"{tag}_y"], rtol=1e-5, atol=1e-6)
        g, dx = omlp.backward(flat, hs, [gy], o + a, Hh, 1, act=act)
        assert np.allclose(g, out[f"{tag}_grad"], rtol=1e-4, atol=1e-6) and np.allclose(dx, out[f"{tag}_dx"], rtol=1e-4, atol=1e-6)
    save("g2b_mlp_unequal", **out)


class _Env:  # dummy `env` kwarg for sac_alpha.py:55-58
    def __init__(self, a):
        self.action_space = type("S", (), {"shape": (a,)})()


def _sac_case(seed, o, a, Hh, B, steps, kwargs, full):
    from rlkit.torch.algorithms.sac.sac_alpha import SoftActorCritic
    from rlkit.torch.common.networks import FlattenMlp
    from rlkit.torch.common.policies import ReparamTanhMultivariateGaussianPolicy
    from oracle.sac_alpha import SacAlphaOracle

    rng = np.random.default_rng(seed)
    pi0 = omlp.init_mlp(rng, o, Hh, a, init_w=1e-3, n_heads=2)
    q10 = omlp.init_mlp(rng, o + a, Hh, 1)
    q20 = omlp.init_mlp(rng, o + a, Hh, 1)
    # make the head outputs non-trivial so log_std / tanh saturate a little
    pol = ReparamTanhMultivariateGaussianPolicy(hidden_sizes=Hh, obs_dim=o, action_dim=a)
    qf1 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1)
    qf2 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1)
    set_flat(pol, pi0), set_flat(qf1, q10), set_flat(qf2, q20)
    tr = SoftActorCritic(policy=pol, qf1=qf1, qf2=qf2, env=_Env(a), **kwargs)
    orc = SacAlphaOracle(o, a, Hh, pi0, q10, q20, **kwargs)

    grads = {}

    def hook(opt, name, mod):
        orig = opt.step

        def step(*aa, **kk):
            grads[name] = get_flat_grad(mod)
            return orig(*aa, **kk)
        opt.step = step
    hook(tr.qf1_optimizer, "q1", qf1), hook(tr.qf2_optimizer, "q2", qf2), hook(tr.policy_optimizer, "pi", pol)

    # float64 shadow of the reference (the fp32 autograd of the log-prob path carries ~1e-4 of cancellation
    # noise, so gradient fixtures are "compared against a float64 run of the reference"): the SAME reference classes with double
    # modules; before every step its parameters, targets, log_alpha and optimiser states are loaded from the fp32 run, it takes the
    # step on the same batch / noise, and only the gradients its optimisers see are kept.  Draws nothing from `rng`.
    grads64 = {}
    if full:
        pol64 = ReparamTanhMultivariateGaussianPolicy(hidden_sizes=Hh, obs_dim=o, action_dim=a).double()
        qf164 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1).double()
        qf264 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1).double()
        tr64 = SoftActorCritic(policy=pol64, qf1=qf164, qf2=qf264, env=_Env(a), **kwargs)
        tr64.target_qf1.double(), tr64.target_qf2.double()     # .copy() rebuilds the targets as float modules

        def hook64(opt, name, mod):
            orig = opt.step

            def step(*aa, **kk):
                grads64[name] = np.concatenate([p.grad.detach().numpy().ravel() for p in mod.parameters()]).astype(np.float64)
                return orig(*aa, **kk)
            opt.step = step
        hook64(tr64.qf1_optimizer, "q1", qf164), hook64(tr64.qf2_optimizer, "q2", qf264), hook64(tr64.policy_optimizer, "pi", pol64)

        def sync64():
            with torch.no_grad():
                for dst, src in ((pol64, pol), (qf164, qf1), (qf264, qf2), (tr64.target_qf1, tr.target_qf1), (tr64.target_qf2, tr.target_qf2)):
                    for pd, ps in zip(dst.parameters(), src.parameters()):
                        pd.copy_(ps.double())
                tr64.log_alpha.copy_(tr.log_alpha)
            tr64.alpha = tr.alpha.clone().double() if torch.is_tensor(tr.alpha) else tr.alpha
            for od, os_ in ((tr64.qf1_optimizer, tr.qf1_optimizer), (tr64.qf2_optimizer, tr.qf2_optimizer),
                            (tr64.policy_optimizer, tr.policy_optimizer), (tr64.alpha_optimizer, tr.alpha_optimizer)):
                # deep copy first: load_state_dict keeps the `step` tensors by reference, and the shadow's step() would advance the
                # fp32 run's counters in place; the load casts exp_avg / exp_avg_sq to the double parameters' dtype
                od.load_state_dict(copy.deepcopy(os_.state_dict()))

    rec = dict(pi0=pi0, q10=q10, q20=q20, dims=np.array([o, a, B, steps] + list(Hh)))
    scal = {k: [] for k in ("qf1_loss", "qf2_loss", "policy_loss", "alpha_loss", "log_alpha",
                            "q1_mean", "q2_mean", "log_pi_mean", "mu_mean", "log_std_mean")}
    for s in range(steps):
        batch = dict(
            observations=rng.normal(0, 1, (B, o)).astype(np.float32),
            actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
            rewards=rng.normal(0, 1, (B, 1)).astype(np.float32),
            terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
            next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))
        e1 = rng.normal(0, 1, (B, a)).astype(np.float32)
        e2 = rng.normal(0, 1, (B, a)).astype(np.float32)
        if full:
            sync64()
            with H.NoiseInjector() as inj:
                inj.push(e1.astype(np.float64)), inj.push(e2.astype(np.float64))
                tr64.train_step({k: t(v).double() for k, v in batch.items()})
        tr.eval_statistics = None
        with H.NoiseInjector() as inj:
            inj.push(e1), inj.push(e2)
            tr.train_step({k: t(v) for k, v in batch.items()})
        st = tr.eval_statistics
        res = orc.train_step(batch, e1, e2)
        scal["qf1_loss"].append(st["QF1 Loss"]); scal["qf2_loss"].append(st["QF2 Loss"])
        scal["policy_loss"].append(st["Policy Loss"]); scal["alpha_loss"].append(st["Alpha Loss"])
        scal["log_alpha"].append(float(tr.log_alpha.detach()))
        scal["q1_mean"].append(st["Q1 Predictions Mean"]); scal["q2_mean"].append(st["Q2 Predictions Mean"])
        scal["log_pi_mean"].append(st["Log Pis Mean"]); scal["mu_mean"].append(st["Policy mu Mean"])
        scal["log_std_mean"].append(st["Policy log std Mean"])
        # oracle cross-check, per step
        for k_ref, k_or in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss"),
                            ("Alpha Loss", "alpha_loss")):
            assert np.allclose(st[k_ref], res[k_or], rtol=2e-4, atol=1e-6), (s, k_ref, st[k_ref], res[k_or])
        for nm, g in (("q1", res["q1_grad"]), ("q2", res["q2_grad"]), ("pi", res["pi_grad"])):
            err = np.abs(grads[nm] - g).max() / (np.abs(grads[nm]).max() + 1e-12)
            assert err < 2e-3, (s, nm, err)
        if full or s == 0:
            rec.update({f"s{s}_{k}": v for k, v in batch.items()})
            rec.update({f"s{s}_eps_next": e1, f"s{s}_eps_cur": e2})
        if full:
            for nm, g in (("q1", res["q1_grad"]), ("q2", res["q2_grad"]), ("pi", res["pi_grad"])):   # the oracle against the float64 reference
                err = np.abs(grads64[nm] - g).max() / np.abs(grads64[nm]).max()
                assert err < 1e-4, (s, nm, err)
                print(f"   step {s} {nm}: oracle vs f64 reference {err:.2e} ; fp32 reference vs f64 reference "
                      f"{np.abs(grads64[nm] - grads[nm]).max() / np.abs(grads64[nm]).max():.2e}")
            rec.update({f"s{s}_grad_q1_f64": grads64["q1"], f"s{s}_grad_q2_f64": grads64["q2"], f"s{s}_grad_pi_f64": grads64["pi"]})
            rec.update({f"s{s}_grad_q1": grads["q1"], f"s{s}_grad_q2": grads["q2"], f"s{s}_grad_pi": grads["pi"],
                        f"s{s}_pi": get_flat(pol), f"s{s}_q1": get_flat(qf1), f"s{s}_q2": get_flat(qf2),
                        f"s{s}_tq1": get_flat(tr.target_qf1), f"s{s}_tq2": get_flat(tr.target_qf2)})
    fin = dict(pi=get_flat(pol), q1=get_flat(qf1), q2=get_flat(qf2), tq1=get_flat(tr.target_qf1),
               tq2=get_flat(tr.target_qf2))
    for k, v in fin.items():
        ov = getattr(orc, k)
        err = np.abs(ov - v).max()
        assert err < 5e-5, (k, err)
        if full:
            rec["final_" + k] = v
        else:  # checksums + strided sample keep the H=256 fixture small
            rec["final_" + k + "_sum"] = np.float64(v.astype(np.float64).sum())
            rec["final_" + k + "_abssum"] = np.float64(np.abs(v.astype(np.float64)).sum())
            rec["final_" + k + "_sample"] = v[::97].copy()
    rec.update({k: np.asarray(v, dtype=np.float64) for k, v in scal.items()})
    rec["seed"] = np.array(seed)
    return rec


def gen_sac_alpha():
    """G4: SoftActorCritic.train_step (sac_alpha.py:78-181), 5 chained steps, everything pinned."""
    kw = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, alpha_lr=3e-4,
              soft_target_tau=0.005, alpha=0.2, train_alpha=True, policy_mean_reg_weight=1e-3,
              policy_std_reg_weight=1e-3, beta_1=0.9)
    save("g4_sac_alpha_small", **_sac_case(404, 11, 3, [32, 32], 32, 5, kw, full=True))
    # GAIL-style hyper-parameters (gail_walker.yaml:12,74: reward_scale 2, beta_1 0.25), Walker dims
    kw2 = dict(kw, reward_scale=2.0, beta_1=0.25, target_entropy=-4.0)
    save("g4_sac_alpha_walker", **_sac_case(405, 17, 6, [32, 32], 16, 3, kw2, full=True))
    # Hopper BASELINE dims (H=256, B=256): losses per step + checksums / strided samples of the weights
    save("g4_sac_alpha_h256", **_sac_case(406, 11, 3, [256, 256], 256, 3, kw, full=False))


def gen_replay():
    """G10: ring semantics of SimpleReplayBuffer via EnvReplayBuffer-free scripted sequence."""
    from rlkit.data_management.simple_replay_buffer import SimpleReplayBuffer
    from oracle.replay import ReplayOracle
    rng = np.random.default_rng(1010)
    cap, o, a = 23, 4, 2
    rb = SimpleReplayBuffer(cap, o, a, random_seed=1995)
    orc = ReplayOracle(cap, o, a, random_seed=1995)
    # script: episode lengths and how each ends: 't' terminal flag on last sample, 'x' terminate_episode()
    script = [(5, "t"), (7, "x"), (3, "t"), (1, "t"), (9, "x"), (6, "t"), (4, "x"), (8, "t"), (2, "x")]
    N = sum(l for l, _ in script)
    obs = rng.normal(0, 1, (N, o)); nobs = rng.normal(0, 1, (N, o)); act = rng.normal(0, 1, (N, a))
    rew = rng.normal(0, 1, (N,)); term = np.zeros(N, np.uint8); ep_end = np.zeros(N, np.uint8)
    snaps, i = [], 0
    for L, how in script:
        for j in range(L):
            last = j == L - 1
            term[i] = 1 if (last and how == "t") else 0
            rb.add_sample(obs[i], act[i], rew[i], term[i], nobs[i])
            orc.add_sample(obs[i], act[i], rew[i], int(term[i]), nobs[i])
            i += 1
        ep_end[i - 1] = 1
        rb.terminate_episode(); orc.terminate_episode()
        ends = np.array(sorted(rb._traj_endpoints.items()), dtype=np.int64).reshape(-1, 2)
        assert rb._top == orc.top and rb._size == orc.size and dict(rb._traj_endpoints) == orc.traj_endpoints
        snaps.append((rb._top, rb._size, i, ends))
    idx = np.array([0, 5, 22, 7, 7, 13])
    gb = rb._get_batch_using_indices(idx)
    trajs = rb.sample_all_trajs()
    starts = np.array(list(rb._traj_endpoints.keys())); ends_ = np.array(list(rb._traj_endpoints.values()))
    out = dict(cap=cap, o=o, a=a, obs=obs, next_obs=nobs, act=act, rew=rew, term=term, ep_end=ep_end,
               snap_top=np.array([s[0] for s in snaps]), snap_size=np.array([s[1] for s in snaps]),
               snap_n=np.array([s[2] for s in snaps]),
               idx=idx, gather_obs=gb["observations"], gather_act=gb["actions"], gather_rew=gb["rewards"],
               gather_term=gb["terminals"], gather_next_obs=gb["next_observations"],
               final_traj_starts=starts, final_traj_ends=ends_,
               traj_lens=np.array([len(tj["rewards"]) for tj in trajs]),
               traj_rew_concat=np.concatenate([tj["rewards"].ravel() for tj in trajs]),
               randint_1995=np.random.RandomState(1995).randint(0, 1000, 8))
    for k, s in enumerate(snaps):
        out[f"snap{k}_ends"] = s[3]
    otr = orc.sample_all_trajs()
    assert [len(x["rewards"]) for x in otr] == list(out["traj_lens"])
    assert np.allclose(np.concatenate([x["rewards"].ravel() for x in otr]), out["traj_rew_concat"], atol=1e-6)
    save("g10_replay", **out)


def gen_replay_trajs():
    """G23: the trajectory-sampling surface of SimpleReplayBuffer (simple_replay_buffer.py:219-226, 334-395): sample_trajs with and without
    samples_per_traj (incl. more samples than a trajectory holds: choice with replacement), sample_all_trajs(samples_per_traj), get_all —
    on the scripted ring of G10 (wrapped, trajectories of unequal length), draws from the buffer's own RandomState(seed)."""
    from rlkit.data_management.simple_replay_buffer import SimpleReplayBuffer
    from oracle.replay import ReplayOracle
    g = np.load(os.path.join(GOLD, "g10_replay.npz"))
    cap, o, a = int(g["cap"]), int(g["o"]), int(g["a"])
    rb, orc = SimpleReplayBuffer(cap, o, a, random_seed=31), ReplayOracle(cap, o, a, random_seed=31)
    for i in range(len(g["rew"])):
        for b in (rb, orc):
            b.add_sample(g["obs"][i], g["act"][i], g["rew"][i], int(g["term"][i]), g["next_obs"][i])
            if g["ep_end"][i]:
                b.terminate_episode()
    calls = [("sample_trajs", dict(num_trajs=3)), ("sample_trajs", dict(num_trajs=2, samples_per_traj=4)),
             ("sample_trajs", dict(num_trajs=9, samples_per_traj=12)), ("sample_all_trajs", dict(samples_per_traj=3)), ("get_all", {})]
    out = dict(seed=np.array(31), n_calls=np.array(len(calls)))
    for ci, (fn, kw) in enumerate(calls):
        ref = getattr(rb, fn)(**kw)
        mine = getattr(orc, fn)(**kw)
        ref, mine = (ref if isinstance(ref, list) else [ref]), (mine if isinstance(mine, list) else [mine])
        assert len(ref) == len(mine)
        for x, y in zip(ref, mine):
            for k in ("observations", "actions", "rewards", "terminals", "next_observations"):
                assert np.allclose(np.asarray(x[k], np.float64), np.asarray(y[k], np.float64), atol=1e-6), (fn, kw, k)
        out[f"c{ci}_lens"] = np.array([len(x["rewards"]) for x in ref])
        out[f"c{ci}_obs"] = np.concatenate([np.asarray(x["observations"]) for x in ref])
        out[f"c{ci}_rew"] = np.concatenate([np.asarray(x["rewards"]).ravel() for x in ref])
    save("g23_replay_trajs", **out)


def gen_rms_actionmap():
    """G11 RunningMeanStd + normalize_obs (normalizer.py:128-152, vecenvs.py:299-327);
    G12 NormalizedBoxEnv action map (wrappers.py:342-346)."""
    from rlkit.data_management.normalizer import RunningMeanStd
    rng = np.random.default_rng(1111)
    rms = RunningMeanStd()
    xs = [rng.normal(2.0, 3.0, (nb, 5)) for nb in (4, 7, 1, 16)]
    means, vars_, counts = [], [], []
    for x in xs:
        rms.update(x)
        means.append(np.array(rms.mean)); vars_.append(np.array(rms.var)); counts.append(rms.count)
    eps = np.finfo(np.float32).eps.item()  # vecenvs.py:107
    probe = rng.normal(2.0, 30.0, (6, 5))
    normed = np.clip((probe - rms.mean) / np.sqrt(rms.var + eps), -10.0, 10.0)
    # action map: lb + (a+1)/2*(ub-lb), clip
    lb = np.array([-1.0, -2.0, 0.0]); ub = np.array([1.0, 2.0, 0.4])
    acts = rng.uniform(-1.5, 1.5, (10, 3))
    scaled = np.clip(lb + (acts + 1.0) * 0.5 * (ub - lb), lb, ub)
    save("g11_g12_rms_actionmap", x0=xs[0], x1=xs[1], x2=xs[2], x3=xs[3], means=np.array(means),
         vars=np.array(vars_), counts=np.array(counts), probe=probe, normed=normed, eps=eps,
         lb=lb, ub=ub, acts=acts, scaled=scaled)


def gen_disc():
    """G8: AdvIRL._do_reward_training (adv_irl.py:133-216) driven as an unbound function on a namespace that
    carries the attributes it touches; G9: the reward modes of _do_policy_training (adv_irl.py:277-298)."""
    import types
    import torch.nn as nn
    import torch.nn.functional as F
    import torch.optim as optim
    from rlkit.torch.algorithms.adv_irl.adv_irl import AdvIRL
    from rlkit.torch.algorithms.adv_irl.disc_models.simple_disc_models import MLPDisc
    from oracle.disc import DiscOracle, disc_reward, RELU, TANH
    out = {}
    for tag, act, D, Hd, B, steps, scale in (("tanh", TANH, 23, 128, 32, 3, 1.0), ("relu", RELU, 14, 64, 16, 2, 1.0),
                                               ("tanh_sat", TANH, 23, 128, 32, 2, 40.0)):
        rng = np.random.default_rng(808 + len(tag))
        flat = omlp.init_mlp(rng, D, [Hd, Hd], 1, init_w=0.3, b_init=0.05)
        if scale != 1.0:   # push some logits past +-10 so the clamp gate matters (simple_disc_models.py:45-47)
            lay = omlp.unpack(flat.copy(), D, [Hd, Hd], 1)
            lay[2] = (lay[2][0] * np.float32(scale), lay[2][1])
            flat = omlp.pack(lay)
        disc = MLPDisc(D, num_layer_blocks=2, hid_dim=Hd, hid_act=tag.split("_")[0], use_bn=False, clamp_magnitude=10.0)
        set_flat(disc, flat)
        # the saturated case keeps the gradient penalty on: a clamped interpolate has dD/dx = 0, counts (0-1)^2 in the
        # penalty and contributes no gradient (torch's norm backward is 0 at 0)
        use_gp = True
        kw = dict(disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=use_gp, grad_pen_weight=8.0)
        orc = DiscOracle(D, Hd, flat, act=act, **kw)
        ns = types.SimpleNamespace(
            discriminator=disc, disc_optimizer=optim.Adam(disc.parameters(), lr=kw["disc_lr"], betas=(kw["disc_momentum"], 0.999)),
            state_only=False, wrap_absorbing=False, disc_optim_batch_size=B, bce=nn.BCEWithLogitsLoss(),
            bce_targets=torch.cat([torch.ones(B, 1), torch.zeros(B, 1)], 0), use_grad_pen=use_gp,
            grad_pen_weight=kw["grad_pen_weight"], disc_eval_statistics=None)
        o_dim = D - 6 if D == 23 else D - 3
        out[f"{tag}_params0"] = flat
        out[f"{tag}_dims"] = np.array([D, Hd, B, steps, o_dim])
        for st in range(steps):
            xe = rng.normal(0, 1, (B, D)).astype(np.float32)
            xp = (rng.normal(0, 1, (B, D)) * 1.5 + 0.3).astype(np.float32)
            eps = rng.random((B, 1)).astype(np.float32)
            batches = {True: dict(observations=t(xe[:, :o_dim]), actions=t(xe[:, o_dim:])),
                       False: dict(observations=t(xp[:, :o_dim]), actions=t(xp[:, o_dim:]))}
            ns.get_batch = lambda bs, from_expert, keys=None: batches[from_expert]
            ns.disc_eval_statistics = None
            with H.NoiseInjector() as inj:
                if use_gp:
                    inj.push(eps)
                AdvIRL._do_reward_training(ns, 0)
            stt = ns.disc_eval_statistics
            res = orc.train_step(xe, xp, eps)
            assert np.allclose(stt["Disc CE Loss"], res["ce_loss"], rtol=1e-4, atol=1e-6), (tag, st, stt["Disc CE Loss"], res["ce_loss"])
            if use_gp:
                assert np.allclose(stt["Grad Pen"] * 8.0, res["grad_pen_loss"], rtol=2e-3, atol=1e-5), (tag, st, stt["Grad Pen"] * 8, res["grad_pen_loss"])
            assert np.allclose(stt["Disc Acc"], res["accuracy"])
            gref = get_flat_grad(disc)
            err = np.abs(gref - res["grad"]).max() / np.abs(gref).max()
            assert err < 5e-3, (tag, st, err)
            assert np.abs(get_flat(disc) - orc.p).max() < 5e-5, (tag, st)
            out.update({f"{tag}_s{st}_x_exp": xe, f"{tag}_s{st}_x_pol": xp, f"{tag}_s{st}_eps": eps,
                        f"{tag}_s{st}_ce": stt["Disc CE Loss"], f"{tag}_s{st}_gp": stt.get("Grad Pen", 0.0), f"{tag}_s{st}_acc": stt["Disc Acc"],
                        f"{tag}_s{st}_grad": gref, f"{tag}_s{st}_params": get_flat(disc)})
        probe = rng.normal(0, 2, (40, D)).astype(np.float32)
        out[f"{tag}_probe"] = probe
        out[f"{tag}_probe_logits"] = n(disc(t(probe)))
        assert np.allclose(orc.logits(probe), out[f"{tag}_probe_logits"], rtol=1e-4, atol=1e-5)
    # G9 reward modes on a logits grid (incl. the softplus threshold and the clamp range)
    grid = np.concatenate([np.linspace(-12, 12, 97), [-25.0, 25.0, 20.5, -20.5]]).astype(np.float32).reshape(-1, 1)
    tg_ = t(grid)
    refs = dict(airl=tg_, gail=F.softplus(tg_, beta=1), gail2=F.softplus(tg_, beta=-1), fairl=torch.exp(tg_) * (-1.0 * tg_))
    out["rew_grid"] = grid
    for mode, r in refs.items():
        out[f"rew_{mode}"] = n(r)
        assert np.allclose(disc_reward(grid, mode), n(r), rtol=1e-5, atol=1e-6), mode
    out["rew_gail2_clip"] = n(torch.clamp(torch.clamp(refs["gail2"], max=-0.5), min=-5.0))
    assert np.allclose(disc_reward(grid, "gail2", rew_clip_min=-5.0, rew_clip_max=-0.5), out["rew_gail2_clip"])
    save("g8_g9_disc", **out)


def gen_disc_blocks():
    """G25: AdvIRL._do_reward_training with MLPDisc(num_layer_blocks = 1 and 3, use_bn=False) (simple_disc_models.py:29-39), tanh and relu,
    gradient penalty on: the reference's autograd double backward against oracle.DiscOracle.train_step_blocks."""
    import types
    import torch.nn as nn
    import torch.optim as optim
    from rlkit.torch.algorithms.adv_irl.adv_irl import AdvIRL
    from rlkit.torch.algorithms.adv_irl.disc_models.simple_disc_models import MLPDisc
    from oracle.disc import DiscOracle, RELU, TANH
    out = {}
    for tag, act, L, D, Hd, B, steps in (("tanh1", TANH, 1, 23, 128, 32, 3), ("tanh3", TANH, 3, 23, 128, 32, 3), ("relu3", RELU, 3, 14, 64, 16, 2),
                                          ("relu1", RELU, 1, 14, 64, 16, 2)):
        rng = np.random.default_rng(2500 + L + 10 * act)
        flat = omlp.init_mlp(rng, D, [Hd] * L, 1, init_w=0.3, b_init=0.05)
        disc = MLPDisc(D, num_layer_blocks=L, hid_dim=Hd, hid_act="tanh" if act == TANH else "relu", use_bn=False, clamp_magnitude=10.0)
        set_flat(disc, flat)
        kw = dict(disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=True, grad_pen_weight=8.0)
        orc = DiscOracle(D, Hd, flat, act=act, num_layer_blocks=L, **kw)
        ns = types.SimpleNamespace(
            discriminator=disc, disc_optimizer=optim.Adam(disc.parameters(), lr=kw["disc_lr"], betas=(kw["disc_momentum"], 0.999)),
            state_only=False, wrap_absorbing=False, disc_optim_batch_size=B, bce=nn.BCEWithLogitsLoss(),
            bce_targets=torch.cat([torch.ones(B, 1), torch.zeros(B, 1)], 0), use_grad_pen=True,
            grad_pen_weight=kw["grad_pen_weight"], disc_eval_statistics=None)
        o_dim = D - 6 if D == 23 else D - 3
        out[f"{tag}_params0"] = flat
        out[f"{tag}_dims"] = np.array([D, Hd, B, steps, o_dim, L])
        for st in range(steps):
            xe = rng.normal(0, 1, (B, D)).astype(np.float32)
            xp = (rng.normal(0, 1, (B, D)) * 1.5 + 0.3).astype(np.float32)
            eps = rng.random((B, 1)).astype(np.float32)
            batches = {True: dict(observations=t(xe[:, :o_dim]), actions=t(xe[:, o_dim:])),
                       False: dict(observations=t(xp[:, :o_dim]), actions=t(xp[:, o_dim:]))}
            ns.get_batch = lambda bs, from_expert, keys=None: batches[from_expert]
            ns.disc_eval_statistics = None
            with H.NoiseInjector() as inj:
                inj.push(eps)
                AdvIRL._do_reward_training(ns, 0)
            stt = ns.disc_eval_statistics
            res = orc.train_step_blocks(xe, xp, eps)
            assert np.allclose(stt["Disc CE Loss"], res["ce_loss"], rtol=1e-4, atol=1e-6), (tag, st)
            assert np.allclose(stt["Grad Pen"] * 8.0, res["grad_pen_loss"], rtol=2e-3, atol=1e-5), (tag, st, stt["Grad Pen"] * 8, res["grad_pen_loss"])
            gref = get_flat_grad(disc)
            err = np.abs(gref - res["grad"]).max() / np.abs(gref).max()
            assert err < 1e-4, (tag, st, err)
            assert np.abs(get_flat(disc) - orc.p).max() < 5e-5, (tag, st)
            out.update({f"{tag}_s{st}_x_exp": xe, f"{tag}_s{st}_x_pol": xp, f"{tag}_s{st}_eps": eps,
                        f"{tag}_s{st}_ce": stt["Disc CE Loss"], f"{tag}_s{st}_gp": stt["Grad Pen"], f"{tag}_s{st}_acc": stt["Disc Acc"]})
            if st == 0:
                out[f"{tag}_s0_grad"] = gref          # the gradient of the first step, the parameters after the last (file size)
        out[f"{tag}_params_final"] = get_flat(disc)
        probe = rng.normal(0, 2, (40, D)).astype(np.float32)
        out[f"{tag}_probe"] = probe
        out[f"{tag}_probe_logits"] = n(disc(t(probe)))
        print(tag, "gradient error vs autograd < 1e-4, parameters within 5e-5 after", steps, "steps")
    save("g25_disc_blocks", **out)


def gen_disc_bn():
    """G26: AdvIRL._do_reward_training with MLPDisc(use_bn=True) — the constructor's DEFAULT (simple_disc_models.py:15,30-31,36-37): BatchNorm1d
    in train mode in both forwards of the step (cross-entropy over 2B rows, gradient penalty over the B interpolates, each with its own batch
    statistics), the penalty's double backward through the batch statistics, running-statistics updates; then the eval-mode forward that
    _do_policy_training relabels rewards with (adv_irl.py:268-274).  tanh / relu, 2 and 3 blocks, chained steps."""
    import types
    import torch.nn as nn
    import torch.optim as optim
    from rlkit.torch.algorithms.adv_irl.adv_irl import AdvIRL
    from rlkit.torch.algorithms.adv_irl.disc_models.simple_disc_models import MLPDisc
    from oracle.disc import DiscBNOracle, RELU, TANH
    out = {}
    for tag, act, L, D, Hd, B, steps in (("tanh2", TANH, 2, 23, 128, 32, 3), ("relu2", RELU, 2, 14, 64, 16, 3), ("tanh3", TANH, 3, 14, 64, 16, 2),
                                          ("relu1", RELU, 1, 23, 100, 32, 2)):
        rng = np.random.default_rng(2600 + L + 10 * act)
        flat = DiscBNOracle.init(rng, D, Hd, L)
        disc = MLPDisc(D, num_layer_blocks=L, hid_dim=Hd, hid_act="tanh" if act == TANH else "relu", use_bn=True, clamp_magnitude=10.0)
        set_flat(disc, flat)
        disc.train()
        kw = dict(disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=True, grad_pen_weight=8.0)
        orc = DiscBNOracle(D, Hd, flat, act=act, num_layer_blocks=L, **kw)
        ns = types.SimpleNamespace(
            discriminator=disc, disc_optimizer=optim.Adam(disc.parameters(), lr=kw["disc_lr"], betas=(kw["disc_momentum"], 0.999)),
            state_only=False, wrap_absorbing=False, disc_optim_batch_size=B, bce=nn.BCEWithLogitsLoss(),
            bce_targets=torch.cat([torch.ones(B, 1), torch.zeros(B, 1)], 0), use_grad_pen=True,
            grad_pen_weight=kw["grad_pen_weight"], disc_eval_statistics=None)
        o_dim = D - 6 if D == 23 else D - 3
        out[f"{tag}_params0"] = flat
        out[f"{tag}_dims"] = np.array([D, Hd, B, steps, o_dim, L])
        bns = [m for m in disc.modules() if isinstance(m, nn.BatchNorm1d)]
        for st in range(steps):
            xe = rng.normal(0, 1, (B, D)).astype(np.float32)
            xp = (rng.normal(0, 1, (B, D)) * 1.5 + 0.3).astype(np.float32)
            eps = rng.random((B, 1)).astype(np.float32)
            batches = {True: dict(observations=t(xe[:, :o_dim]), actions=t(xe[:, o_dim:])),
                       False: dict(observations=t(xp[:, :o_dim]), actions=t(xp[:, o_dim:]))}
            ns.get_batch = lambda bs, from_expert, keys=None: batches[from_expert]
            ns.disc_eval_statistics = None
            with H.NoiseInjector() as inj:
                inj.push(eps)
                AdvIRL._do_reward_training(ns, 0)
            stt = ns.disc_eval_statistics
            res = orc.train_step(xe, xp, eps)
            assert np.allclose(stt["Disc CE Loss"], res["ce_loss"], rtol=1e-4, atol=1e-6), (tag, st)
            assert np.allclose(stt["Grad Pen"] * 8.0, res["grad_pen_loss"], rtol=2e-3, atol=1e-5), (tag, st, stt["Grad Pen"] * 8, res["grad_pen_loss"])
            gref = get_flat_grad(disc)
            err = np.abs(gref - res["grad"]).max() / np.abs(gref).max()
            assert err < 1e-4, (tag, st, err)
            # the Linear biases under a BatchNorm have gradient exactly 0 (the batch mean is subtracted): what autograd / the oracle hold there is
            # rounding noise (~1e-9), which Adam normalises to a full +-lr step in a noise-determined direction.  They do not influence the
            # function; every OTHER parameter must agree to 5e-5, the dead biases only stay within steps * lr
            dead = orc.dead_bias_mask()
            d = np.abs(get_flat(disc) - orc.p)
            assert d[~dead].max() < 5e-5 and d[dead].max() <= (st + 1) * 2.02 * kw["disc_lr"], (tag, st, d[~dead].max(), d[dead].max())
            for l, m in enumerate(bns):
                assert np.abs(n(m.running_mean) - orc.rm[l]).max() < 1e-3 and np.abs(n(m.running_var) - orc.rv[l]).max() < 1e-5, (tag, st, l)
            out.update({f"{tag}_s{st}_x_exp": xe, f"{tag}_s{st}_x_pol": xp, f"{tag}_s{st}_eps": eps,
                        f"{tag}_s{st}_ce": stt["Disc CE Loss"], f"{tag}_s{st}_gp": stt["Grad Pen"], f"{tag}_s{st}_acc": stt["Disc Acc"]})
            if st == 0:
                out[f"{tag}_s0_grad"] = gref          # the gradient of the first step, the parameters after the last (file size)
        out[f"{tag}_params_final"] = get_flat(disc)
        out[f"{tag}_running_mean"] = np.stack([n(m.running_mean) for m in bns])
        out[f"{tag}_running_var"] = np.stack([n(m.running_var) for m in bns])
        probe = rng.normal(0, 2, (40, D)).astype(np.float32)
        disc.eval()                                   # adv_irl.py:268: the policy's rewards come from the eval-mode discriminator
        out[f"{tag}_probe"] = probe
        out[f"{tag}_probe_logits_eval"] = n(disc(t(probe)))
        disc.train()
        assert np.allclose(orc.logits(probe), out[f"{tag}_probe_logits_eval"], rtol=1e-3, atol=2e-3), (tag, np.abs(orc.logits(probe) - out[f"{tag}_probe_logits_eval"]).max())
        out[f"{tag}_dead_bias_mask"] = dead
        print(tag, "gradient error vs autograd < 1e-4, parameters within 5e-5, running statistics within 1e-5 after", steps, "steps")
    save("g26_disc_bn", **out)


def gen_disc_branches():
    """G24: the AdvIRL branches the YAMLs leave off — state_only=True (adv_irl.py:140-162: discriminator input = cat(obs, next_obs);
    :269 in the reward relabel) and policy_optim_batch_size_from_expert > 0 (:239-255: the policy batch = cat([rows from the policy
    buffer, rows from the expert buffer]) along dim 0, relabelled as a whole) — by running AdvIRL._do_reward_training /
    _do_policy_training as unbound functions on a namespace, as gen_disc does."""
    import types
    import torch.nn as nn
    import torch.optim as optim
    from rlkit.torch.algorithms.adv_irl.adv_irl import AdvIRL
    from rlkit.torch.algorithms.adv_irl.disc_models.simple_disc_models import MLPDisc
    from oracle.disc import DiscOracle, disc_reward, TANH
    rng = np.random.default_rng(2424)
    o, a, Hd, B, steps = 11, 3, 64, 16, 2
    D = 2 * o
    flat = omlp.init_mlp(rng, D, [Hd, Hd], 1, init_w=0.3, b_init=0.05)
    disc = MLPDisc(D, num_layer_blocks=2, hid_dim=Hd, hid_act="tanh", use_bn=False, clamp_magnitude=10.0)
    set_flat(disc, flat)
    kw = dict(disc_lr=1e-3, disc_momentum=0.0, use_grad_pen=True, grad_pen_weight=10.0)   # the defaults of adv_irl.py:46-50
    orc = DiscOracle(D, Hd, flat, act=TANH, **kw)
    ns = types.SimpleNamespace(
        discriminator=disc, disc_optimizer=optim.Adam(disc.parameters(), lr=kw["disc_lr"], betas=(kw["disc_momentum"], 0.999)),
        state_only=True, wrap_absorbing=False, disc_optim_batch_size=B, bce=nn.BCEWithLogitsLoss(),
        bce_targets=torch.cat([torch.ones(B, 1), torch.zeros(B, 1)], 0), use_grad_pen=True, grad_pen_weight=kw["grad_pen_weight"],
        disc_eval_statistics=None)
    out = dict(dims=np.array([o, a, Hd, B, steps]), params0=flat)

    def rows(n, shift):
        return dict(observations=rng.normal(shift, 1, (n, o)).astype(np.float32), actions=np.tanh(rng.normal(0, 1, (n, a))).astype(np.float32),
                    rewards=rng.normal(0, 1, (n, 1)).astype(np.float32), terminals=(rng.random((n, 1)) < 0.2).astype(np.float32),
                    next_observations=rng.normal(shift, 1, (n, o)).astype(np.float32))
    for st in range(steps):
        be, bp = rows(B, 0.4), rows(B, -0.3)
        eps = rng.random((B, 1)).astype(np.float32)
        seen = []
        batches = {True: {k: t(v) for k, v in be.items()}, False: {k: t(v) for k, v in bp.items()}}

        def get_batch(bs, from_expert, keys=None):
            seen.append(tuple(keys))
            return {k: batches[from_expert][k] for k in keys}
        ns.get_batch = get_batch
        ns.disc_eval_statistics = None
        with H.NoiseInjector() as inj:
            inj.push(eps)
            AdvIRL._do_reward_training(ns, 0)
        assert seen == [("observations", "next_observations")] * 2, seen     # actions are never asked for
        stt = ns.disc_eval_statistics
        xe, xp = np.concatenate([be["observations"], be["next_observations"]], 1), np.concatenate([bp["observations"], bp["next_observations"]], 1)
        res = orc.train_step(xe, xp, eps)
        assert np.allclose(stt["Disc CE Loss"], res["ce_loss"], rtol=1e-4, atol=1e-6)
        assert np.allclose(stt["Grad Pen"] * 10.0, res["grad_pen_loss"], rtol=2e-3, atol=1e-5)
        gref = get_flat_grad(disc)
        assert np.abs(gref - res["grad"]).max() < 5e-3 * np.abs(gref).max()
        assert np.abs(get_flat(disc) - orc.p).max() < 5e-5
        for tag, b in (("exp", be), ("pol", bp)):
            out.update({f"s{st}_{tag}_{k}": v for k, v in b.items()})
        out.update({f"s{st}_eps": eps, f"s{st}_ce": stt["Disc CE Loss"], f"s{st}_gp": stt["Grad Pen"], f"s{st}_acc": stt["Disc Acc"],
                    f"s{st}_grad": gref, f"s{st}_params": get_flat(disc)})
    # ---- _do_policy_training with 5 of 16 rows from the expert buffer, state_only relabel, gail2 + clips
    Bp_, nfe = 16, 5
    bpol, bexp = rows(Bp_ - nfe, -0.3), rows(nfe, 0.4)
    asked, got = [], {}

    def get_batch2(bs, from_expert, keys=None):
        asked.append((bs, from_expert))
        src = bexp if from_expert else bpol
        assert src["observations"].shape[0] == bs
        return {k: t(v) for k, v in src.items()}

    class Trainer:
        def train_step(self, batch):
            got.update({k: n(v) for k, v in batch.items()})
    ns2 = types.SimpleNamespace(discriminator=disc, state_only=True, wrap_absorbing=False, policy_optim_batch_size=Bp_,
                                policy_optim_batch_size_from_expert=nfe, get_batch=get_batch2, mode="gail2", clip_max_rews=True,
                                clip_min_rews=True, rew_clip_max=-0.05, rew_clip_min=-3.0, policy_trainer=Trainer(), disc_eval_statistics={})
    AdvIRL._do_policy_training(ns2, 0)
    assert asked == [(Bp_ - nfe, False), (nfe, True)], asked
    for k in ("observations", "actions", "terminals", "next_observations"):
        assert np.array_equal(got[k], np.concatenate([bpol[k], bexp[k]])), k      # policy rows first, expert rows last
    lg = orc.logits(np.concatenate([got["observations"], got["next_observations"]], 1))
    assert np.allclose(disc_reward(lg, "gail2", rew_clip_min=-3.0, rew_clip_max=-0.05), got["rewards"], rtol=1e-4, atol=1e-5)
    out.update({f"pt_pol_{k}": v for k, v in bpol.items()})
    out.update({f"pt_exp_{k}": v for k, v in bexp.items()})
    out.update(pt_rewards=got["rewards"], pt_dims=np.array([Bp_, nfe]), pt_clip=np.array([-3.0, -0.05], np.float32),
               pt_rew_stats=np.array([ns2.disc_eval_statistics[k] for k in ("Disc Rew Mean", "Disc Rew Std", "Disc Rew Max", "Disc Rew Min")]))
    save("g24_disc_branches", **out)


def gen_ppo():
    """G7: PPO.calc_adv + train_step (ppo.py:57-170) on scripted trajectories with injected permutations.
    g7_ppo: ordinary regime; g7b_ppo_clip: narrow policy (log_std ~ -3) so that clip_grad_norm_(20) bites."""
    _gen_ppo_case("g7_ppo", 707, -0.3, 2)
    _gen_ppo_case("g7b_ppo_clip", 708, -3.0, 1)
    # g7c: use_value_clip=True (ppo.py:137-143) with a small clip_eps and a larger value lr so that, over three epochs, v moves past
    # v_old +- clip_eps on part of the rows: both branches of the max and both sides of the clamp carry gradient
    _gen_ppo_case("g7c_ppo_vclip", 709, -0.3, 3, use_value_clip=True, clip_eps=0.1, value_lr=3e-3)
    # g7d: the policy class's DEFAULT head, conditioned_std=True (policies.py:368-374): log_std from a second head, clamped; the head's
    # bias is set so that some rows sit outside [LOG_SIG_MIN, LOG_SIG_MAX] = [-20, 2] (the clamp's gate is exercised)
    _gen_ppo_case("g7d_ppo_condstd", 710, -0.3, 2, conditioned_std=True)
    # g7e: hidden_sizes the kernels have no width for, unequal ([48, 24]: networks.py:23-60 takes any list) — policy and value net
    _gen_ppo_case("g7e_ppo_unequal", 711, -0.3, 2, hidden_sizes=[48, 24])


def _gen_ppo_case(name, seed, ls_mean, epochs, **extra):
    from rlkit.torch.algorithms.ppo.ppo import PPO
    from rlkit.torch.common.networks import FlattenMlp
    from rlkit.torch.common.policies import ReparamMultivariateGaussianPolicy
    from oracle.ppo import PPOOracle, gae_one_traj
    rng = np.random.default_rng(seed)
    o, a, Hh = 11, 3, list(extra.pop("hidden_sizes", [64, 64]))
    kw = dict(reward_scale=1.0, discount=0.99, clip_eps=0.2, policy_lr=3e-4, value_lr=3e-4, gae_tau=0.95,
              value_l2_reg=1e-3, mini_batch_size=16, update_epoch=epochs)
    kw.update(extra)
    cond = bool(extra.pop("conditioned_std", False))
    kw.pop("conditioned_std", None)
    vf = FlattenMlp(hidden_sizes=Hh, input_size=o, output_size=1, hidden_activation=torch.tanh)
    pol = ReparamMultivariateGaussianPolicy(hidden_sizes=Hh, obs_dim=o, action_dim=a, conditioned_std=cond,
                                            hidden_activation=torch.tanh)
    vf0 = omlp.init_mlp(rng, o, Hh, 1)
    if cond:
        pi0 = omlp.init_mlp(rng, o, Hh, a, init_w=1e-3, n_heads=2, last_scale=(0.1, 0.0))   # fc.. | last_fc (x0.1, b 0) | last_fc_log_std
        nls = Hh[-1] * a + a
        pi0[-nls:-a] *= 1000.0                                                           # the log-std head varies over the rows ...
        pi0[-a:] = np.array([-0.3, 2.0, -0.5], np.float32)                               # ... and one dimension straddles LOG_SIG_MAX = 2
        names = [k for k, _ in pol.named_parameters()]
        assert names[-4:] == ["last_fc.weight", "last_fc.bias", "last_fc_log_std.weight", "last_fc_log_std.bias"], names
        set_flat(pol, pi0)
    else:
        pim = omlp.init_mlp(rng, o, Hh, a, init_w=1e-3, last_scale=(0.1, 0.0))   # policies.py:378-379
        ls0 = rng.normal(ls_mean, 0.2, a).astype(np.float32)
        pi0 = np.concatenate([pim, ls0])                      # our layout: mean net | action_log_std
        set_flat(pol, np.concatenate([ls0, pim]))             # torch yields action_log_std first
        assert [k for k, _ in pol.named_parameters()][0] == "action_log_std"
    set_flat(vf, vf0)
    tr = PPO(policy=pol, vf=vf, **kw)
    orc = PPOOracle(o, a, Hh, pi0, vf0, conditioned_std=cond, **kw)
    lens = [2, 7, 50, 13]
    trajs = []
    for L in lens:
        trajs.append(dict(observations=rng.normal(0, 1, (L, o)).astype(np.float32),
                          actions=rng.normal(0, 0.8, (L, a)).astype(np.float32),
                          rewards=rng.normal(1.0, 1.0, (L, 1)).astype(np.float32)))
    N = sum(lens)
    perms = [rng.permutation(N) for _ in range(kw["update_epoch"])]
    if cond:   # centre dimension 1 of the log-std head on LOG_SIG_MAX over THESE observations, so that the clamp gates part of its rows
        allobs = np.concatenate([tj["observations"] for tj in trajs])
        lsr0 = omlp.forward(pi0, allobs, o, Hh, a, n_heads=2, act=omlp.TANH)[0][1]
        pi0[-a + 1] += np.float32(2.0 - np.median(lsr0[:, 1]))
        set_flat(pol, pi0)
        orc = PPOOracle(o, a, Hh, pi0, vf0, conditioned_std=True, **kw)
    ttrajs = [{k: t(v) for k, v in tj.items()} for tj in trajs]
    with torch.no_grad():
        ref_obs, ref_act, ref_ret, ref_adv, ref_val = tr.calc_adv(ttrajs)
        ref_lp = pol.get_log_prob(ref_obs, ref_act)
    obs_, act_, R_, A_, V_ = orc.calc_adv(trajs)
    n_out0 = 0
    if cond:   # the clamp's gate must be exercised by the INITIAL policy: some, not all, entries of the log-std head outside [-20, 2]
        orc.log_prob(obs_, act_)
        outm = (orc._lsr < -20.0) | (orc._lsr > 2.0)
        n_out0 = int(outm.sum())
        print(name, "log-std head entries outside the clamp at the start:", n_out0, "of", outm.size, "per dim", outm.sum(0))
        assert 0 < n_out0 < outm.size and 0 < outm.sum(0)[1] < outm.shape[0]
    assert np.allclose(R_, n(ref_ret), rtol=1e-5, atol=1e-5) and np.allclose(A_, n(ref_adv), rtol=2e-4, atol=2e-5)
    assert np.allclose(orc.log_prob(obs_, act_)[0], n(ref_lp), rtol=1e-5, atol=1e-5)
    q = list(perms)
    orig = torch.randperm
    torch.randperm = lambda nn_: torch.as_tensor(q.pop(0))
    try:
        tr.train_step(ttrajs)
    finally:
        torch.randperm = orig
    res = orc.train_step(trajs, perms)
    pi_ref_t = get_flat(pol)
    pi_ref = pi_ref_t if cond else np.concatenate([pi_ref_t[a:], pi_ref_t[:a]])
    assert np.abs(orc.vf - get_flat(vf)).max() < 5e-5, np.abs(orc.vf - get_flat(vf)).max()
    assert np.abs(orc.pi - pi_ref).max() < 5e-5, np.abs(orc.pi - pi_ref).max()
    out = dict(dims=np.array([o, a] + Hh), lens=np.array(lens), vf0=vf0, pi0=pi0, perms=np.array(perms),
               returns=n(ref_ret), advantages=n(ref_adv), values=n(ref_val), fixed_log_probs=n(ref_lp),
               vf_final=get_flat(vf), pi_final=pi_ref)
    for i, tj in enumerate(trajs):
        out.update({f"t{i}_{k}": v for k, v in tj.items()})
    out.update(epochs=np.array(epochs), pi_grad_norm_last=np.array(res["pi_grad_norm"]))
    if cond:
        out.update(kw_conditioned_std=np.array(True), n_outside_clamp=np.array(n_out0))
    if extra:
        out.update({"kw_" + k: np.array(v) for k, v in extra.items()})
        if extra.get("use_value_clip"):   # how many rows ended outside the clip window (the fixture must exercise both branches)
            v_end = orc.v(obs_)[0]
            outside = np.abs(v_end - V_) > np.float32(kw["clip_eps"])
            print(name, "rows with |v - v_old| > clip_eps after training:", int(outside.sum()), "of", outside.size)
            print("   max |v-v_old|", float(np.abs(v_end - V_).max()), "count", int(outside.sum()), outside.size)
            assert 0 < outside.sum()
            out["n_outside_clip"] = np.array(int(outside.sum()))
    print(name, "last policy grad norm", res["pi_grad_norm"])
    save(name, **out)


def _hook_grads(grads, opt, name, mod):
    orig = opt.step

    def step(*aa, **kk):
        grads[name] = get_flat_grad(mod)
        return orig(*aa, **kk)
    opt.step = step


def _rand_batch(rng, B, o, a):
    return dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32),
                actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                rewards=rng.normal(0, 1, (B, 1)).astype(np.float32),
                terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))


def gen_td3():
    """G6: TD3.train_step (td3.py:72-124), 4 chained steps (delayed policy/target update at steps 0 and 2), the
    target policy's clipped noise injected (policies.py:182-186).  G6b: the same with Mlp's default output activation
    (identity, networks.py:31) instead of the tanh the run script passes."""
    _td3_case("g6_td3", True)
    _td3_case("g6b_td3_identity", False)


def _td3_case(tag, tanh_out):
    from rlkit.torch.algorithms.td3.td3 import TD3
    from rlkit.torch.common.networks import FlattenMlp
    from rlkit.torch.common.policies import MlpGaussianNoisePolicy
    from oracle.td3 import TD3Oracle
    rng = np.random.default_rng(606)
    o, a, Hh, B, steps = 11, 3, [64, 64], 32, 4
    kw = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, policy_and_target_update_period=2,
              soft_target_tau=0.005)
    pi0 = omlp.init_mlp(rng, o, Hh, a, init_w=1e-3)
    pi0[-(Hh[-1] * a + a):] *= 300.0      # head weights up so that tanh bends (init_w=1e-3 keeps it linear)
    q10, q20 = omlp.init_mlp(rng, o + a, Hh, 1), omlp.init_mlp(rng, o + a, Hh, 1)
    pol = MlpGaussianNoisePolicy(hidden_sizes=Hh, obs_dim=o, action_dim=a, policy_noise=0.2, policy_noise_clip=0.5,
                                 **(dict(output_activation=torch.tanh) if tanh_out else {}))
    qf1 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1)
    qf2 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1)
    set_flat(pol, pi0), set_flat(qf1, q10), set_flat(qf2, q20)
    tr = TD3(policy=pol, qf1=qf1, qf2=qf2, **kw)
    orc = TD3Oracle(o, a, Hh, pi0, q10, q20, policy_noise=0.2, policy_noise_clip=0.5, output_activation="tanh" if tanh_out else "identity", **kw)
    grads = {}
    _hook_grads(grads, tr.qf1_optimizer, "q1", qf1), _hook_grads(grads, tr.qf2_optimizer, "q2", qf2)
    _hook_grads(grads, tr.policy_optimizer, "pi", pol)
    rec = dict(pi0=pi0, q10=q10, q20=q20, dims=np.array([o, a, B, steps] + Hh))
    for s in range(steps):
        batch = _rand_batch(rng, B, o, a)
        eps = rng.normal(0, 1, (B, a)).astype(np.float32)
        eps[0] = [4.0, -4.0, 0.1]                       # 0.2*4 = 0.8 > 0.5: the clip is active
        tr.eval_statistics = None
        grads.pop("pi", None)
        with H.NoiseInjector() as inj:
            inj.push(eps)
            tr.train_step({k: t(v) for k, v in batch.items()})
        st = tr.eval_statistics
        res = orc.train_step(batch, eps)
        for k_ref, k_or in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss")):
            assert np.allclose(st[k_ref], res[k_or], rtol=2e-4, atol=1e-6), (s, k_ref, st[k_ref], res[k_or])
        assert ("pi" in grads) == (s % 2 == 0) == ("pi_grad" in res)
        for nm in grads:
            err = np.abs(grads[nm] - res[nm + "_grad"]).max() / (np.abs(grads[nm]).max() + 1e-12)
            assert err < 2e-3, (s, nm, err)
        rec.update({f"s{s}_{k}": v for k, v in batch.items()})
        rec.update({f"s{s}_eps": eps, f"s{s}_qf1_loss": st["QF1 Loss"], f"s{s}_qf2_loss": st["QF2 Loss"],
                    f"s{s}_policy_loss": st["Policy Loss"], f"s{s}_grad_q1": grads["q1"], f"s{s}_grad_q2": grads["q2"],
                    f"s{s}_pi": get_flat(pol), f"s{s}_q1": get_flat(qf1), f"s{s}_q2": get_flat(qf2),
                    f"s{s}_tpi": get_flat(tr.target_policy), f"s{s}_tq1": get_flat(tr.target_qf1),
                    f"s{s}_tq2": get_flat(tr.target_qf2), f"s{s}_q_target_mean": st["Q Targets Mean"]})
        if "pi" in grads:
            rec[f"s{s}_grad_pi"] = grads["pi"]
        for k, mod in (("pi", pol), ("q1", qf1), ("q2", qf2), ("tpi", tr.target_policy), ("tq1", tr.target_qf1)):
            err = np.abs(getattr(orc, k) - get_flat(mod)).max()
            assert err < 5e-5, (s, k, err)
    save(tag, **rec)


def gen_sac_v():
    """G5: SoftActorCritic (V-function variant) train_step (sac/sac.py:70-179), 3 chained steps."""
    from rlkit.torch.algorithms.sac.sac import SoftActorCritic
    from rlkit.torch.common.networks import FlattenMlp
    from rlkit.torch.common.policies import ReparamTanhMultivariateGaussianPolicy
    from oracle.sac_v import SacVOracle
    rng = np.random.default_rng(505)
    o, a, Hh, B, steps = 11, 3, [64, 64], 32, 3
    kw = dict(reward_scale=1.0, discount=0.99, alpha=0.2, policy_lr=3e-4, qf_lr=3e-4, vf_lr=3e-4, soft_target_tau=0.005,
              policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9)
    pi0 = omlp.init_mlp(rng, o, Hh, a, init_w=1e-3, n_heads=2)
    q10, q20, vf0 = omlp.init_mlp(rng, o + a, Hh, 1), omlp.init_mlp(rng, o + a, Hh, 1), omlp.init_mlp(rng, o, Hh, 1)
    pol = ReparamTanhMultivariateGaussianPolicy(hidden_sizes=Hh, obs_dim=o, action_dim=a)
    qf1 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1)
    qf2 = FlattenMlp(hidden_sizes=Hh, input_size=o + a, output_size=1)
    vf = FlattenMlp(hidden_sizes=Hh, input_size=o, output_size=1)
    set_flat(pol, pi0), set_flat(qf1, q10), set_flat(qf2, q20), set_flat(vf, vf0)
    tr = SoftActorCritic(policy=pol, qf1=qf1, qf2=qf2, vf=vf, **kw)
    orc = SacVOracle(o, a, Hh, pi0, q10, q20, vf0, **kw)
    grads = {}
    _hook_grads(grads, tr.qf1_optimizer, "q1", qf1), _hook_grads(grads, tr.qf2_optimizer, "q2", qf2)
    _hook_grads(grads, tr.vf_optimizer, "vf", vf), _hook_grads(grads, tr.policy_optimizer, "pi", pol)
    rec = dict(pi0=pi0, q10=q10, q20=q20, vf0=vf0, dims=np.array([o, a, B, steps] + Hh))
    for s in range(steps):
        batch = _rand_batch(rng, B, o, a)
        eps = rng.normal(0, 1, (B, a)).astype(np.float32)
        tr.eval_statistics = None
        with H.NoiseInjector() as inj:
            inj.push(eps)
            tr.train_step({k: t(v) for k, v in batch.items()})
        st = tr.eval_statistics
        res = orc.train_step(batch, eps)
        for k_ref, k_or in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("VF Loss", "vf_loss"),
                            ("Policy Loss", "policy_loss")):
            assert np.allclose(st[k_ref], res[k_or], rtol=2e-4, atol=1e-6), (s, k_ref, st[k_ref], res[k_or])
        for nm in ("q1", "q2", "vf", "pi"):
            err = np.abs(grads[nm] - res[nm + "_grad"]).max() / (np.abs(grads[nm]).max() + 1e-12)
            assert err < 2e-3, (s, nm, err)
        rec.update({f"s{s}_{k}": v for k, v in batch.items()})
        rec.update({f"s{s}_eps": eps, f"s{s}_qf1_loss": st["QF1 Loss"], f"s{s}_qf2_loss": st["QF2 Loss"],
                    f"s{s}_vf_loss": st["VF Loss"], f"s{s}_policy_loss": st["Policy Loss"],
                    f"s{s}_grad_q1": grads["q1"], f"s{s}_grad_vf": grads["vf"], f"s{s}_grad_pi": grads["pi"],
                    f"s{s}_pi": get_flat(pol), f"s{s}_q1": get_flat(qf1), f"s{s}_q2": get_flat(qf2),
                    f"s{s}_vf": get_flat(vf), f"s{s}_tvf": get_flat(tr.target_vf)})
        for k, mod in (("pi", pol), ("q1", qf1), ("q2", qf2), ("vf", vf), ("tvf", tr.target_vf)):
            err = np.abs(getattr(orc, k) - get_flat(mod)).max()
            assert err < 5e-5, (s, k, err)
    save("g5_sac_v", **rec)


def gen_bc():
    """G13: BC._do_update_step (bc/bc.py:81-106) driven as an unbound function on a namespace carrying the attributes it
    touches; modes MLE and MSE (the MSE mode samples its action: torch.randn injected), 3 chained Adam steps each."""
    import types
    import torch.optim as optim
    from rlkit.torch.algorithms.bc.bc import BC
    from rlkit.torch.common.policies import ReparamTanhMultivariateGaussianPolicy
    from oracle.bc import BCOracle
    out = {}
    for mode, seed in (("MLE", 1301), ("MSE", 1302)):
        rng = np.random.default_rng(seed)
        o, a, Hh, B, steps = 11, 3, [64, 64], 32, 3
        pi0 = omlp.init_mlp(rng, o, Hh, a, init_w=1e-3, n_heads=2)
        pi0[-(2 * (Hh[-1] * a + a)):] *= 200.0            # heads up so that tanh / log_std matter
        pol = ReparamTanhMultivariateGaussianPolicy(hidden_sizes=Hh, obs_dim=o, action_dim=a)
        set_flat(pol, pi0)
        opt = optim.Adam(pol.parameters(), lr=1e-3, betas=(0.5, 0.999))
        grads = {}
        _hook_grads(grads, opt, "pi", pol)
        orc = BCOracle(o, a, Hh, pi0, mode=mode, lr=1e-3, momentum=0.5)
        out[f"{mode}_dims"] = np.array([o, a, B, steps] + Hh)
        out[f"{mode}_pi0"] = pi0
        for s in range(steps):
            obs = rng.normal(0, 1, (B, o)).astype(np.float32)
            acts = np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32)
            acts[0, 0] = 0.999999                          # near-saturated expert action: the atanh epsilon matters
            eps = rng.normal(0, 1, (B, a)).astype(np.float32)
            ns = types.SimpleNamespace(mode=mode, batch_size=B, exploration_policy=pol, optimizer=opt, eval_statistics=None,
                                       get_batch=lambda *aa, **kk: dict(observations=t(obs), actions=t(acts)))
            with H.NoiseInjector() as inj:
                if mode == "MSE":
                    inj.push(eps)
                BC._do_update_step(ns, 0, use_expert_buffer=True)
            stat = float(ns.eval_statistics["Log-Likelihood" if mode == "MLE" else "MSE"])
            res = orc.update(obs, acts, eps)
            assert np.allclose(stat, res["stat"], rtol=2e-4, atol=1e-5), (mode, s, stat, res["stat"])
            err = np.abs(grads["pi"] - res["grad"]).max() / (np.abs(grads["pi"]).max() + 1e-12)
            assert err < 2e-3, (mode, s, err)
            assert np.abs(orc.pi - get_flat(pol)).max() < 5e-5
            out.update({f"{mode}_s{s}_obs": obs, f"{mode}_s{s}_acts": acts, f"{mode}_s{s}_eps": eps, f"{mode}_s{s}_stat": stat,
                        f"{mode}_s{s}_grad": grads["pi"], f"{mode}_s{s}_pi": get_flat(pol)})
    save("g13_bc", **out)


def gen_terminals():
    """G14: the batched terminal predicates (rlkit/envs/terminals.py:6-117) on inputs that straddle every threshold and
    carry NaN / inf / large-negative entries."""
    # the module is loaded by path: importing the package rlkit.envs pulls in envpool / gym / dmc2gym, none installed here;
    # terminals.py itself only needs numpy.  get_terminal_func's name rule (:6-11) is applied by hand.
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_terminals", os.path.join(H.REF, "rlkit", "envs", "terminals.py"))
    ref_terminals = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_terminals)

    def get_terminal_func(env_name):
        cls_name = "".join(s_[0].upper() + s_[1:] for s_ in env_name.split("_")) + "TerminalFunc"
        return getattr(ref_terminals, cls_name).is_terminal

    from oracle import terminals as oterm
    rng = np.random.default_rng(1414)
    out = {}
    for kind, name, o in (("inverted_pendulum", "inverted_pendulum", 4), ("inverted_double_pendulum", "inverted_double_pendulum", 11),
                          ("hopper", "hopper", 11), ("walker2d", "walker2d", 17), ("halfcheetah", "halfcheetah", 17),
                          ("humanoid", "humanoid", 45), ("ant", "ant", 27)):
        nrow = 96
        x = rng.normal(0.0, 1.0, (nrow, o)).astype(np.float32)
        x[:, 0] = rng.uniform(0.0, 2.4, nrow)
        x[:, 1] = rng.uniform(-1.3, 1.3, nrow) * rng.choice([0.1, 0.25, 1.0], nrow)
        x[5, 3 % o] = np.nan; x[9, o - 1] = np.inf; x[11, 2 % o] = 150.0; x[13, 2 % o] = -150.0; x[17, 0] = np.nan
        x[19, 0] = 0.7; x[21, 0] = 0.8; x[23, 0] = 2.0; x[25, 0] = 1.0; x[27, 0] = 0.2; x[29, 1] = 0.2; x[31, 1] = -1.0
        f = get_terminal_func(name)
        act = np.zeros((nrow, 1), np.float32)
        done = np.asarray(f(x, act, x))
        assert done.shape == (nrow, 1) and done.dtype == bool
        assert np.array_equal(done, oterm.is_terminal(kind, x)), kind
        out[kind + "_x"], out[kind + "_done"] = x, done
    save("g14_terminals", **out)


def gen_eval_stats():
    """G15: get_generic_path_information / get_average_returns / create_stats_ordered_dict (core/eval_util.py:15-142) on scripted
    paths of unequal length, with and without the `is_success` env-info branch."""
    from rlkit.core.eval_util import create_stats_ordered_dict, get_average_returns, get_generic_path_information
    rng = np.random.default_rng(1515)
    out = {}
    for tag, with_success in (("plain", False), ("success", True)):
        paths, lens = [], (7, 1, 12, 4)
        for pi_, T in enumerate(lens):
            infos = [dict(env_id=pi_, **({"is_success": float((pi_ % 2 == 0) and t == T - 1)} if with_success else {})) for t in range(T)]
            paths.append(dict(observations=rng.normal(0, 1, (T, 5)), actions=rng.uniform(-1, 1, (T, 3)),
                              rewards=rng.normal(0.5, 1.0, (T, 1)), next_observations=rng.normal(0, 1, (T, 5)),
                              terminals=np.zeros((T, 1)), env_infos=infos))
        st = get_generic_path_information(paths, stat_prefix="Test")
        keys = list(st.keys())
        out[tag + "_keys"] = np.array(keys)
        out[tag + "_vals"] = np.array([float(np.asarray(st[k]).reshape(-1)[0]) for k in keys])
        out[tag + "_avg_return"] = float(get_average_returns(paths))
        mean_std = get_average_returns(paths, std=True)
        out[tag + "_avg_return_std"] = np.array([float(mean_std[0]), float(mean_std[1])])
        out[tag + "_lens"] = np.array(lens)
        for i, p_ in enumerate(paths):
            out[f"{tag}_rew{i}"], out[f"{tag}_act{i}"] = p_["rewards"], p_["actions"]
    # create_stats_ordered_dict corner cases: Number, size-1 array, tuple, exclude_max_min
    c = OrderedDictFlat()
    c.add(create_stats_ordered_dict("A", 3.5))
    c.add(create_stats_ordered_dict("B", np.array([2.0])))
    c.add(create_stats_ordered_dict("C", (np.array([1.0, 2.0, 4.0]), np.array([5.0]))))
    c.add(create_stats_ordered_dict("D", np.array([1.0, 3.0]), stat_prefix="P", exclude_max_min=True))
    c.add(create_stats_ordered_dict("E", []))
    out["corner_keys"], out["corner_vals"] = np.array(c.keys), np.array(c.vals)
    save("g15_eval_stats", **out)


class OrderedDictFlat:
    def __init__(self):
        self.keys, self.vals = [], []

    def add(self, d):
        for k, v in d.items():
            self.keys.append(k); self.vals.append(float(np.asarray(v).reshape(-1)[0]))


def gen_variants():
    """G16: exp_spec -> variant list (launcher_util.build_nested_variant_generator, the grid run_experiment.py:25-45 walks):
    order and content of the yielded dicts, as JSON."""
    import copy
    import json
    from rlkit.launchers.launcher_util import build_nested_variant_generator
    specs = [
        dict(meta_data=dict(script_path="run_scripts/x.py", exp_name="x", num_workers=2),
             variables=dict(seed=[0, 1, 2], sac_params=dict(reward_scale=[2.0, 4.0]), adv_irl_params=dict(grad_pen_weight=[8.0])),
             constants=dict(net_size=256, sac_params=dict(discount=0.99), adv_irl_params=dict(mode="gail2"))),
        dict(meta_data=dict(script_path="s.py", exp_name="nested", num_workers=1),     # shape of the reference's own self-test (:467-500)
             variables=dict(hi=dict(one=[1, 2, 3, 4], two=[5678], three=dict(apple=["yummy", "sour", "sweet"])), bye=["omg", "lmfao", "waddup"]),
             constants=dict(hi=dict(three=dict(constant_banana="potassium"), other_constant_stuff=dict(idk="something funny and cool")),
                            yoyoyo="I like candy", wow=1e-4)),
        dict(meta_data=dict(script_path="s.py", exp_name="single", num_workers=1), variables=None,
             constants=dict(a=1, b=dict(c=2))),
        dict(meta_data=dict(script_path="s.py", exp_name="zeta_first", num_workers=1),   # key order vs alphabetical order
             variables=dict(zeta=[1, 2], alpha=dict(mid=[10, 20], aaa=[0.1, 0.2])), constants=dict(alpha=dict(k=0))),
    ]
    out = {}
    for i, spec in enumerate(specs):
        vs = [copy.deepcopy(v) for v in build_nested_variant_generator(copy.deepcopy(spec))()]
        out[f"spec{i}"] = json.dumps(spec)
        out[f"variants{i}"] = json.dumps(vs)
        print(i, len(vs))
    save("g16_variants", **out)


def gen_logger_csv():
    """G17: progress.csv as rlkit/core/logger.py writes it (record_tabular stringifies, header = keys of the first dump in
    insertion order, :226-227,300-318) for a scripted two-epoch sequence with python / numpy scalars and a string cell."""
    import contextlib
    import io
    import tempfile
    from rlkit.core import logger
    logger.set_log_tboard(False)
    logger._log_wandb = False
    rows = [[("Epoch", 0), ("AverageReturn", np.float64(12.5)), ("QF1 Loss", np.float32(0.25)), ("Alpha", 0.2),
             ("Number of env steps total", 4096), ("Note", "a,b")],
            [("Epoch", 1), ("AverageReturn", np.float64(1234.56789012345)), ("QF1 Loss", np.float32(1e-7)), ("Alpha", 0.19999),
             ("Number of env steps total", 8192), ("Note", "x")]]
    path = os.path.join(tempfile.mkdtemp(), "progress.csv")
    logger.add_tabular_output(path)
    with contextlib.redirect_stdout(io.StringIO()):
        for r in rows:
            for k, v in r:
                logger.record_tabular(k, v)
            logger.dump_tabular(with_prefix=False, with_timestamp=False)
    logger.remove_tabular_output(path)
    text = open(path, newline="").read()
    print(repr(text))
    save("g17_logger_csv", csv_text=np.array(text))


def gen_logdir():
    """G18: log-directory naming (launcher_util.create_exp_name / create_log_dir :176-206) at a frozen clock and the
    variant.json text logger.log_variant writes (:379-382) for a nested variant."""
    import datetime
    import tempfile
    from rlkit.core import logger
    from rlkit.launchers import launcher_util as LU

    class Frozen(datetime.datetime):
        @classmethod
        def now(cls, tz=None):
            return cls(2024, 3, 9, 7, 5, 1, tzinfo=tz)

    real = LU.datetime.datetime
    LU.datetime.datetime = Frozen
    try:
        base = tempfile.mkdtemp()
        d = LU.create_log_dir("sac_hopper_hip", exp_id=3, seed=17, base_log_dir=base)
    finally:
        LU.datetime.datetime = real
    rel = os.path.relpath(d, base)
    variant = dict(seed=17, exp_id=3, exp_name="sac_hopper_hip", net_size=256, sac_params=dict(reward_scale=1.0, alpha=0.2, policy_lr=3e-4),
                   env_specs=dict(env_name="hopper", env_kwargs={}, env_num=4096), flags=[True, None, 1e-7], script_path="run_scripts/x.py")
    path = os.path.join(d, "variant.json")
    logger.log_variant(path, variant)
    text = open(path).read()
    print(rel); print(text[:200])
    save("g18_logdir", rel_dir=np.array(rel), variant_json=np.array(text))


def gen_absorbing():
    """G19: SimpleReplayBuffer.add_path(path, absorbing=True, env) (simple_replay_buffer.py:134-216) on two scripted paths (one ends
    with a terminal transition, one does not) into a ring small enough to wrap; env.action_space.sample() is a scripted stream."""
    from rlkit.data_management.simple_replay_buffer import SimpleReplayBuffer
    from oracle.replay import ReplayOracle
    rng = np.random.default_rng(1919)
    cap, o, a = 16, 3, 2
    acts_stream = rng.uniform(-1, 1, (8, a))

    class Space:
        def __init__(self):
            self.i = 0

        def sample(self):
            self.i += 1
            return acts_stream[self.i - 1]
    env = type("E", (), {})()
    env.action_space = Space()
    paths = []
    for L, terminal_last in ((5, True), (4, False), (6, True)):
        term = np.zeros((L, 1), bool)
        term[-1, 0] = terminal_last
        paths.append(dict(observations=rng.normal(0, 1, (L, o)), actions=rng.uniform(-1, 1, (L, a)), rewards=rng.normal(0, 1, (L, 1)),
                          next_observations=rng.normal(0, 1, (L, o)), terminals=term))
    rb = SimpleReplayBuffer(cap, o, a, random_seed=7)
    orc = ReplayOracle(cap, o, a, random_seed=7)
    it = iter(acts_stream)
    for pth in paths:
        rb.add_path(pth, absorbing=True, env=env)
        orc.add_path(pth, absorbing=True, sample_action=lambda: next(it))
    idx = np.arange(cap)
    gb = rb._get_batch_using_indices(idx)
    ob = orc.gather(idx)
    for k in ("observations", "actions", "rewards", "terminals", "next_observations", "absorbing"):
        assert np.allclose(np.asarray(gb[k], np.float64), np.asarray(ob[k], np.float64), atol=1e-6), k
    assert rb._top == orc.top and rb._size == orc.size and dict(rb._traj_endpoints) == orc.traj_endpoints
    out = dict(cap=cap, o=o, a=a, acts_stream=acts_stream, n_paths=len(paths), top=rb._top, size=rb._size,
               traj_starts=np.array(list(rb._traj_endpoints.keys())), traj_ends=np.array(list(rb._traj_endpoints.values())),
               ring_obs=gb["observations"], ring_act=gb["actions"], ring_rew=gb["rewards"], ring_term=gb["terminals"],
               ring_next_obs=gb["next_observations"], ring_absorbing=gb["absorbing"])
    for i, pth in enumerate(paths):
        for k, v in pth.items():
            out[f"p{i}_{k}"] = v
    save("g19_absorbing", **out)


def gen_her():
    """G20-G22: the goal-conditioned trainers and the relabelling buffer of Hindsight Experience Replay.
    g20: her/td3.py TD3.train_step, 4 chained steps (target action = clamped noise as that file computes it, target clip active on
         some rows and not on others, policy loss with the action L2 term, delayed update at steps 0 and 2);
    g21: her/sac.py SAC.train_step, 3 chained steps (sac_alpha with concatenated inputs, target entropy -|A|);
    g22: relabel_replay_buffer.py HindsightReplayBuffer.random_batch, `future` and `final`, her_ratio 0.8 and 0, on scripted paths."""
    from rlkit.torch.algorithms.her.td3 import TD3
    from rlkit.torch.algorithms.her.sac import SAC
    from rlkit.torch.common.networks import FlattenMlp
    from rlkit.torch.common.policies import MlpGaussianAndEpsilonPolicy, ReparamTanhMultivariateGaussianPolicy
    from oracle.td3 import TD3Oracle
    from oracle.sac_alpha import SacAlphaOracle
    # ---------------------------------------------------------------- g20
    rng = np.random.default_rng(2020)
    o, gd, a, Hh, B, steps = 10, 3, 4, [64, 64], 32, 4
    kw = dict(reward_scale=1.0, discount=0.9, policy_lr=3e-4, qf_lr=3e-4, policy_and_target_update_period=2, soft_target_tau=0.005)
    pi0 = omlp.init_mlp(rng, o + gd, Hh, a, init_w=1e-3)
    pi0[-(Hh[-1] * a + a):] *= 300.0
    q10, q20 = omlp.init_mlp(rng, o + gd + a, Hh, 1), omlp.init_mlp(rng, o + gd + a, Hh, 1)
    for q in (q10, q20):                   # shift the critics so that min(TQ) straddles the clip window [-10, 0]
        q[-1] = -5.0
        q[-(Hh[-1] + 1):-1] *= 3000.0
    space = type("S", (), {"shape": (a,), "sample": lambda self: np.zeros(a)})()
    pol = MlpGaussianAndEpsilonPolicy(hidden_sizes=Hh, obs_dim=o + gd, action_dim=a, action_space=space, output_activation=torch.tanh,
                                      max_sigma=0.3, min_sigma=0.3)
    qf1 = FlattenMlp(hidden_sizes=Hh, input_size=o + gd + a, output_size=1)
    qf2 = FlattenMlp(hidden_sizes=Hh, input_size=o + gd + a, output_size=1)
    set_flat(pol, pi0), set_flat(qf1, q10), set_flat(qf2, q20)
    tr = TD3(policy=pol, qf1=qf1, qf2=qf2, **kw)
    orc = TD3Oracle(o + gd, a, Hh, pi0, q10, q20, policy_noise=0.3, policy_noise_clip=0.0, her=True, **kw)
    grads = {}
    _hook_grads(grads, tr.qf1_optimizer, "q1", qf1), _hook_grads(grads, tr.qf2_optimizer, "q2", qf2)
    _hook_grads(grads, tr.policy_optimizer, "pi", pol)
    rec = dict(pi0=pi0, q10=q10, q20=q20, dims=np.array([o, gd, a, B, steps] + Hh), sigma=np.float32(0.3),
               clip=np.array([tr.clip_return_l, tr.clip_return_r], np.float32))
    n_clipped = 0
    for s in range(steps):
        b = _rand_batch(rng, B, o, a)
        b["desired_goals"] = rng.normal(0, 1, (B, gd)).astype(np.float32)
        b["next_desired_goals"] = b["desired_goals"].copy()
        eps = rng.normal(0, 1, (B, a)).astype(np.float32)
        eps[0] = [4.0, -4.0, 0.1, 3.9]                 # 0.3 * 4 = 1.2 > max_act: the clamp is active
        tr.eval_statistics = None
        grads.pop("pi", None)
        with H.NoiseInjector() as inj:
            inj.push(eps)
            tr.train_step({k: t(v) for k, v in b.items()})
        st = tr.eval_statistics
        cat = dict(b, observations=np.concatenate([b["observations"], b["desired_goals"]], 1),
                   next_observations=np.concatenate([b["next_observations"], b["next_desired_goals"]], 1))
        res = orc.train_step(cat, eps)
        raw_tq = (res["q_target"] - b["rewards"]) / np.maximum(0.9 * (1 - b["terminals"]), 1e-9)
        n_clipped += int(((np.abs(raw_tq - tr.clip_return_l) < 1e-6) | (np.abs(raw_tq - tr.clip_return_r) < 1e-6)).sum())
        for k_ref, k_or in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss")):
            assert np.allclose(st[k_ref], res[k_or], rtol=2e-4, atol=1e-5), (s, k_ref, st[k_ref], res[k_or])
        assert ("pi" in grads) == (s % 2 == 0) == ("pi_grad" in res)
        for nm in grads:
            err = np.abs(grads[nm] - res[nm + "_grad"]).max() / (np.abs(grads[nm]).max() + 1e-12)
            assert err < 2e-3, (s, nm, err)
        rec.update({f"s{s}_{k}": v for k, v in b.items()})
        rec.update({f"s{s}_eps": eps, f"s{s}_qf1_loss": st["QF1 Loss"], f"s{s}_qf2_loss": st["QF2 Loss"],
                    f"s{s}_policy_loss": st["Policy Loss"], f"s{s}_grad_q1": grads["q1"], f"s{s}_grad_q2": grads["q2"],
                    f"s{s}_pi": get_flat(pol), f"s{s}_q1": get_flat(qf1), f"s{s}_q2": get_flat(qf2),
                    f"s{s}_tpi": get_flat(tr.target_policy), f"s{s}_tq1": get_flat(tr.target_qf1), f"s{s}_q_target_mean": st["Q Targets Mean"]})
        if "pi" in grads:
            rec[f"s{s}_grad_pi"] = grads["pi"]
        for k, mod in (("pi", pol), ("q1", qf1), ("q2", qf2), ("tpi", tr.target_policy), ("tq1", tr.target_qf1)):
            assert np.abs(getattr(orc, k) - get_flat(mod)).max() < 5e-5, (s, k)
    assert 10 < n_clipped < steps * B - 10, n_clipped          # the clip bites on some rows, not on all
    save("g20_her_td3", **rec)
    # ---------------------------------------------------------------- g21
    rng = np.random.default_rng(2121)
    o, gd, a, Hh, B, steps = 10, 3, 4, [64, 64], 16, 3
    kws = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, alpha_lr=3e-4, soft_target_tau=0.005, alpha=0.2,
               train_alpha=True, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9)
    pi0 = omlp.init_mlp(rng, o + gd, Hh, a, init_w=1e-3, n_heads=2)
    q10, q20 = omlp.init_mlp(rng, o + gd + a, Hh, 1), omlp.init_mlp(rng, o + gd + a, Hh, 1)
    pol = ReparamTanhMultivariateGaussianPolicy(hidden_sizes=Hh, obs_dim=o + gd, action_dim=a)
    qf1 = FlattenMlp(hidden_sizes=Hh, input_size=o + gd + a, output_size=1)
    qf2 = FlattenMlp(hidden_sizes=Hh, input_size=o + gd + a, output_size=1)
    set_flat(pol, pi0), set_flat(qf1, q10), set_flat(qf2, q20)
    tr = SAC(policy=pol, qf1=qf1, qf2=qf2, env=_Env(a), **kws)
    assert tr.target_entropy == -a
    orc = SacAlphaOracle(o + gd, a, Hh, pi0, q10, q20, target_entropy=-float(a), **kws)
    rec = dict(pi0=pi0, q10=q10, q20=q20, dims=np.array([o, gd, a, B, steps] + Hh))
    for s in range(steps):
        b = _rand_batch(rng, B, o, a)
        b["desired_goals"] = rng.normal(0, 1, (B, gd)).astype(np.float32)
        b["next_desired_goals"] = b["desired_goals"].copy()
        e1, e2 = rng.normal(0, 1, (B, a)).astype(np.float32), rng.normal(0, 1, (B, a)).astype(np.float32)
        tr.eval_statistics = None
        with H.NoiseInjector() as inj:
            inj.push(e1), inj.push(e2)
            tr.train_step({k: t(v) for k, v in b.items()})
        st = tr.eval_statistics
        cat = dict(b, observations=np.concatenate([b["observations"], b["desired_goals"]], 1),
                   next_observations=np.concatenate([b["next_observations"], b["next_desired_goals"]], 1))
        res = orc.train_step(cat, e1, e2)
        for k_ref, k_or in (("QF1 Loss", "qf1_loss"), ("Policy Loss", "policy_loss"), ("Alpha Loss", "alpha_loss")):
            assert np.allclose(st[k_ref], res[k_or], rtol=2e-4, atol=1e-6), (s, k_ref, st[k_ref], res[k_or])
        rec.update({f"s{s}_{k}": v for k, v in b.items()})
        rec.update({f"s{s}_eps_next": e1, f"s{s}_eps_cur": e2, f"s{s}_qf1_loss": st["QF1 Loss"], f"s{s}_policy_loss": st["Policy Loss"],
                    f"s{s}_alpha_loss": st["Alpha Loss"], f"s{s}_log_alpha": float(tr.log_alpha.detach()),
                    f"s{s}_pi": get_flat(pol), f"s{s}_q1": get_flat(qf1), f"s{s}_tq1": get_flat(tr.target_qf1)})
        for k, mod in (("pi", pol), ("q1", qf1), ("q2", qf2), ("tq1", tr.target_qf1)):
            assert np.abs(getattr(orc, k) - get_flat(mod)).max() < 5e-5, (s, k)
    save("g21_her_sac", **rec)
    # ---------------------------------------------------------------- g22
    import types
    if "rlkit.envs" not in sys.modules:        # rlkit/envs/__init__.py imports envpool / gym envs that are absent here; the buffer needs
        pkg = types.ModuleType("rlkit.envs")   # only goal_env_utils' two default callables, which the env below overrides
        pkg.__path__ = []
        sys.modules["rlkit.envs"] = pkg
        geu = types.ModuleType("rlkit.envs.goal_env_utils")
        geu.compute_reward = geu.compute_distance = None
        sys.modules["rlkit.envs.goal_env_utils"] = geu
    from rlkit.data_management.relabel_replay_buffer import HindsightReplayBuffer
    Box = sys.modules["gym.spaces"].Box
    o, gd, a, cap = 5, 2, 3, 40

    class DSpace(sys.modules["gym.spaces"].Dict):
        def __init__(self):
            self.spaces = dict(observation=Box(-np.ones(o), np.ones(o)), desired_goal=Box(-np.ones(gd), np.ones(gd)),
                               achieved_goal=Box(-np.ones(gd), np.ones(gd)))

    def compute_reward(ag, dg, info=None):
        return -(np.linalg.norm(ag - dg, axis=-1) > 0.5).astype(np.float32)
    env = type("E", (), dict(observation_space=DSpace(), action_space=Box(-np.ones(a), np.ones(a)), compute_reward=staticmethod(compute_reward)))()
    rng = np.random.default_rng(2222)
    rec = dict(dims=np.array([o, gd, a, cap]))
    paths = []
    for L in (7, 5, 9, 6, 8, 10):          # 45 samples into a 40-slot ring: the first trajectory is overwritten
        obs = [dict(observation=rng.normal(0, 1, o), desired_goal=rng.normal(0, 1, gd), achieved_goal=rng.normal(0, 1, gd)) for _ in range(L + 1)]
        for ob in obs[1:]:
            ob["desired_goal"] = obs[0]["desired_goal"]
        paths.append(dict(obs=obs, act=rng.uniform(-1, 1, (L, a)), rew=rng.normal(0, 1, L), term=[False] * (L - 1) + [L % 2 == 0]))
    for ci, (rtype, ratio) in enumerate((("future", 0.8), ("final", 0.8), ("future", 0.0))):
        ref = HindsightReplayBuffer(cap, env, random_seed=77, relabel_type=rtype, her_ratio=ratio)
        for pth in paths:
            for i in range(len(pth["act"])):
                ref.add_sample(pth["obs"][i], pth["act"][i], pth["rew"][i], pth["term"][i], pth["obs"][i + 1])
            ref.terminate_episode()
        np.random.seed(500 + ci)           # `future` draws its index from the GLOBAL numpy stream (relabel_replay_buffer.py:88)
        bt = ref.random_batch(12)
        rec.update({f"c{ci}_{k}": np.asarray(v) for k, v in bt.items()})
        rec[f"c{ci}_endpoints"] = np.array(sorted(ref._traj_endpoints.items()))
    for pi_, pth in enumerate(paths):
        rec[f"p{pi_}_obs"] = np.array([x["observation"] for x in pth["obs"]])
        rec[f"p{pi_}_dg"] = np.array([x["desired_goal"] for x in pth["obs"]])
        rec[f"p{pi_}_ag"] = np.array([x["achieved_goal"] for x in pth["obs"]])
        rec[f"p{pi_}_act"], rec[f"p{pi_}_rew"], rec[f"p{pi_}_term"] = pth["act"], pth["rew"], np.array(pth["term"])
    save("g22_her_buffer", **rec)


# generation order: a group that reads another group's file comes after it (replay_trajs loads g10_replay.npz, written by replay)
GROUPS = dict(mlp=gen_mlp, mlp_unequal=gen_mlp_unequal, head=gen_head, sac_alpha=gen_sac_alpha, sac_v=gen_sac_v, td3=gen_td3, ppo=gen_ppo,
              disc=gen_disc, disc_bn=gen_disc_bn, disc_blocks=gen_disc_blocks, disc_branches=gen_disc_branches, replay=gen_replay,
              replay_trajs=gen_replay_trajs, her=gen_her, absorbing=gen_absorbing, bc=gen_bc, rms=gen_rms_actionmap, terminals=gen_terminals,
              eval_stats=gen_eval_stats, variants=gen_variants, logger_csv=gen_logger_csv, logdir=gen_logdir)

if __name__ == "__main__":
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()
