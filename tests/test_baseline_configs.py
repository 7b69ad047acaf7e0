"""BASELINE.json configs 3, 4, 5 at FULL size on the GPU: size-independent properties plus oracle checks on what
the fused loops actually consumed.

  C4  PPO Hopper, 8192 envs x 128-step rollout: one whole iteration (rollout -> calc_adv -> 10 epochs of minibatches) is finite, the
      segment table is the episode-end table, and the GAE of sampled segments is oracle.ppo.gae_one_traj (ppo.py:57-100).
  C5  SAC Humanoid widths, 4 co-resident seeds x 1024 envs, grouped lock-step: every agent == oracle.sac_alpha on the rows / noise the
      fused step drew (sac_alpha.py:78-181).
  C3  GAIL Walker2d: ilsx_advirl_train's loop (adv_irl.py:126-131: discriminator step, then policy step on relabelled rewards) ==
      oracle.disc + oracle.sac_alpha for 3 iterations, every batch / weight / noise draw rebuilt on the host from the Philox stream.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ C4
def test_c4_ppo_8192x128_iteration_properties():
    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
    from oracle.ppo import gae_one_traj
    o, a, H, n_env, T = 11, 3, 256, 8192, 128
    N = n_env * T
    ctx = ia.Context(0, seed=404)
    try:
        pol = ReparamMultivariateGaussianPolicy([H, H], o, a, conditioned_std=False, hidden_activation="tanh", ctx=ctx, seed=1)
        vf = FlattenMlp([H, H], 1, o, hidden_activation="tanh", ctx=ctx, seed=2)
        kw = dict(mini_batch_size=32768, update_epoch=10, gae_tau=0.95, discount=0.99, reward_scale=1.0)
        tr = PPO(pol, vf, max_samples=N, **kw)
        env = HipVectorEnv("hopper", n_env, seed=0, ctx=ctx, norm_obs=True)
        p0, v0 = tr.get_flat_params(0), tr.get_flat_params(1)
        assert tr.train_from_rollout(env, T, max_path_length=1000) == N
        st = tr.get_eval_statistics()
        _, obs_b, act_b, rew_b, ends_b, lastv = tr._roll
        ends = ends_b.numpy().reshape(n_env, T).astype(bool)
        # a segment closes at every episode end and at the end of the rollout: #segments = #ends + #envs still running at step T
        assert st["PPO Samples"] == N and st["PPO Segments"] == ends.sum() + n_env - ends[:, -1].sum()
        assert ends.sum() > n_env          # random-init Hopper falls within ~20-60 steps: several episodes per env in 128 steps
        p1, v1 = tr.get_flat_params(0), tr.get_flat_params(1)
        assert np.isfinite(p1).all() and np.isfinite(v1).all()
        assert np.abs(p1 - p0).max() > 1e-4 and np.abs(v1 - v0).max() > 1e-4          # 320 minibatch steps moved both networks
        rew = rew_b.numpy()
        assert np.isfinite(rew).all() and np.isfinite(obs_b.numpy()).all() and np.abs(act_b.numpy()).max() < 20
        # GAE of sampled segments == the reference's recurrence (with the post-update value net: recompute through ilsx_ppo_gae)
        cut = ends.copy(); cut[:, -1] = True
        offs = np.concatenate([[0], np.flatnonzero(cut.ravel()) + 1]).astype(np.int32)
        boot = np.zeros(offs.size - 1, np.float32)
        open_env = np.flatnonzero(~ends[:, -1])
        seg_of_last = np.searchsorted(offs, (open_env + 1) * T, side="left") - 1
        boot[seg_of_last] = lastv.numpy()[open_env]
        bdev = ctx.from_numpy(boot)
        outs = [ctx.empty((N,)) for _ in range(4)]
        _lib.check(ctx.lib.ilsx_ppo_gae(tr.h, obs_b.ptr, act_b.ptr, rew_b.ptr, offs.ctypes.data_as(C.c_void_p), offs.size - 1, bdev.ptr,
                                        *[x.ptr for x in outs]))
        R, A, V, lp = [x.numpy() for x in outs]
        assert np.isfinite(R).all() and np.isfinite(A).all() and np.isfinite(lp).all()
        rng = np.random.default_rng(0)
        lens = np.diff(offs)
        picks = list(rng.choice(np.flatnonzero(lens >= 2), 6, replace=False)) + [int(np.argmax(lens)), int(seg_of_last[0])]
        for sgi in picks:
            s0, s1 = offs[sgi], offs[sgi + 1]
            Rr, Ar, _ = gae_one_traj(V[s0:s1, None], rew[s0:s1, None], kw["discount"], kw["gae_tau"], bootstrap=boot[sgi])
            np.testing.assert_allclose(R[s0:s1], Rr[:, 0], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(A[s0:s1], Ar[:, 0], rtol=2e-4, atol=5e-5)       # suffix-scan order + per-segment standardisation
        # every multi-sample segment is standardised on its own (ppo.py:86): mean 0, unbiased std 1
        for sgi in picks:
            seg = A[offs[sgi]:offs[sgi + 1]]
            assert abs(seg.mean()) < 1e-4 and abs(seg.std(ddof=1) - 1) < 1e-3
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------ C5
def test_c5_humanoid_4x1024_grouped_lockstep_matches_oracle():
    import ilswiss_amd as ia
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from oracle import mlp as omlp
    from oracle.sac_alpha import SacAlphaOracle
    o, a, H, B, K, NENV, n_steps = 376, 17, 256, 256, 4, 1024, 3
    kw = dict(reward_scale=1.0, discount=0.99, soft_target_tau=0.005, policy_lr=3e-4, qf_lr=3e-4, alpha=0.2,
              policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3)
    ctx = ia.Context(0, seed=505)
    try:
        envs, rbs, trs, orcs = [], [], [], []
        rng = np.random.default_rng(5)
        for k in range(K):
            env = HipVectorEnv("humanoid", NENV, seed=10 + k, ctx=ctx)
            rb = ia.SimpleReplayBuffer(50_000, o, a, random_seed=100 + k, ctx=ctx)
            for _ in range(4):      # 4096 real Humanoid transitions per seed (random actions, as min_steps_before_training fills the ring)
                env.rollout_step(policy=None, replay=rb, max_path_length=1000, random_actions=True)
            params = (omlp.init_mlp(rng, o, [H, H], a, init_w=1e-3, n_heads=2), omlp.init_mlp(rng, o + a, [H, H], 1),
                      omlp.init_mlp(rng, o + a, [H, H], 1))
            pol = ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx)
            q1, q2 = ia.FlattenMlp([H, H], 1, o + a, ctx=ctx), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx)
            pol.set_flat_params(params[0]), q1.set_flat_params(params[1]), q2.set_flat_params(params[2])
            tr = ia.SoftActorCritic(pol, q1, q2, max_batch=B, **kw)
            tr.eval_statistics = {}
            envs.append(env), rbs.append(rb), trs.append(tr), orcs.append(SacAlphaOracle(o, a, [H, H], *params, **kw))
        assert all(rb.num_steps_can_sample() == 4 * NENV for rb in rbs)
        inputs = [[tr.debug_batch(rb, s, B) for s in range(n_steps)] for tr, rb in zip(trs, rbs)]
        assert not np.array_equal(inputs[0][0][3], inputs[1][0][3])          # every seed draws its own rows
        obs0 = inputs[0][0][0]["observations"]
        assert np.isfinite(obs0).all() and np.abs(obs0).max() > 1.0          # real simulator rows (376-wide), not placeholders
        grp = ia.SoftActorCriticGroup(trs)
        grp.train_from_replay(rbs, n_steps, B)
        for k in range(K):
            for batch, e1, e2, _ in inputs[k]:
                res = orcs[k].train_step(batch, e1, e2)
            assert trs[k].rng_step == n_steps
            for nm, key in (("qf1", "q1_grad"), ("qf2", "q2_grad"), ("policy", "pi_grad")):
                got, ref = trs[k].get_grads(nm), res[key]
                assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (k, nm, np.abs(got - ref).max() / np.abs(ref).max())
            np.testing.assert_allclose(trs[k].log_alpha, orcs[k].log_alpha[0], rtol=0, atol=1e-6)
            for nm, ov in (("policy", orcs[k].pi), ("qf1", orcs[k].q1), ("qf2", orcs[k].q2), ("target_qf1", orcs[k].tq1),
                           ("target_qf2", orcs[k].tq2)):
                np.testing.assert_allclose(trs[k].get_params(nm), ov, rtol=0, atol=5e-5, err_msg=f"agent {k} {nm}")
        grp.close()
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------ C3
def _stream(lib, h, kind):
    from ilswiss_amd import _lib
    s, seed = C.c_uint32(), C.c_uint64()
    _lib.check(lib.ilsx_debug_rng_stream(h, kind, C.byref(s), C.byref(seed)))
    return s.value, seed.value


def test_c3_gail_walker_loop_order_matches_oracles():
    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
    from oracle import mlp as omlp
    from oracle import philox
    from oracle.disc import TANH, DiscOracle, disc_reward
    from oracle.sac_alpha import SacAlphaOracle
    o, a, H, Hd, B, loops = 17, 6, 256, 128, 256, 3                      # gail_walker.yaml: Walker2d dims, 256-256 SAC, 128-128 tanh disc
    sac_kw = dict(reward_scale=2.0, discount=0.99, soft_target_tau=0.005, policy_lr=3e-4, qf_lr=3e-4, alpha=0.2, beta_1=0.25,
                  policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3)
    disc_kw = dict(disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=True, grad_pen_weight=8.0)
    ctx = ia.Context(0, seed=0xC3C3)
    try:
        rng = np.random.default_rng(33)

        def ring(n, shift, seed):
            rb = ia.SimpleReplayBuffer(n, o, a, random_seed=seed, ctx=ctx)
            data = (rng.normal(shift, 1, (n, o)).astype(np.float32), np.tanh(rng.normal(shift, 1, (n, a))).astype(np.float32),
                    rng.normal(0, 1, n).astype(np.float32), (rng.random(n) < 0.02).astype(np.uint8), rng.normal(shift, 1, (n, o)).astype(np.float32))
            rb.add_rows(*data)
            return rb, data
        erb, edata = ring(4000, 0.3, 71)          # 4 expert trajectories x 1000 rows
        prb, pdata = ring(20000, -0.2, 72)        # gail_walker.yaml:46
        params = (omlp.init_mlp(rng, o, [H, H], a, init_w=1e-3, n_heads=2), omlp.init_mlp(rng, o + a, [H, H], 1),
                  omlp.init_mlp(rng, o + a, [H, H], 1))
        dflat = omlp.init_mlp(rng, o + a, [Hd, Hd], 1, init_w=0.2, b_init=0.02)
        pol = ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx)
        q1, q2 = ia.FlattenMlp([H, H], 1, o + a, ctx=ctx), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx)
        pol.set_flat_params(params[0]), q1.set_flat_params(params[1]), q2.set_flat_params(params[2])
        sac = ia.SoftActorCritic(pol, q1, q2, max_batch=B, **sac_kw)
        disc = MLPDisc(o + a, hid_dim=Hd, hid_act="tanh", use_bn=False, ctx=ctx)
        disc.set_flat_params(dflat)
        irl = AdvIRLTrainer("gail2", disc, sac, erb, disc_optim_batch_size=B, policy_optim_batch_size=B, num_update_loops_per_train_call=loops,
                            num_disc_updates_per_loop_iter=1, num_policy_updates_per_loop_iter=1, replay_buffer=prb, **disc_kw)
        (es, eseed), (ps, pseed) = _stream(ctx.lib, erb.h, 0), _stream(ctx.lib, prb.h, 0)
        (ss, cseed), (ds, _) = _stream(ctx.lib, sac.h, 1), _stream(ctx.lib, disc.h, 2)
        assert (eseed, pseed) == (71, 72) and len({es, ps, ss, ss + 1, ds}) == 5
        irl.train(1)                               # ONE ilsx_advirl_train call: 3 x (disc step ; relabel + SAC step)
        dorc = DiscOracle(o + a, Hd, dflat, act=TANH, **disc_kw)
        sorc = SacAlphaOracle(o, a, [H, H], *params, **sac_kw)
        ectr = pctr = 0
        nbuf = ctx.empty((B, a))

        def normals(step, stream):
            _lib.check(ctx.lib.ilsx_debug_philox(ctx.h, C.c_uint64(cseed), C.c_uint64(step), stream, B, a, None, nbuf.ptr))
            return nbuf.numpy().copy()
        rewards_seen = []
        for it in range(loops):
            # adv_irl.py:133-216: expert batch, then policy batch, then the interpolation weights
            ectr += 1; ie = philox.replay_draw(eseed, ectr, es, B, 4000)
            pctr += 1; ip = philox.replay_draw(pseed, pctr, ps, B, 20000)
            xe = np.concatenate([edata[0][ie], edata[1][ie]], 1)
            xp = np.concatenate([pdata[0][ip], pdata[1][ip]], 1)
            dres = dorc.train_step(xe, xp, philox.disc_eps(cseed, it + 1, ds, B))
            # adv_irl.py:238-314: a fresh policy batch, relabelled by the just-updated discriminator, one SAC step on it
            pctr += 1; ib = philox.replay_draw(pseed, pctr, ps, B, 20000)
            batch = dict(observations=pdata[0][ib], actions=pdata[1][ib], terminals=pdata[3][ib].astype(np.float32).reshape(B, 1),
                         next_observations=pdata[4][ib])
            batch["rewards"] = disc_reward(dorc.logits(np.concatenate([batch["observations"], batch["actions"]], 1)), "gail2").astype(np.float32)
            rewards_seen.append(batch["rewards"])
            sres = sorc.train_step(batch, normals(it, ss), normals(it, ss + 1))
        st = irl.get_eval_statistics()
        # the discriminator's statistics are those of the FIRST batch of the epoch (adv_irl.py:205-216), "Disc Rew *" those of the LAST
        # relabelled policy batch: the reference overwrites them after every policy step (:303-314)
        np.testing.assert_allclose(st["Disc Rew Mean"], rewards_seen[-1].mean(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(st["Disc Rew Min"], rewards_seen[-1].min(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(st["Disc Rew Std"], rewards_seen[-1].std(), rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(disc.get_flat_params(), dorc.p, rtol=0, atol=5e-5)
        got, ref = disc.get_flat_grads(), dres["grad"]
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
        for nm, key in (("qf1", "q1_grad"), ("qf2", "q2_grad"), ("policy", "pi_grad")):
            got, ref = sac.get_grads(nm), sres[key]
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (nm, np.abs(got - ref).max() / np.abs(ref).max())
        for nm, ov in (("policy", sorc.pi), ("qf1", sorc.q1), ("qf2", sorc.q2), ("target_qf1", sorc.tq1), ("target_qf2", sorc.tq2)):
            np.testing.assert_allclose(sac.get_params(nm), ov, rtol=0, atol=5e-5, err_msg=nm)
        np.testing.assert_allclose(sac.log_alpha, sorc.log_alpha[0], rtol=0, atol=1e-6)
        # order matters: a policy step relabelled by the discriminator BEFORE its update of the same iteration ends elsewhere
        assert np.abs(rewards_seen[1] - disc_reward(DiscOracle(o + a, Hd, dflat, act=TANH, **disc_kw).logits(
            np.concatenate([pdata[0][ib], pdata[1][ib]], 1)), "gail2")).max() > 1e-3
    finally:
        ctx.close()


def test_c3_broken_window_rolls_the_whole_advirl_call_back():
    """ilsx_advirl_train checkpoints agent AND discriminator (and the rings' draw counters) at entry when its policy steps may run on the
    merged phase kernels; a window that reports a broken hand-off (shared GPU) makes it restore all of them and run the call again on one
    launch per stage.  With ilsx_sac_debug_break_phase armed before the second call, three calls must end bit-exactly where three
    undisturbed calls end — discriminator included."""
    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
    from oracle import mlp as omlp
    o, a, H, Hd, B, loops = 17, 6, 256, 128, 256, 4
    sac_kw = dict(reward_scale=2.0, discount=0.99, soft_target_tau=0.005, policy_lr=3e-4, qf_lr=3e-4, alpha=0.2, beta_1=0.25)
    outs = []
    for broken in (False, True):
        ctx = ia.Context(0, seed=0xC3C4)
        rng = np.random.default_rng(34)

        def ring(n, shift, seed):
            rb = ia.SimpleReplayBuffer(n, o, a, random_seed=seed, ctx=ctx)
            rb.add_rows(rng.normal(shift, 1, (n, o)).astype(np.float32), np.tanh(rng.normal(shift, 1, (n, a))).astype(np.float32),
                        rng.normal(0, 1, n).astype(np.float32), (rng.random(n) < 0.02).astype(np.uint8), rng.normal(shift, 1, (n, o)).astype(np.float32))
            return rb
        erb, prb = ring(4000, 0.3, 71), ring(20000, -0.2, 72)
        params = (omlp.init_mlp(rng, o, [H, H], a, init_w=1e-3, n_heads=2), omlp.init_mlp(rng, o + a, [H, H], 1), omlp.init_mlp(rng, o + a, [H, H], 1))
        dflat = omlp.init_mlp(rng, o + a, [Hd, Hd], 1, init_w=0.2, b_init=0.02)
        pol = ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx)
        q1, q2 = ia.FlattenMlp([H, H], 1, o + a, ctx=ctx), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx)
        pol.set_flat_params(params[0]), q1.set_flat_params(params[1]), q2.set_flat_params(params[2])
        sac = ia.SoftActorCritic(pol, q1, q2, max_batch=B, **sac_kw)
        disc = MLPDisc(o + a, hid_dim=Hd, hid_act="tanh", use_bn=False, ctx=ctx)
        disc.set_flat_params(dflat)
        irl = AdvIRLTrainer("gail2", disc, sac, erb, disc_optim_batch_size=B, policy_optim_batch_size=B, num_update_loops_per_train_call=loops,
                            num_disc_updates_per_loop_iter=1, num_policy_updates_per_loop_iter=1, replay_buffer=prb,
                            disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=True, grad_pen_weight=8.0)
        irl.train(1)
        if broken:
            _lib.check(ctx.lib.ilsx_sac_debug_break_phase(sac.h))
        irl.end_epoch()
        irl.train(1)
        st = dict(irl.get_eval_statistics())
        irl.train(1)
        ps = sac.phase_state()
        assert ps["fallbacks"] == (1 if broken else 0) and ps["disabled"] == broken, ps
        outs.append((disc.get_flat_params().copy(), {k: sac.get_params(k).copy() for k in ("policy", "qf1", "qf2", "target_qf1", "target_qf2")},
                     sac.log_alpha, st))
        ctx.close()
    (d0, s0, la0, st0), (d1, s1, la1, st1) = outs
    np.testing.assert_array_equal(d0, d1)
    for k in s0:
        np.testing.assert_array_equal(s0[k], s1[k], err_msg=k)
    assert la0 == la1
    for k, v in st0.items():
        assert v == st1[k] or (np.isnan(v) and np.isnan(st1[k])), (k, v, st1[k])
