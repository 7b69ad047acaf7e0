#!/usr/bin/env python
"""Secondary benchmark lines for the other BASELINE.json configs; `bench.py` stays the headline.

    python bench_aux.py ppo     # C4: PPO Hopper dims, 8192 envs x 128-step rollout = 1,048,576 samples
    python bench_aux.py gail    # C3: GAIL Walker2d dims, 1 discriminator step + 1 SAC step per loop iteration
    python bench_aux.py td3     # TD3 / SAC-V grad-steps/s at the SAC config's sizes
    python bench_aux.py seeds   # K co-resident SAC seeds on one GPU (multi-stream), aggregate grad-steps/s
    python bench_aux.py humanoid  # C5 per-GPU share: 4 co-resident SAC seeds x 1024 Humanoid envs (obs 376, act 17)

Each prints one JSON line.  Synthetic inputs of the configs' shapes, random-init networks.
"""
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

import ilswiss_amd
from ilswiss_amd import _lib

PEAK_HBM_GBS = 8000.0


PEAK_F32_MFMA_TFLOPS = 157.3


def sac_flops(o, a, H, B):
    """ALGORITHMIC FLOPs of one SAC-alpha gradient step by profiling slot: 0 forward, 1 backward-to-activations,
    2 weight gradients."""
    Wq, Wp = (o + a) * H + H * H + H, o * H + H * H + 2 * H * a
    fwd = 6 * Wq + 2 * Wp
    bwd_dx = 2 * (H * H + H) + 2 * (H + H * H + H * a) + (H * H + 2 * H * a)
    bwd_dw = 2 * Wq + Wp
    # the merged phase kernels (k_sac_phase_a = F1 F2 B1, k_sac_phase_c = F3 B2 B3) mix forward and backward work in one launch
    phase_a = 4 * Wq + 2 * Wp + 2 * (H * H + H)                      # pi(s'), Q1, Q2, pi(s) ; TQ1, TQ2 ; bwd{Q1, Q2 <- TD}
    phase_c = 2 * Wq + 2 * (H + H * H + H * a) + (H * H + 2 * H * a)  # Q1, Q2 (s, a~) ; bwd{Q1, Q2 -> da} ; bwd{pi}
    return {0: 2 * B * fwd, 1: 2 * B * bwd_dx, 2: 2 * B * bwd_dw, "total": 2 * B * (fwd + bwd_dx + bwd_dw),
            "k_sac_phase_a": 2 * B * phase_a, "k_sac_phase_c": 2 * B * phase_c}


def prof_slots(ctx, fn):
    """Run fn() with the library's per-launch HIP-event timing on; {slot: (kernel spelling as launched, launches, total ms)}."""
    lib = ctx.lib
    _lib.check(lib.ilsx_prof_reset(ctx.h))
    _lib.check(lib.ilsx_prof_enable(ctx.h, 1))
    fn()
    _lib.check(lib.ilsx_prof_enable(ctx.h, 0))
    out = {}
    for kid in range(16):
        nl, ms = C.c_uint64(), C.c_double()
        _lib.check(lib.ilsx_prof_read(ctx.h, kid, C.byref(nl), C.byref(ms)))
        if nl.value:
            name = lib.ilsx_prof_kernel(ctx.h, kid).decode().strip("()") or lib.ilsx_kernel_name(kid).decode()
            out[kid] = (name, nl.value, ms.value)
    return out


def mfma_roofline(prof, flops_by_slot, note=None):
    """`roofline` block of the dominant MFMA kernel of a leg: achieved = the slot's algorithmic FLOPs / its summed launch time (= average
    FLOPs per launch / average launch duration)."""
    cand = [k for k in prof if k in flops_by_slot and flops_by_slot[k] > 0]
    if not cand:
        return None
    dom = max(cand, key=lambda k: prof[k][2])
    name, nl, ms = prof[dom]
    ach = flops_by_slot[dom] / (ms * 1e-3) / 1e12
    r = dict(bound="mfma", kernel=name, achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=None,
             avg_launch_us=1e3 * ms / nl, launches=nl, algorithmic_flop_per_launch=flops_by_slot[dom] / nl,
             kernel_ms={prof[k][0]: prof[k][2] for k in prof})
    if note:
        r["note"] = note
    return r


def _kernel_time(ctx, kid, fn):
    lib = ctx.lib
    _lib.check(lib.ilsx_prof_reset(ctx.h))
    _lib.check(lib.ilsx_prof_enable(ctx.h, 1))
    fn()
    _lib.check(lib.ilsx_prof_enable(ctx.h, 0))
    nl, ms = C.c_uint64(), C.c_double()
    _lib.check(lib.ilsx_prof_read(ctx.h, kid, C.byref(nl), C.byref(ms)))
    return nl.value, ms.value


def bench_ppo(ctx):
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
    o, a, H, n_env, T = 11, 3, 256, 8192, 128
    N = n_env * T
    rng = np.random.default_rng(0)
    pol = ReparamMultivariateGaussianPolicy([H, H], o, a, conditioned_std=False, hidden_activation="tanh", ctx=ctx, seed=1)
    vf = FlattenMlp([H, H], 1, o, hidden_activation="tanh", ctx=ctx, seed=2)
    out = {}
    obs = ctx.from_numpy(rng.normal(0, 1, (N, o)).astype(np.float32))
    act = ctx.from_numpy(rng.normal(0, 0.5, (N, a)).astype(np.float32))
    rew = ctx.from_numpy(rng.normal(1, 1, (N,)).astype(np.float32))
    offs = (np.arange(n_env + 1) * T).astype(np.int32)
    for mb, epochs, tag in ((32768, 10, "mb32768"), (64, 1, "mb64_1epoch_64k_samples")):
        tr = PPO(pol, vf, mini_batch_size=mb, update_epoch=epochs, gae_tau=0.95, max_samples=N)
        n_use = N if mb > 64 else 65536
        ntraj = n_use // T
        call = lambda: _lib.check(ctx.lib.ilsx_ppo_train(tr.h, obs.ptr, act.ptr, rew.ptr, offs.ctypes.data_as(C.c_void_p),  # noqa: E731
                                                         ntraj, None, None))
        call(); ctx.sync()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            call()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        out[tag] = dict(samples=n_use, update_epoch=epochs, mini_batch_size=mb, train_step_s=dt,
                        sample_updates_per_s=n_use * epochs / dt, minibatch_steps_per_s=epochs * -(-n_use // mb) / dt)
        if mb > 64:
            gae = lambda: _lib.check(ctx.lib.ilsx_ppo_gae(tr.h, obs.ptr, act.ptr, rew.ptr, offs.ctypes.data_as(C.c_void_p),  # noqa: E731
                                                          n_env, None, None, None, None, None))
            nl, ms = _kernel_time(ctx, 12, lambda: [gae() for _ in range(20)])
            us = ms * 1e3 / nl
            alg = 16.0 * N   # values + rewards in, returns + advantages out (fp32); SURVEY §8d counts 5 streams = 20 B
            out["gae"] = dict(kernel="k_ppo_gae", avg_launch_us=us, algorithmic_bytes_per_launch=alg,
                              achieved_GBps=alg / (us * 1e-6) / 1e9, frac_of_hbm_peak=alg / (us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                              samples=N, trajectories=n_env)
            t0 = time.perf_counter()
            for _ in range(5):
                gae()
            ctx.sync()
            out["calc_adv_s"] = (time.perf_counter() - t0) / 5   # vf forward + GAE + fixed log-probs over 1M rows
    # end to end on the HIP Hopper stepper: rollout (policy -> physics -> record -> running obs statistics) + calc_adv + update
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    env = HipVectorEnv("hopper", n_env, seed=0, ctx=ctx, norm_obs=True)
    tr = PPO(pol, vf, mini_batch_size=32768, update_epoch=10, gae_tau=0.95, max_samples=N)
    tr.train_from_rollout(env, T, max_path_length=1000)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        tr.train_from_rollout(env, T, max_path_length=1000)
    ctx.sync()
    it = (time.perf_counter() - t0) / 3
    obs_b, act_b, rew_b, ends_b, lastv = tr._roll[1:]
    t0 = time.perf_counter()
    _lib.check(ctx.lib.ilsx_ppo_rollout(tr.h, env.h, T, 1000, obs_b.ptr, act_b.ptr, rew_b.ptr, ends_b.ptr, lastv.ptr))
    ctx.sync()
    roll = time.perf_counter() - t0
    out["end_to_end"] = dict(iteration_s=it, rollout_s=roll, env_steps_per_s_rollout=N / roll, samples_per_s_whole_iteration=N / it)
    # roofline of the update's dominant MFMA kernel: one ilsx_ppo_train call on the rollout just collected (calc_adv + 10 x 32 minibatches)
    offs = (np.arange(n_env + 1) * T).astype(np.int32)
    prof = prof_slots(ctx, lambda: (_lib.check(ctx.lib.ilsx_ppo_train(tr.h, obs_b.ptr, act_b.ptr, rew_b.ptr, offs.ctypes.data_as(C.c_void_p),
                                                                       n_env, None, None)), ctx.sync()))
    Wv, Wp = o * H + H * H + H, o * H + H * H + H * a
    steps = 10 * (N // 32768)
    fl = {0: 2.0 * N * (Wv + Wp) + steps * 2.0 * 32768 * (Wv + Wp),                  # calc_adv's two full forwards + every minibatch forward
          1: steps * 2.0 * 32768 * ((H + H * H) + (H * a + H * H)),
          2: steps * 2.0 * 32768 * (Wv + Wp)}
    roof = mfma_roofline(prof, fl, "slot FLOPs: forward = 2N(Wv+Wp) [calc_adv] + 320 minibatches x 2*32768*(Wv+Wp); backward-to-activations and "
                                   "weight gradients per minibatch likewise (Wv = oH+HH+H, Wp = oH+HH+Ha)")
    # memory-side bytes per launch of that kernel at 32768-row minibatches, from the newest committed counter summary (tools/pmc_ppo.sh)
    try:
        src = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_ppopmc_summary.json")))[-1]
        summ = json.load(open(src))
        rows = summ.get(roof["kernel"].split("<")[0], [])
        try:   # do the counters describe the kernels this run executes?  (tools/pmc_ppo.sh records the hash of the library's sources)
            from bench import csrc_sha256
            roof["traffic_stale"] = summ.get("_meta", {}).get("csrc_sha256") != csrc_sha256()
        except Exception:
            roof["traffic_stale"] = None
        if rows:
            mb = min(rows, key=lambda r: r["grid"])     # the minibatch launches (calc_adv's 1M-row forwards have the larger grid)
            roof["traffic"] = mb["read_bytes"] + mb["written_bytes"]
            roof["traffic_source"] = "profiles/" + os.path.basename(src)
    except Exception:
        pass
    g = out["gae"]
    return dict(roofline=roof,
                roofline_gae=dict(bound="hbm", kernel="k_ppo_gae", achieved=g["achieved_GBps"], peak=PEAK_HBM_GBS, unit="GB/s",
                                  frac=g["frac_of_hbm_peak"], traffic=None, avg_launch_us=g["avg_launch_us"],
                                  algorithmic_bytes_per_launch=g["algorithmic_bytes_per_launch"]),
                **_ppo_line(out))


def _ppo_line(out):
    return dict(metric="PPO Hopper-v2 dims, 8192 envs x 128-step rollout, GAE + minibatch update", unit="sample-updates/s",
                value=out["mb32768"]["sample_updates_per_s"], dtype="f32", data="synthetic",
                config=dict(workload="o=11,a=3, tanh 256-256 policy + value net, gamma .99, lambda .95, clip .2, 10 epochs; "
                                     "minibatch 32768 at 1,048,576 samples and the reference's 64 at 65,536 samples (ppo_hopper.yaml:41-50)"),
                detail=out)


def bench_gail(ctx):
    from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.replay import SimpleReplayBuffer
    from ilswiss_amd.sac import SoftActorCritic
    o, a, H, B = 17, 6, 256, 256
    rng = np.random.default_rng(0)
    pol = ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=1)
    q1, q2 = FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=2), FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3)
    sac = SoftActorCritic(pol, q1, q2, reward_scale=2.0, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, beta_1=0.25,
                          max_batch=B)
    disc = MLPDisc(o + a, hid_dim=128, hid_act="tanh", use_bn=False, clamp_magnitude=10.0, ctx=ctx, seed=4)

    def fill(rb, n):
        rb.add_rows(rng.normal(0, 1, (n, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (n, a))).astype(np.float32),
                     rng.normal(0, 1, n).astype(np.float32), (rng.random(n) < 1e-3), rng.normal(0, 1, (n, o)).astype(np.float32))
    exp_rb = SimpleReplayBuffer(4000, o, a, random_seed=1, ctx=ctx)   # 4 expert trajectories x 1000 rows
    rb = SimpleReplayBuffer(20000, o, a, random_seed=2, ctx=ctx)      # gail_walker.yaml:46
    fill(exp_rb, 4000), fill(rb, 20000)
    alg = AdvIRLTrainer("gail2", disc, sac, exp_rb, replay_buffer=rb, disc_optim_batch_size=B, policy_optim_batch_size=B,
                        num_disc_updates_per_loop_iter=1, num_policy_updates_per_loop_iter=1, disc_lr=3e-4, disc_momentum=0.9,
                        grad_pen_weight=8.0)   # gail_walker.yaml:55-66
    sac.eval_statistics, alg.disc_eval_statistics = {}, {}   # steady state: no statistics read-back
    alg.train(200); ctx.sync()
    n = 2000
    t0 = time.perf_counter()
    alg.train(n)
    ctx.sync()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(n):
        alg._do_reward_training()
    ctx.sync()
    dtd = time.perf_counter() - t0
    npf = 200
    prof = prof_slots(ctx, lambda: (alg.train(npf), ctx.sync()))
    D, Hd = o + a, 128
    Wd = D * Hd + Hd * Hd + Hd
    sf = sac_flops(o, a, H, B)
    phase = 13 in prof   # the SAC steps ran on the merged phase kernels (library slots 13 / 14): their forwards / backwards are not in slots 0 / 1
    fl = {0: npf * ((0 if phase else sf[0]) + 2.0 * (3 * B + B) * Wd),                # SAC forwards + discriminator forward (3B rows) + relabel forward (B rows)
          1: npf * (0 if phase else sf[1]),
          13: npf * sf["k_sac_phase_a"] if phase else 0, 14: npf * sf["k_sac_phase_c"] if phase else 0,
          2: npf * (sf[2] + 2.0 * (4 * B * (D * Hd + Hd * Hd) + 3 * B * Hd)),           # + the discriminator's row-stacked weight-gradient jobs
          11: npf * 2.0 * (2 * B * Hd * Hd + B * (3 * Hd * Hd + 2 * D * Hd))}         # k_disc_bwd: GEMM 1 on all rows, GEMM 2/3 + the two W1 contractions on GP rows
    roof = mfma_roofline(prof, fl, "per loop iteration: 1 discriminator step (k_disc_prep, forward over 3B rows, k_disc_bwd, stacked dW, Adam, tail) + "
                                   "relabel forward + 1 SAC step (4 launches: phase A, dW{Q}, phase C, dW{pi}; the first step of a train call: 8)")
    return dict(roofline=roof, metric="GAIL Walker2d-v2 dims: discriminator step + SAC step", unit="loop-iterations/s", value=n / dt,
                dtype="f32", data="synthetic",
                config=dict(workload="o=17,a=6; disc 23-128-128-1 tanh, B=256+256, WGAN-GP weight 8; SAC 256-256, B=256, "
                                     "reward_scale 2, beta_1 0.25; expert buffer 4x1000 rows (gail_walker.yaml)"),
                detail=dict(loop_iterations_per_s=n / dt, disc_steps_per_s_alone=n / dtd, us_per_iteration=1e6 * dt / n))


def bench_td3(ctx):
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.replay import SimpleReplayBuffer
    from ilswiss_amd.sac_v import SoftActorCriticV
    from ilswiss_amd.td3 import TD3, MlpGaussianNoisePolicy
    o, a, H, B = 11, 3, 256, 256
    rng = np.random.default_rng(0)
    rb = SimpleReplayBuffer(100000, o, a, random_seed=2, ctx=ctx)
    n = 100000
    rb.add_rows(rng.normal(0, 1, (n, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (n, a))).astype(np.float32),
                 rng.normal(0, 1, n).astype(np.float32), (rng.random(n) < 1e-3), rng.normal(0, 1, (n, o)).astype(np.float32))
    mk = lambda i, s: FlattenMlp([H, H], 1, i, ctx=ctx, seed=s)  # noqa: E731
    td3 = TD3(MlpGaussianNoisePolicy([H, H], o, a, policy_noise=0.2, output_activation="tanh", ctx=ctx, seed=1), mk(o + a, 2), mk(o + a, 3),
              policy_lr=3e-4, qf_lr=3e-4, max_batch=B)
    sv = SoftActorCriticV(ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=4), mk(o + a, 5), mk(o + a, 6), mk(o, 7),
                          alpha=0.2, policy_lr=3e-4, qf_lr=3e-4, vf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
    res = {}
    for name, tr in (("td3", td3), ("sac_v", sv)):
        tr.eval_statistics = {}
        tr.train_from_replay(rb, 200, B); ctx.sync()
        t0 = time.perf_counter()
        tr.train_from_replay(rb, 2000, B)
        ctx.sync()
        res[name + "_grad_steps_per_s"] = 2000 / (time.perf_counter() - t0)
    return dict(metric="TD3 / SAC-V grad-steps/s, Hopper dims, 256-256 MLP, batch 256", unit="grad-steps/s",
                value=res["td3_grad_steps_per_s"], dtype="f32", data="synthetic", detail=res)


def bench_seeds(_ctx):
    """Co-resident seeds on one GPU (SURVEY config 5's "4 seeds per GPU"): K independent SAC runs, each with its own
    context = HIP stream, replay ring and captured step graph, issued round-robin from one host thread."""
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.replay import SimpleReplayBuffer
    from ilswiss_amd.sac import SoftActorCritic
    o, a, H, B, CAP = 11, 3, 256, 256, 200_000
    rng = np.random.default_rng(0)
    data = (rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
            rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
    res = {}
    for K in (1, 2, 4, 8):
        runs = []
        for k in range(K):
            c = ilswiss_amd.Context(0, seed=100 * k)
            rb = SimpleReplayBuffer(CAP, o, a, random_seed=k, ctx=c)
            rb.add_rows(*data)
            tr = SoftActorCritic(ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=c, seed=k),
                                 FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 1), FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 2),
                                 policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
            tr.eval_statistics = {}
            runs.append((c, rb, tr))
        for c, rb, tr in runs:
            tr.train_from_replay(rb, 300, B)
        for c, rb, tr in runs:
            c.sync()
        n, chunk = 4000, 500
        t0 = time.perf_counter()
        for _ in range(n // chunk):
            for c, rb, tr in runs:
                tr.train_from_replay(rb, chunk, B)
        for c, rb, tr in runs:
            c.sync()
        dt = time.perf_counter() - t0
        res[f"streams_K{K}"] = dict(aggregate_grad_steps_per_s=K * n / dt, per_run=n / dt)
        for c, rb, tr in runs:
            c.close()
    # grouped launches: ONE context, every stage of the step is one launch for all K agents (ilsx_sac_group)
    from ilswiss_amd.sac import SoftActorCriticGroup
    for K in (1, 2, 4, 8, 16):
        c = ilswiss_amd.Context(0, seed=7)
        rbs, trs = [], []
        for k in range(K):
            rb = SimpleReplayBuffer(CAP, o, a, random_seed=k, ctx=c)
            rb.add_rows(*data)
            tr = SoftActorCritic(ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=c, seed=k),
                                 FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 1), FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 2),
                                 policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
            tr.eval_statistics = {}
            rbs.append(rb), trs.append(tr)
        grp = SoftActorCriticGroup(trs)
        grp.train_from_replay(rbs, 300, B)
        c.sync()
        n = 3000
        t0 = time.perf_counter()
        grp.train_from_replay(rbs, n, B)
        c.sync()
        dt = time.perf_counter() - t0
        res[f"grouped_K{K}"] = dict(aggregate_grad_steps_per_s=K * n / dt, per_run=n / dt, us_per_lockstep=1e6 * dt / n)
        grp.close()
        c.close()
    return dict(metric="co-resident SAC seeds on one GPU: aggregate grad-steps/s", unit="grad-steps/s (aggregate)",
                value=res["grouped_K8"]["aggregate_grad_steps_per_s"], dtype="f32", data="synthetic",
                config=dict(workload="K independent SAC runs, Hopper dims, 256-256 MLP, batch 256 each; 'streams' = one HIP stream + "
                                     "step graph per run, 'grouped' = ilsx_sac_group (one launch per stage for all runs)"),
                detail=res)


def bench_humanoid(ctx, R=None):
    """BASELINE config 5's share of ONE GPU: 4 seeds x 1024 Humanoid-v2 envs (obs 376 / act 17 / 256-256 SAC, batch 256), the four
    runs stepped in lockstep by ilsx_sac_group.  Reports the 3-D stepper alone, the grouped SAC step alone, and the loop
    (1 vec-env step per seed : 250 grad steps per seed — the 4096 : 1000 ratio of the headline config at 1024 envs)."""
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from ilswiss_amd.networks import FlattenMlp, ReparamTanhMultivariateGaussianPolicy
    from ilswiss_amd.replay import SimpleReplayBuffer
    from ilswiss_amd.sac import SoftActorCritic, SoftActorCriticGroup
    o, a, H, B, K, N, CAP = 376, 17, 256, 256, 4, 1024, 200_000
    envs, rbs, trs, pols = [], [], [], []
    # every run in a context of its own (same device, its own stream and Philox key), as run_experiment.py --group 4 builds them: the four
    # 1024-env stepper launches (one wavefront per env, 6 envs per CU: 1536 slots) then overlap instead of queueing on one stream
    ctxs = [ctx] + [ctx.sibling(ctx.seed + 1 + k) for k in range(K - 1)]
    for k, c in enumerate(ctxs):
        env = HipVectorEnv("humanoid", N, seed=10 + k, ctx=c)
        rb = SimpleReplayBuffer(CAP, o, a, random_seed=k, ctx=c)
        pol = ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=c, seed=k)
        tr = SoftActorCritic(pol, FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 1), FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 2),
                             policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
        tr.eval_statistics = {}
        envs.append(env), rbs.append(rb), trs.append(tr), pols.append(pol)
    grp = SoftActorCriticGroup(trs, ctx=ctx)
    _sync1 = ctx.sync

    def sync_all():
        for c in ctxs:
            c.sync() if c is not ctx else _sync1()
    import ctypes as C
    _arr = lambda xs: (C.c_void_p * K)(*[x.h for x in xs])   # noqa: E731
    _envs, _pols, _rbs, _mins = _arr(envs), _arr(pols), _arr(rbs), (C.c_int64 * K)(*([0] * K))

    def lockstep(n=1):   # n lock-step vec steps of the four runs: what DeviceRLAlgorithmGroup calls between two train triggers
        from ilswiss_amd import _lib
        _lib.check(ctx.lib.ilsx_rollout_steps_lockstep(_envs, _pols, _rbs, K, n, 1000, _mins, 0, 0))
    for t in range(12):     # fill: 12 x 1024 transitions per seed, random actions (min_steps_before_training)
        for env, rb in zip(envs, rbs):
            env.rollout_step(policy=None, replay=rb, max_path_length=1000, random_actions=True)
    sync_all()
    res = {}
    n_env_steps = 30
    nl, ms = _kernel_time(ctx, 9, lambda: [envs[0].rollout_step(policy=pols[0], replay=rbs[0], max_path_length=1000) for _ in range(n_env_steps)] and ctx.sync())
    res["env_step_kernel_ms_1024_envs"] = ms / max(nl, 1)
    t0 = time.perf_counter()
    for _ in range(n_env_steps):
        for env, rb, p in zip(envs, rbs, pols):
            env.rollout_step(policy=p, replay=rb, max_path_length=1000)
        sync_all()
    dt = time.perf_counter() - t0
    res["rollout_env_steps_per_s_one_run_at_a_time"] = K * N * n_env_steps / dt
    t0 = time.perf_counter()
    lockstep(n_env_steps)
    sync_all()
    dt = time.perf_counter() - t0
    res["rollout_env_steps_per_s"] = K * N * n_env_steps / dt
    grp.train_from_replay(rbs, 100, B); sync_all()
    n = 1000
    t0 = time.perf_counter()
    grp.train_from_replay(rbs, n, B)
    sync_all()
    dt = time.perf_counter() - t0
    res["grouped_grad_steps_per_s"] = K * n / dt
    res["us_per_lockstep"] = 1e6 * dt / n
    iters, per = 6, 250
    if R is not None:     # bench.py --gpus N: every rank runs this leg on its own GPU; the loop is timed between two barriers
        R.barrier(ctx)
    t0 = time.perf_counter()
    for _ in range(iters):
        lockstep(1)
        sync_all()                      # the group loop waits for every run's stream before a train call (algorithm.py)
        grp.train_from_replay(rbs, per, B)
    sync_all()
    dt = time.perf_counter() - t0
    res["loop_grad_steps_per_s"] = K * per * iters / dt
    res["loop_env_steps_per_s"] = K * N * iters / dt
    if R is not None:
        dt_max = R.max_over_ranks([dt])[0]
        res["ranks"] = R.world
        res["loop_grad_steps_per_s_over_ranks"] = R.world * K * per * iters / dt_max
        res["loop_env_steps_per_s_over_ranks"] = R.world * K * N * iters / dt_max
    ep, ret = 0, 0.0
    for env in envs:
        e_, r_ = env.rollout_stats()
        ep, ret = ep + e_, ret + r_
    res["episodes"], res["mean_return"] = ep, (ret / ep if ep else None)
    npf = 50
    prof = prof_slots(ctx, lambda: (grp.train_from_replay(rbs, npf, B), ctx.sync()))
    sf = sac_flops(o, a, H, B)
    roof = mfma_roofline(prof, {k: K * npf * sf[k] for k in (0, 1, 2)},
                         "grouped launches carry all 4 seeds; wide inputs run the forward as two launches (layer 0, then layer 1 + heads)")
    res["stepper_kernel"] = dict(kernel="k_env3dw_step<23>", avg_launch_ms_1024_envs=res["env_step_kernel_ms_1024_envs"],
                                 bound="neither (fp64 issue-bound, one wavefront per env; DESIGN 3b)")
    grp.close()
    for e_ in envs:
        e_.close()
    for c in ctxs[1:]:
        c.close()
    return dict(roofline=roof, metric="SAC Humanoid-v2 share of one GPU: 4 seeds x 1024 envs, aggregate grad-steps/s in the loop", unit="grad-steps/s (aggregate)",
                value=res["loop_grad_steps_per_s"], dtype="f32 (networks) / f64 (stepper)", data="synthetic",
                config=dict(workload="4 co-resident SAC runs, Humanoid-v2 model (obs 376, act 17), 1024 envs each, 256-256 MLP, batch 256, "
                                     "1 vec-env step : 250 grad steps per run"), detail=res)


if __name__ == "__main__":
    ctx = ilswiss_amd.Context(0, seed=0)
    which = sys.argv[1:] or ["ppo", "gail", "td3", "seeds", "humanoid"]
    for w in which:
        print(json.dumps(dict(ppo=bench_ppo, gail=bench_gail, td3=bench_td3, seeds=bench_seeds, humanoid=bench_humanoid)[w](ctx)), flush=True)
