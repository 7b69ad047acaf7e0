#!/usr/bin/env python
"""bench.py — the BASELINE.json metric on MI355X: env-steps/s + SAC grad-steps/s, Hopper-v2 dims,
4096 parallel envs, 256-256 MLPs, batch 256 (config C2).

One "step" = one iteration of the reference's outer loop (rlkit/core/base_algorithm.py:183-286) at
env_num = 4096 with the YAML knobs of exp_specs/sac/sac_hopper.yaml:17-20: ONE vec-env step of all 4096
envs (policy inference -> physics -> replay insert; 4096 env-steps) followed by ONE train call of
num_train_steps_per_train_call = 1000 SAC gradient steps (on-device replay sampling + update).
`value` is the whole-loop gradient-step rate (grad steps / total wall time, max over ranks);
`env_steps_per_s` / `grad_steps_per_s_train_phase` are the reference's own phase-timed definitions
(Sample Time / Train Time, base_algorithm.py:284-290,329-343).

N > 1: one process per GPU, independent replicas (seeds shard with no data-path collective,
run_experiment.py:57-78) -> weak scaling; torch.distributed (RCCL) only for the barrier + max-time.  The ranks come
either from the launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the env) or, when `--gpus N` is given WITHOUT a RANK in the env, from bench.py itself: it
spawns N copies of itself, one per GPU, on 127.0.0.1 and relays rank 0's JSON line.  With N > 1 the line also carries
`split_run`: ONE run split over the N GPUs (B/N rows per rank, RCCL all-reduce of the gradient arena on the library's
stream between backward and update) — the only leg with a data-path collective, never `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

O, A, H, B = 11, 3, 256, 256          # Hopper-v2 obs/act, hidden width, batch (BASELINE.json)
N_ENV = 4096
GRAD_PER_CALL = 1000                  # sac_hopper.yaml:20
REPLAY_CAP = 1_000_000                # sac_hopper.yaml:28
SAC_KW = dict(reward_scale=1.0, discount=0.99, soft_target_tau=0.005, policy_lr=3e-4, qf_lr=3e-4, alpha=0.2,
              policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3)  # sac_hopper.yaml:36-47
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0


def _latest_pmc_summary():
    """Newest committed counter summary (profiles/rNN_*_pmc_summary.json, written by tools/profile_round.sh); the bench line
    names the file it read so that a stale one is visible."""
    import glob
    import re
    # rNN_pmc_summary.json / rNN_x_pmc_summary.json only: the stepper / grouped / PPO counter summaries (rNN_envpmc_..., rNN_grppmc_..., rNN_ppopmc_...) have other contents
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*pmc_summary.json"))
                   if re.fullmatch(r"r\d\d(_[a-z0-9]+)?_pmc_summary\.json", os.path.basename(f)))
    return files[-1] if files else None


PMC_SUMMARY = _latest_pmc_summary()
PMC_NAMES = {0: ("k_mlp2_fwd_split", "k_mlp_fwd"), 1: ("k_mlp2_bwd_split", "k_mlp_bwd_dx"), 2: ("k_mlp_bwd_dw",),
             6: ("k_replay_sample_many",), 13: ("k_sac_phase_a",), 14: ("k_sac_phase_c",)}
MLP_SLOTS = (0, 1, 2, 13, 14)   # library profiling slots of the step's MFMA kernels (13 / 14: the merged phase kernels)


def csrc_sha256():
    """Hash of the library's sources, as tools/pmc_summary.py records it in a counter summary's `_meta`."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ilswiss_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ilswiss_amd", "csrc", "*.h"))
                    + glob.glob(os.path.join(ROOT, "ilswiss_amd", "csrc", "*.inc")) + [os.path.join(ROOT, "include", "ilsx.h")]):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


def pmc_traffic_stale():
    """True when the committed counter summary was collected on other kernel sources than the ones this run executes (or does not say):
    `roofline.traffic` then describes older kernels and the line says so."""
    try:
        with open(PMC_SUMMARY) as f:
            return json.load(f).get("_meta", {}).get("csrc_sha256") != csrc_sha256()
    except (OSError, TypeError, ValueError, AttributeError):   # absent, unreadable or malformed: the line must come out
        return None


def pmc_traffic(kid):
    """HBM bytes per launch of profiling id `kid` from the committed rocprofv3 counter passes (tools/pmc_collect.sh ->
    profiles/r01_g_pmc_summary.json): FETCH_SIZE x 1024 x 2 (gfx950 correction of MI355X_MICROARCH.md §HBM) + WRITE_SIZE x
    1024, each collected in its own --pmc pass over the same kernels at the same sizes.  None if the file is absent."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
    except (OSError, TypeError, ValueError):
        return None
    if not isinstance(d, dict):
        return None
    for name in PMC_NAMES.get(kid, ()):
        k = d.get(name)
        if k and "hbm_read_bytes_corrected" in k and "hbm_write_bytes_raw" in k:
            return k["hbm_read_bytes_corrected"] + k["hbm_write_bytes_raw"]
    return None


def grouped_insts_per_mfma():
    """Non-MFMA instructions a wave of the grouped (K = 8) lock-step kernels issues per MFMA, per kernel family, from the newest committed
    counter summary of tools/pmc_grp.sh (profiles/rNN_grppmc_summary.json: SQ_INSTS_{VALU, SALU, LDS, SMEM, VMEM_RD, VMEM_WR, MFMA} summed over
    all waves of all launches).  On this part fp32 MFMA time and the VALU / LDS / memory issue of the same SIMD ADD, so this
    ratio — not occupancy — is what the launches' length is made of."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_grppmc_summary.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if not isinstance(d, dict):
        return None
    out = dict(source=os.path.relpath(files[-1], ROOT), stale=d.get("_meta", {}).get("csrc_sha256") != csrc_sha256())
    for k in ("k_mlp2_fwd_split", "k_mlp2_bwd_split", "k_mlp_bwd_dw"):
        v = d.get(k) or d.get(k + "_low")   # the grouped weight-gradient tile's eight-waves-per-SIMD instance (k_mlp_bwd_dw_low)
        if not v or not v.get("SQ_INSTS_MFMA"):
            continue
        mf = v["SQ_INSTS_MFMA"]
        other = v.get("SQ_INSTS_VALU", 0.0) - mf + sum(v.get(c, 0.0) for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        out[k] = dict(non_mfma_per_mfma=other / mf, salu_per_wave=v.get("SQ_INSTS_SALU", 0.0) / max(1.0, v.get("SQ_WAVES", 1.0)),
                      mfma_per_wave=mf / max(1.0, v.get("SQ_WAVES", 1.0)))
    return out


def co_resident_seeds(K=8, n=1500, R=None):
    """SURVEY config 5's per-GPU shape ("several seeds per GPU"): K independent SAC runs of the same config stepped in
    lock-step by ilsx_sac_group (one launch per stage for all of them).  Reported beside the headline, never as `value`.
    The MFMA roofline entry is the grouped forward launch (K x 4 tasks x 16 row tiles x 4 column slices workgroups).
    R (N > 1): every rank runs the leg on its own GPU between two barriers; `aggregate_over_ranks` = all ranks' steps / the slowest rank's time."""
    import ctypes as C

    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    rank, world, local = (R.rank, R.world, R.local) if R is not None else (0, 1, 0)
    c = ia.Context(local, seed=4242 + rank)
    rng = np.random.default_rng(7 + rank)
    cap = 200_000
    rows = synth_rows(rng, cap)
    rbs, trs = [], []
    for k in range(K):
        rb = ia.SimpleReplayBuffer(cap, O, A, random_seed=k, ctx=c)
        rb.add_rows(*rows)
        tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy([H, H], O, A, ctx=c, seed=3 * k),
                                ia.FlattenMlp([H, H], 1, O + A, ctx=c, seed=3 * k + 1),
                                ia.FlattenMlp([H, H], 1, O + A, ctx=c, seed=3 * k + 2), max_batch=B, **SAC_KW)
        tr.eval_statistics = {}
        rbs.append(rb), trs.append(tr)
    grp = ia.SoftActorCriticGroup(trs)
    grp.train_from_replay(rbs, 200, B)
    c.sync()
    if R is not None:
        R.barrier(c)
    t0 = time.perf_counter()
    grp.train_from_replay(rbs, n, B)
    c.sync()
    dt = time.perf_counter() - t0
    dt_max = R.max_over_ranks([dt])[0] if R is not None else dt
    _lib.check(c.lib.ilsx_prof_reset(c.h))
    _lib.check(c.lib.ilsx_prof_enable(c.h, 1))
    grp.train_from_replay(rbs, 100, B)
    _lib.check(c.lib.ilsx_prof_enable(c.h, 0))
    nl, ms = C.c_uint64(), C.c_double()
    _lib.check(c.lib.ilsx_prof_read(c.h, 0, C.byref(nl), C.byref(ms)))
    fl = K * flops_per_step()[0] / (nl.value / 100.0)
    avg_s = ms.value * 1e-3 / nl.value
    out = dict(K=K, aggregate_grad_steps_per_s=K * n / dt, per_run_grad_steps_per_s=n / dt, us_per_lockstep=1e6 * dt / n,
               ranks=world, aggregate_over_ranks=world * K * n / dt_max, insts_per_mfma=grouped_insts_per_mfma(),
               roofline=dict(bound="mfma", kernel=kernel_spelling(c.lib, c, 0), achieved=fl / avg_s / 1e12, peak=PEAK_F32_MFMA_TFLOPS,
                             unit="TFLOP/s", frac=fl / avg_s / 1e12 / PEAK_F32_MFMA_TFLOPS, traffic=None, avg_launch_us=avg_s * 1e6,
                             algorithmic_flop_per_launch=fl))
    grp.close()
    c.close()
    return out


def kernel_spelling(lib, ctx, kid):
    """The kernel a profiling slot last launched, as spelled at its launch site (e.g. k_mlp2_bwd_split<256, ACT_RELU, 4, false>:
    the name rocprofv3 lists it under, up to the enum spelling); falls back to the slot's generic name."""
    name = lib.ilsx_prof_kernel(ctx.h, kid).decode().strip("()")
    return name or lib.ilsx_kernel_name(kid).decode()


def flops_per_step():
    """ALGORITHMIC FLOPs of one SAC-alpha gradient step, split by kernel."""
    from bench_aux import sac_flops
    return sac_flops(O, A, H, B)


def synth_rows(rng, n):
    """C2 synthetic transitions: obs,next_obs ~ N(0,1); act = tanh(N(0,1)); rew ~ N(0,1);
    done ~ Bernoulli(1e-3)."""
    return (rng.standard_normal((n, O), dtype=np.float32), np.tanh(rng.standard_normal((n, A), dtype=np.float32)),
            rng.standard_normal(n, dtype=np.float32), (rng.random(n) < 1e-3).astype(np.uint8),
            rng.standard_normal((n, O), dtype=np.float32))


def cpu_baseline(budget_s=18.0, full=False):
    """SURVEY §8d's CPU legs on the GPU box's host cores, same synthetic inputs as the GPU run (C2):
      * `value`: the numpy oracle (oracle/sac_alpha.py, hand-written backward) — replay gather + train_step, all BLAS threads;
      * `torch_cpu`: the PyTorch-CPU restatement (oracle/sac_alpha_torch.py: autograd + torch.optim.Adam, the reference's own
        idiom, ~1.8k ATen calls per step) at 1 thread and at all cores, train_step alone and with random_batch + conversion;
      * `get_actions_ms`: policy inference for 4096 observations (policies.py:245-246) on the same cores.
    Bounded samples (a few seconds each) so the default run stays within minutes."""
    import torch

    from oracle import mlp as omlp
    from oracle.replay import ReplayOracle
    from oracle.sac_alpha import SacAlphaOracle
    from oracle.sac_alpha_torch import SacAlphaTorch, _mlp
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    hid = [H, H]
    init = (omlp.init_mlp(rng, O, hid, A, init_w=1e-3, n_heads=2), omlp.init_mlp(rng, O + A, hid, 1), omlp.init_mlp(rng, O + A, hid, 1))
    orc = SacAlphaOracle(O, A, hid, *init, **SAC_KW)
    n = 100_000
    rb = ReplayOracle(n, O, A)
    ob, ac, rw, dn, nob = synth_rows(rng, n)
    rb.obs[:], rb.act[:], rb.rew[:, 0], rb.term[:, 0], rb.next_obs[:] = ob, ac, rw, dn, nob
    rb.size = n

    def batch():
        bt = rb.gather(rb.draw_indices(B))
        bt["terminals"] = bt["terminals"].astype(np.float32)
        return bt

    def timed(fn, budget):
        for _ in range(3):
            fn()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget:
            fn()
            k += 1
        return k, time.perf_counter() - t0
    eps = lambda: rng.standard_normal((B, A), dtype=np.float32)   # noqa: E731
    # protocol (BASELINE.md §3 asks 1k warm-up + 10k timed steps, median of 3; that is ~3 minutes of host time at ~200 steps/s, more
    # than the bounded sample this line may take): 200 warm-up steps, then 3 timed windows of budget_s / 3 seconds each, MEDIAN rate
    def counted(fn, steps):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        return steps, time.perf_counter() - t0
    if full:   # --full-cpu-baseline: BASELINE.md §3's protocol to the letter — 1k warm-up steps, 10k timed steps, median of 3 repeats
        for _ in range(1000):
            orc.train_step(batch(), eps(), eps())
        runs = [counted(lambda: orc.train_step(batch(), eps(), eps()), 10000) for _ in range(3)]
        label = "BASELINE.md §3 protocol (--full-cpu-baseline): 1000 warm-up steps, 3 repeats of 10000 timed steps, median rate"
    else:
        for _ in range(200):
            orc.train_step(batch(), eps(), eps())
        runs = [timed(lambda: orc.train_step(batch(), eps(), eps()), budget_s / 3.0) for _ in range(3)]
        label = (f"short protocol (default): 200 warm-up steps, 3 windows of {budget_s / 3.0:.1f} s, median rate; BASELINE.md §3's 1k + 10k x 3 "
                 "protocol (~3 min of host time) runs under --full-cpu-baseline")
    rates = sorted(kk / dd for kk, dd in runs)
    k, dt = sum(r[0] for r in runs), sum(r[1] for r in runs)
    out = dict(value=rates[1], unit="grad-steps/s", cores=int(cores), kind="port",
               sample=f"{k} SAC-alpha grad steps (replay gather + train_step, B={B}, H={H}, Hopper dims) in {dt:.1f} s, "
                      "oracle/sac_alpha.py numpy fp32",
               protocol=f"{label} (min {rates[0]:.1f}, max {rates[2]:.1f})", protocol_name="baseline_md_3" if full else "short")
    tc = {}
    ncpu = os.cpu_count() or 1
    fixed = batch()
    # all logical cores is a pathological setting for a 256-wide MLP (measured on the 256-thread GPU host: 0.05 steps/s against 139 at one
    # thread): the multi-thread point is 16, where intra-op threading of these matrix sizes stops paying
    for threads in (1, min(16, ncpu)):
        torch.set_num_threads(threads)
        ag = SacAlphaTorch(O, A, hid, *init, **SAC_KW)
        k1, d1 = timed(lambda: ag.train_step(fixed, eps(), eps()), 2.0)
        k2, d2 = timed(lambda: ag.train_step(batch(), eps(), eps()), 2.0)
        obs4096 = torch.as_tensor(rng.standard_normal((N_ENV, O), dtype=np.float32))
        with torch.no_grad():
            k3, d3 = timed(lambda: torch.tanh(_mlp(ag.pi_p, obs4096, 2, 2)[0] + torch.randn(N_ENV, A)).numpy(), 0.5)
        tc[f"threads_{threads}"] = dict(train_step_per_s=k1 / d1, sample_convert_train_step_per_s=k2 / d2, get_actions_ms_4096=1e3 * d3 / k3)
    out["torch_cpu"] = dict(tc, note="oracle/sac_alpha_torch.py (autograd + torch.optim.Adam), fp32, same inputs; "
                                     f"host has {ncpu} logical cores")
    out["env_stepper"] = cpu_env_baseline(ncpu)
    return out


def cpu_env_baseline(ncpu):
    """env-steps/s of the scalar compiled restatement of the Hopper stepper (oracle/planar_env.c == oracle/planar_env.py to 1e-10,
    tests/test_env_oracle.py) on the host: one core, and every core with OpenMP over envs.  Uniform actions, auto-reset."""
    from ilswiss_amd.envs.models import MODELS
    from ilswiss_amd.envs.vecenv import model_struct
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_build", "liborc_planar.so")
    if not os.path.exists(so):
        return dict(error="oracle/_build/liborc_planar.so not built (__graft_entry__.build() / make -C oracle)")
    lib = C.CDLL(so)
    lib.orc_planar_bench.restype = C.c_double
    ms, cs, res = model_struct(MODELS["hopper"]()), C.c_double(), {}
    for threads, n_env, n_steps in ((1, 64, 1500), (ncpu, 4096, 250)):
        lib.orc_planar_bench(C.byref(ms), n_env, 20, 1000, threads, C.byref(cs))      # warm (thread pool)
        dt = lib.orc_planar_bench(C.byref(ms), n_env, n_steps, 1000, threads, C.byref(cs))
        res[f"threads_{threads}"] = dict(env_steps_per_s=n_env * n_steps / dt, sample=f"{n_env} envs x {n_steps} steps in {dt:.2f} s")
    res["note"] = "oracle/planar_env.c (gcc -O2, dense formulation of oracle/planar_env.py), Hopper model, uniform actions, auto-reset"
    res["humanoid"] = cpu_env3d_baseline(ncpu)
    return res


def cpu_env3d_baseline(ncpu):
    """The same for BASELINE config 5's env: env-steps/s of oracle/spatial_env.c — the scalar compiled restatement of the
    3-D stepper's oracle (== oracle/spatial_env.py to 1e-9, tests/test_env3d_oracle.py) — on the Humanoid-v2 model, one core and every core."""
    from ilswiss_amd.envs.models3d import humanoid
    from ilswiss_amd.envs.vecenv import spatial_struct
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_build", "liborc_spatial.so")
    if not os.path.exists(so):
        return dict(error="oracle/_build/liborc_spatial.so not built (__graft_entry__.build() / make -C oracle)")
    lib = C.CDLL(so)
    lib.orc_spatial_bench.restype = C.c_double
    ms, cs, res = spatial_struct(humanoid()), C.c_double(), {}
    for threads, n_env, n_steps in ((1, 16, 60), (ncpu, 1024, 200 if ncpu >= 32 else 40)):
        lib.orc_spatial_bench(C.byref(ms), n_env, 2, 1000, threads, C.byref(cs))      # warm (thread pool)
        dt = lib.orc_spatial_bench(C.byref(ms), n_env, n_steps, 1000, threads, C.byref(cs))
        res[f"threads_{threads}"] = dict(env_steps_per_s=n_env * n_steps / dt, sample=f"{n_env} envs x {n_steps} steps in {dt:.2f} s")
    res["note"] = ("oracle/spatial_env.c (gcc -O2, dense Jacobian formulation of oracle/spatial_env.py), Humanoid-v2 model (23 dof, 376-dim "
                   "observation), uniform actions, auto-reset; beside k_env3dw_step<23> in `humanoid_4x1024`")
    return res


def spawn_ranks(n, argv):
    """`--gpus N` without a launcher: N copies of this script, one per GPU (RANK = LOCAL_RANK = i), rendezvous on 127.0.0.1.
    Rank 0's stdout is relayed; a failing rank fails the run."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=r == 0 or None))
    # rank 0's stdout is drained on a thread while ALL children are polled: a rank that dies before rendezvous would otherwise leave
    # rank 0 (and this process, blocked in communicate()) inside init_process_group until torch's own timeout
    import threading
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:
            failed = bad
            for p in procs:
                if p.poll() is None:
                    p.kill()         # exactly the PIDs started above
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    rd.join(timeout=5.0)
    out = "".join(c or "" for c in chunks)
    sys.stdout.write(out)
    sys.stdout.flush()
    if failed is not None or any(rcs):
        raise SystemExit(f"bench.py: ranks exited with {rcs}" + (f" (rank(s) {failed} failed first; the others were stopped)" if failed else ""))
    return out


def pin_to_gpu_numa(local):
    """Best-effort NUMA pinning of this rank's host threads to the node its GPU hangs off (an 8-GPU MI355X box has 2 sockets; a rank
    whose launch thread sits on the far socket pays a cross-socket hop on every doorbell write).  rocm-smi reports the node; anything
    missing -> no pinning.  Returns what was done, for the `scaling` block."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "--showtoponuma", "--json"], capture_output=True, text=True, timeout=20).stdout
        info = json.loads(out)
        card = info.get(f"card{local}") or {}
        node = next((int(v) for k, v in card.items() if "numa node" in k.lower() and str(v).lstrip("-").isdigit()), None)
        if node is None or node < 0:
            return dict(pinned=False, reason="rocm-smi reported no NUMA node")
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus)
        return dict(pinned=True, numa_node=node, cpus=len(cpus))
    except Exception as e:   # noqa: BLE001
        return dict(pinned=False, reason=repr(e)[:120])


RCCL_ENV_KEYS = ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_P2P_LEVEL", "NCCL_DEBUG", "RCCL_MSCCL_ENABLE",
                 "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")


def _flush_c_stdio():
    """fflush(NULL): whatever native libraries (RCCL) left in the C stdio buffers goes out now, not at process exit"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass


class Ranks:
    """Process-group plumbing shared by the real run and the CPU dry run (tests/test_bench_spawn.py): backend "nccl" (= RCCL)
    on GPUs, "gloo" for --dry-run."""

    def __init__(self, dry_run):
        self.rank = int(os.environ.get("RANK", 0))
        self.local = int(os.environ.get("LOCAL_RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.dist, self.dry = None, dry_run
        self.ncoll = 0      # one-word collectives issued so far (barrier / max_over_ranks of one value): what `leg` catches up on
        if self.world > 1 or os.environ.get("ILSX_BENCH_FORCE_DIST"):   # the override exercises the path with one rank
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
            import torch
            import torch.distributed as dist
            if dry_run:
                dist.init_process_group("gloo")
            else:
                torch.cuda.set_device(self.local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local))
            self.dist = dist

    def barrier(self, ctx=None):
        """device work of this rank finished, then every rank here.  The rendezvous is an all-reduce of ONE float64 — the same collective as
        `max_over_ranks([x])` — so that a rank whose leg failed can stand in for the barriers it did not reach (`leg`)."""
        if ctx is not None:
            ctx.sync()
        if self.dist is not None:
            if not self.dry:
                import torch
                torch.cuda.synchronize()
            self.max_over_ranks([0.0])

    def leg(self, fn, n_coll):
        """A secondary leg holding `n_coll` one-word collectives (barriers, `max_over_ranks` of one value).  An exception on this rank is
        recorded in the leg's place and the collectives it did not reach are joined with neutral values: the other ranks are not left
        waiting inside RCCL, their numbers and the headline line still come out."""
        c0 = self.ncoll
        try:
            return fn()
        except Exception as e:   # noqa: BLE001 — secondary leg
            out = dict(error=repr(e)[:300])
            try:
                while self.dist is not None and self.ncoll < c0 + n_coll:
                    self.max_over_ranks([0.0])
            except Exception as e2:   # noqa: BLE001 — the communicator itself is gone: nothing left to keep in step
                out["catch_up_error"] = repr(e2)[:200]
            return out

    def max_over_ranks(self, values):
        if self.dist is None:
            return list(values)
        self.ncoll += len(values) == 1
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if self.dry else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def gather(self, value):
        """value of every rank, in rank order (one float per rank)"""
        if self.dist is None:
            return [float(value)]
        import torch
        t = torch.zeros(self.world, dtype=torch.float64, device="cpu" if self.dry else "cuda")
        t[self.rank] = float(value)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def split_run_leg(R, n=400):
    """ONE SAC run split over the world's GPUs: every rank a full replica, B/G rows per rank from its own replay
    shard, gradient arena all-reduced by RCCL on the library's stream (critics, then actor | alpha).  G = 1 (single-GPU box with
    ILSX_BENCH_FORCE_DIST): the same code path on a one-rank communicator (ILSX_SPLIT_FORCE)."""
    import ilswiss_amd as ia
    from ilswiss_amd.parallel import SplitRunStep
    G = R.world
    if G == 1:
        os.environ["ILSX_SPLIT_FORCE"] = "1"
    if B % G:
        return None
    # identical parameter init on every rank comes from the seeded nets below; the ctx seed keys the Philox policy noise, so the rank is
    # mixed into it — with one seed for all ranks every shard would draw the same eps rows (G-fold correlated noise in the "stratified" batch)
    ctx = ia.Context(R.local, seed=555 + 7919 * R.rank)
    hid = [H, H]
    tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy(hid, O, A, ctx=ctx, seed=1),
                            ia.FlattenMlp(hid, 1, O + A, ctx=ctx, seed=2), ia.FlattenMlp(hid, 1, O + A, ctx=ctx, seed=3),
                            max_batch=B // G, grad_world=G, **SAC_KW)
    tr.eval_statistics = {}
    shard = REPLAY_CAP // G
    rb = ia.SimpleReplayBuffer(shard, O, A, random_seed=100 + R.rank, ctx=ctx)
    rng = np.random.default_rng(100 + R.rank)
    left = shard
    while left > 0:
        k = min(left, 250_000)
        rb.add_rows(*synth_rows(rng, k))
        left -= k
    step = SplitRunStep(tr)
    step.train_from_replay(rb, 50, B // G)
    R.barrier(ctx)
    t0 = time.perf_counter()
    step.train_from_replay(rb, n, B // G)
    R.barrier(ctx)
    dt = R.max_over_ranks([time.perf_counter() - t0])[0]
    # every replica must have taken identical optimiser steps
    chk = float(np.abs(tr.get_params("qf1")).sum())
    lo, hi = -R.max_over_ranks([-chk])[0], R.max_over_ranks([chk])[0]
    # the communicator's own size, read back from the library on EVERY rank (the smallest over ranks is reported: all must say G)
    import ctypes as C
    nr, rk = C.c_int(), C.c_int()
    ctx.lib.ilsx_comm_info(ctx.h, C.byref(nr), C.byref(rk))
    rccl_ranks = int(-R.max_over_ranks([-float(nr.value)])[0])
    fb, dis, last, oa, oc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ctx.lib.ilsx_sac_phase_state(tr.h, C.byref(fb), C.byref(dis), C.byref(last), C.byref(oa), C.byref(oc))
    ctx.close()
    return dict(ranks=G, rccl_ranks=rccl_ranks, rccl_rank_of_rank0=int(rk.value), phase_kernels=bool(last.value) and not dis.value,
                step_form=("A, dW{Q}, all-reduce, Adam{Q}, C, dW{pi}, all-reduce, Adam{pi} (merged phase kernels, deferred tail)" if last.value and not dis.value
                           else "one launch per stage, all-reduce x2, k_sac_stats / k_sac_finish"),
                local_batch=B // G, grad_steps_per_s=n / dt, us_per_step=1e6 * dt / n, replicas_identical=bool(lo == hi),
                allreduce_bytes_per_step=4 * (tr.qf1.num_params * 2 + tr.policy.num_params + 4),
                collective="ncclAllReduce(float32, sum) x2 per step on the ctx stream (librccl via dlopen)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-cpu-baseline", action="store_true",
                    help="time the CPU leg on BASELINE.md §3's protocol (1k warm-up + 3 x 10k timed steps, ~3 min) instead of the short one")
    ap.add_argument("--no-seeds", action="store_true", help="skip the co-resident seeds leg")
    ap.add_argument("--no-aux", action="store_true", help="skip the config 3 / 4 / 5 legs (PPO 8192x128, GAIL Walker2d, Humanoid 4x1024)")
    ap.add_argument("--no-split-run", action="store_true", help="skip the split-run (RCCL all-reduce) leg at N > 1")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU plumbing check (tests): spawn / rendezvous (gloo) / barrier / max-over-ranks / one JSON line, no GPU work")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])
        return None
    R = Ranks(args.dry_run)
    rank, local, world = R.rank, R.local, R.world
    if world != args.gpus and not os.environ.get("ILSX_BENCH_FORCE_DIST"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        R.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.001 * (1 + rank))     # rank r is slower: the max must pick the slowest
        R.barrier()
        mine = time.perf_counter() - t0
        dt = R.max_over_ranks([mine])[0]
        per_rank = R.gather(1e3 * mine / args.steps)

        def dry_leg():     # the shape of a secondary leg: barrier, timed part, max over ranks; ILSX_BENCH_DRY_FAIL_RANK raises on one rank first
            if os.environ.get("ILSX_BENCH_DRY_FAIL_RANK") == str(rank):
                raise RuntimeError("dry-run leg failed on this rank")
            R.barrier()
            return dict(dt_max=R.max_over_ranks([0.001 * (1 + rank)])[0])
        leg = R.leg(dry_leg, 2 if world > 1 else 0)
        leg_errors = R.gather(1.0 if "error" in leg else 0.0)
        if rank == 0:
            print(json.dumps(dict(metric="dry-run", dry_run=True, n_gpus=world, steps=args.steps, warmup=args.warmup, leg=leg, leg_errors=leg_errors,
                                  ms_per_step=1e3 * dt / args.steps, value=world * args.steps / dt,
                                  scaling_detail=dict(per_rank_ms=per_rank, max_over_min=max(per_rank) / min(per_rank)))))
        R.close()
        return None

    import ctypes as C

    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    numa = pin_to_gpu_numa(local) if world > 1 or os.environ.get("ILSX_BENCH_PIN") else dict(pinned=False, reason="single rank")
    ctx = ia.Context(local, seed=1000 + rank)   # independent seed per replica
    lib = ctx.lib
    hid = [H, H]
    pol = ia.ReparamTanhMultivariateGaussianPolicy(hid, O, A, ctx=ctx, seed=10 + rank)
    q1 = ia.FlattenMlp(hid, 1, O + A, ctx=ctx, seed=20 + rank)
    q2 = ia.FlattenMlp(hid, 1, O + A, ctx=ctx, seed=30 + rank)
    tr = ia.SoftActorCritic(pol, q1, q2, max_batch=B, **SAC_KW)
    rb = ia.SimpleReplayBuffer(REPLAY_CAP, O, A, random_seed=rank, ctx=ctx)
    rng = np.random.default_rng(rank)
    chunk = 250_000
    for _ in range(REPLAY_CAP // chunk):   # pre-filled N = 1,000,000 rows (C2)
        rb.add_rows(*synth_rows(rng, chunk))

    from bench_rollout import Rollout
    ro = Rollout(ctx, pol, rb, N_ENV, seed=rank)

    tr.eval_statistics = {}  # no per-step stats readback inside the loop

    def step():
        ro.vec_step()                                    # 4096 env-steps
        tr.train_from_replay(rb, GRAD_PER_CALL, B)       # 1000 grad steps

    for _ in range(args.warmup):
        step()
    R.barrier(ctx)
    t0 = time.perf_counter()
    t_sample = t_train = 0.0
    for _ in range(args.steps):
        a0 = time.perf_counter()
        ro.vec_step()
        ctx.sync()
        a1 = time.perf_counter()
        tr.train_from_replay(rb, GRAD_PER_CALL, B)
        ctx.sync()
        a2 = time.perf_counter()
        t_sample += a1 - a0
        t_train += a2 - a1
    R.barrier(ctx)
    dt = time.perf_counter() - t0
    per_rank_ms = R.gather(1e3 * dt / args.steps)
    dt, t_sample, t_train = R.max_over_ranks([dt, t_sample, t_train])
    phase_timed = tr.phase_state()   # read HERE: the legs below put other contexts' kernels on this GPU, and a window that meets them falls back
    want_split = (world > 1 or os.environ.get("ILSX_BENCH_FORCE_DIST")) and not args.no_split_run

    # ---- roofline leg (rank 0): HIP events around every kernel launch (library instrumentation; graph bypassed).  BEFORE the legs below: they
    # put other contexts' kernels on this GPU, a train window that meets them is rolled back to one launch per stage and stays there, and the
    # kernel this leg names would no longer be the one the timed loop ran (round 6's first lines named k_mlp2_fwd_split for that reason)
    prof = {}
    if rank == 0:
        _lib.check(lib.ilsx_prof_reset(ctx.h))
        _lib.check(lib.ilsx_prof_enable(ctx.h, 1))
        tr.train_from_replay(rb, 200, B)
        _lib.check(lib.ilsx_prof_enable(ctx.h, 0))
        for kid in range(16):
            nl, ms = C.c_uint64(), C.c_double()
            _lib.check(lib.ilsx_prof_read(ctx.h, kid, C.byref(nl), C.byref(ms)))
            if nl.value:
                prof[kid] = (kernel_spelling(lib, ctx, kid), nl.value, ms.value)
    phase_roofline = tr.phase_state()

    # config 5's per-GPU shapes on EVERY rank (N > 1: 8 GPUs x grouped seeds — what north_star's "32 seeds x 1024 envs over 8 GPUs" means):
    # K = 8 grouped Hopper runs and 4 x 1024 grouped Humanoid runs, each rank on its own GPU, no collective in the data path
    legs = {}
    multi = world > 1 or bool(os.environ.get("ILSX_BENCH_FORCE_DIST"))   # the override runs the every-rank form of the legs on a one-rank group
    # each leg holds two one-word collectives (a barrier in front of its timed loop, the max of the ranks' times behind it): Ranks.leg keeps a
    # rank whose leg raised in step with the others
    if not args.no_seeds:
        legs["co_resident_seeds"] = R.leg(lambda: co_resident_seeds(R=R if multi else None), 2 if multi else 0)
    if multi and not args.no_aux:
        import bench_aux

        def humanoid_leg():
            hctx = ia.Context(local, seed=77 + rank)
            try:
                return bench_aux.bench_humanoid(hctx, R=R)
            finally:
                hctx.close()
        legs["humanoid_4x1024"] = R.leg(humanoid_leg, 2)

    result = None
    if rank == 0:
        grad_total = world * args.steps * GRAD_PER_CALL
        env_total = world * args.steps * N_ENV
        fl = flops_per_step()
        # a slot's algorithmic FLOPs per step: the merged phase kernels (one launch = three stages) carry their own count
        def slot_flops(k):
            return fl[{13: "k_sac_phase_a", 14: "k_sac_phase_c"}.get(k, k)]
        mlp = [k for k in prof if k in MLP_SLOTS]
        per_kernel = []
        for k in mlp:
            kname, knl, kms = prof[k]
            per_kernel.append(dict(kernel=kname, launches_per_step=knl / 200.0, avg_launch_us=1e3 * kms / knl, us_per_step=1e3 * kms / 200.0,
                                   algorithmic_flop_per_launch=slot_flops(k) / (knl / 200.0),
                                   frac=slot_flops(k) / (kms * 1e-3 / 200.0) / 1e12 / PEAK_F32_MFMA_TFLOPS))
        # dominant = the slot with the most time per step; slots within 3 % of it (the two phase launches take the same time to within
        # run-to-run noise) are ordered by the work they do, so the choice does not flip between runs
        tmax = max(prof[k][2] for k in mlp)
        dom = max((k for k in mlp if prof[k][2] >= 0.97 * tmax), key=slot_flops)
        name, nl, ms = prof[dom]
        launches_per_step = nl / 200.0
        flops_per_launch = slot_flops(dom) / launches_per_step
        avg_s = ms * 1e-3 / nl
        achieved = flops_per_launch / avg_s / 1e12
        roofline = dict(bound="mfma", kernel=name, achieved=achieved, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=achieved / PEAK_F32_MFMA_TFLOPS, traffic=pmc_traffic(dom),
                        traffic_source=os.path.relpath(PMC_SUMMARY, ROOT) if PMC_SUMMARY else None, traffic_stale=pmc_traffic_stale(), avg_launch_us=avg_s * 1e6,
                        algorithmic_flop_per_launch=flops_per_launch,
                        selection="most time per step; slots within 3 % of the maximum are tied and the one with more algorithmic work is named",
                        kernels=per_kernel,
                        kernel_ms_per_grad_step={prof[k][0]: prof[k][2] / 200.0 for k in prof})
        # ---- HBM-bound kernel: replay sample (4096 batches x 256 rows per launch)
        nb = 4096
        rec = C.c_int()
        _lib.check(lib.ilsx_replay_record_floats(rb.h, C.byref(rec)))
        out = ctx.empty((nb * B, rec.value))
        _lib.check(lib.ilsx_replay_sample_many(rb.h, nb, B, out.ptr))
        _lib.check(lib.ilsx_prof_reset(ctx.h))
        _lib.check(lib.ilsx_prof_enable(ctx.h, 1))
        for _ in range(10):
            _lib.check(lib.ilsx_replay_sample_many(rb.h, nb, B, out.ptr))
        _lib.check(lib.ilsx_prof_enable(ctx.h, 0))
        nl, ms = C.c_uint64(), C.c_double()
        _lib.check(lib.ilsx_prof_read(ctx.h, 6, C.byref(nl), C.byref(ms)))
        alg_bytes = 2.0 * nb * B * (2 * O + A + 2) * 4
        gbs = alg_bytes / (ms.value * 1e-3 / nl.value) / 1e9
        roofline_replay = dict(bound="hbm", kernel="k_replay_sample_many", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                               frac=gbs / PEAK_HBM_GBS, traffic=pmc_traffic(6), avg_launch_us=ms.value * 1e3 / nl.value,
                               algorithmic_bytes_per_launch=alg_bytes)
        # the same gather from a ring well past the 256 MB Infinity Cache (MI355X_MICROARCH.md): 4M records x 128 B = 512 MB, every
        # sampled row a fresh HBM line — the 1e6-row ring of the headline config (128 MB) is cache-resident after the first pass
        big = ia.SimpleReplayBuffer(4_000_000, O, A, random_seed=77, ctx=ctx)
        chunk_rows = synth_rows(rng, 500_000)
        for _ in range(8):
            big.add_rows(*chunk_rows)
        _lib.check(lib.ilsx_replay_sample_many(big.h, nb, B, out.ptr))
        _lib.check(lib.ilsx_prof_reset(ctx.h))
        _lib.check(lib.ilsx_prof_enable(ctx.h, 1))
        for _ in range(10):
            _lib.check(lib.ilsx_replay_sample_many(big.h, nb, B, out.ptr))
        _lib.check(lib.ilsx_prof_enable(ctx.h, 0))
        _lib.check(lib.ilsx_prof_read(ctx.h, 6, C.byref(nl), C.byref(ms)))
        gbs_big = alg_bytes / (ms.value * 1e-3 / nl.value) / 1e9
        roofline_replay["beyond_infinity_cache"] = dict(ring_bytes=4_000_000 * rec.value * 4, achieved=gbs_big, frac=gbs_big / PEAK_HBM_GBS,
                                                         avg_launch_us=ms.value * 1e3 / nl.value)
        result = dict(
            metric="env-steps/s + SAC grad-steps/s, Hopper-v2 4096 envs", value=grad_total / dt, unit="grad-steps/s",
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps,
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload="SAC Hopper-v2 dims (o=11,a=3), 4096 parallel envs, 256-256 MLP, batch 256, "
                                 "replay 1e6 rows; step = 1 vec-env step (4096 env-steps) + 1000 grad steps "
                                 "(sac_hopper.yaml:17-20)", env=ro.describe(), replicas=world,
                        parallelism=f"{world} independent replicas (seed sharding, no collective)"),
            env_steps_per_s=env_total / dt, env_steps_per_s_sample_phase=env_total / t_sample,
            grad_steps_per_s_train_phase=grad_total / t_train, roofline=roofline, roofline_replay=roofline_replay,
            # which step path the timed loop ran on: the merged phase kernels (4 launches per step) unless a window had to be rolled back
            # (another process's kernels on this GPU) and the agent fell back to one launch per stage (include/ilsx.h ilsx_sac_phase_state)
            phase_kernels=phase_timed, phase_kernels_roofline_leg=phase_roofline, phase_kernels_after_legs=tr.phase_state(),
            # per-rank step times behind the max-over-ranks `value` (the driver computes efficiency itself from its per-N runs); the
            # replica leg has NO data-path collective, so a slow rank is a placement / clock matter, not a communication one
            scaling_detail=dict(per_rank_ms=per_rank_ms, max_over_min=max(per_rank_ms) / min(per_rank_ms), numa_rank0=numa,
                                rccl_env={k: os.environ[k] for k in RCCL_ENV_KEYS if k in os.environ},
                                note="RCCL is used for the barrier + max-time of this leg and for the gradient all-reduce of `split_run` "
                                     "only; message sizes there are 0.55 MB (critics) and 0.28 MB (actor | alpha): latency-bound on "
                                     "xGMI, so NCCL_ALGO / NCCL_PROTO are left to RCCL's tuner unless set in the environment"))
        result.update(legs)
        if world == 1 and not args.no_aux:
            # BASELINE.json configs 3, 4, 5 (their single-GPU shapes) beside the headline, each with the roofline block of its dominant
            # kernel (live HIP-event timing, kernel named as rocprofv3 lists it); never `value`
            import bench_aux
            actx = ia.Context(local, seed=77)
            for key, fn in (("ppo_8192x128", bench_aux.bench_ppo), ("gail_walker", bench_aux.bench_gail),
                            ("humanoid_4x1024", bench_aux.bench_humanoid)):
                if key in legs:
                    continue     # already run in its every-rank form above
                try:
                    result[key] = fn(actx)
                except Exception as e:   # noqa: BLE001 — the headline line must come out whatever a secondary leg does
                    result[key] = dict(error=repr(e)[:300])
            actx.close()
            result["extra_keys"] = ["co_resident_seeds", "ppo_8192x128", "gail_walker", "humanoid_4x1024", "roofline_replay", "cpu_baseline"]
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(full=args.full_cpu_baseline)
            except Exception as e:   # noqa: BLE001 — the line must come out; an absent baseline is visible as such
                result["cpu_baseline"] = dict(error=repr(e)[:300])
    if want_split:
        # the split-run leg comes LAST and under a watchdog: it is the one part of this file that needs a working multi-rank RCCL
        # communicator, and the headline line must come out whatever happens to it (error -> recorded; no progress in 180 s -> rank 0
        # prints the line with the reason, every rank exits)
        import threading
        done = threading.Event()

        def dog():
            if not done.wait(180.0):
                if rank == 0:
                    result["split_run"] = dict(error="split-run leg made no progress in 180 s (RCCL communicator?)")
                    print(json.dumps(result), flush=True)
                os._exit(0)
        threading.Thread(target=dog, daemon=True).start()
        try:
            split = split_run_leg(R)
        except Exception as e:   # noqa: BLE001
            split = dict(error=repr(e)[:300])
        done.set()
        if rank == 0 and split is not None:
            result["split_run"] = split
    # The JSON line is the LAST thing on stdout.  RCCL's version banner (the GPU boxes export NCCL_DEBUG=VERSION) sits in every rank's C stdio
    # buffer until that process exits — behind a line rank 0 printed earlier; so every rank flushes its C streams, the ranks meet once more
    # (R.close: barrier + destroy), and only then rank 0 prints.  A final rendezvous that hangs or raises does not keep the line in: a timer emits it.
    import threading
    sys.stdout.flush()
    _flush_c_stdio()
    line = json.dumps(result) if rank == 0 else None
    emitted = threading.Event()

    def emit():
        if rank == 0 and not emitted.is_set():
            emitted.set()
            print(line, flush=True)
    if R.dist is not None:
        timer = threading.Timer(90.0, lambda: (emit(), os._exit(0)))
        timer.daemon = True
        timer.start()
        try:
            R.close()
        except Exception:   # noqa: BLE001 — the communicator is gone; the line still comes out
            pass
        timer.cancel()
        _flush_c_stdio()
    emit()
    ctx.close()
    return result


if __name__ == "__main__":
    main()
