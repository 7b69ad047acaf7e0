"""GPU suite (-m gpu): the path bench.py and every run script actually time — `train_from_replay` (rows drawn from the
replay ring INSIDE the step's first forward launch, Philox noise drawn inside the policy-finish prologue, the alpha /
counter tail deferred into the next step's first launch, the step replayed from a hipGraph) and the grouped
(co-resident seeds) form — checked against the oracle, not against itself.

How: `ilsx_sac_debug_batch` rebuilds, with the STANDALONE sample kernel (k_replay_sample, the one behind
`ilsx_replay_sample`) and a standalone Philox kernel, the batch and the two N(0,1) draws of gradient step k.  The oracle
(`oracle/sac_alpha.py`, pinned by the reference's own g4 vectors) is stepped on those inputs; the fused path must land on
the oracle's parameters (tolerances of test_hip_parity.py::test_sac_steps_vs_oracle), and the rows its first launch
gathered and published must be bit-identical to the standalone kernel's and to the host copy of the ring.
Reference: rlkit/torch/algorithms/sac/sac_alpha.py:78-181 + rlkit/data_management/simple_replay_buffer.py:239-253.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAC_KW = dict(reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, alpha_lr=3e-4, soft_target_tau=0.005,
              alpha=0.2, train_alpha=True, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9)
NETS = ("policy", "qf1", "qf2", "target_qf1", "target_qf2")


def _ring_data(rng, n, o, a):
    return (rng.normal(0, 1, (n, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (n, a))).astype(np.float32),
            rng.normal(0, 1, n).astype(np.float32), (rng.random(n) < 0.05).astype(np.uint8),
            rng.normal(0, 1, (n, o)).astype(np.float32))


def _agent(ia, ctx, o, a, hidden, params, kw, B):
    pol = ia.ReparamTanhMultivariateGaussianPolicy(hidden, o, a, ctx=ctx)
    q1, q2 = ia.FlattenMlp(hidden, 1, o + a, ctx=ctx), ia.FlattenMlp(hidden, 1, o + a, ctx=ctx)
    pol.set_flat_params(params[0]), q1.set_flat_params(params[1]), q2.set_flat_params(params[2])
    return ia.SoftActorCritic(pol, q1, q2, max_batch=B, **kw)


def _init(rng, o, a, hidden):
    from oracle import mlp as omlp
    return (omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2), omlp.init_mlp(rng, o + a, hidden, 1),
            omlp.init_mlp(rng, o + a, hidden, 1))


def _check_against_oracle(tr, orc, tag, res=None):
    if res is not None:   # gradients of the LAST step taken (the arena keeps them): SURVEY §8c's chained-step bound, all three networks
        for nm, key in (("qf1", "q1_grad"), ("qf2", "q2_grad"), ("policy", "pi_grad")):
            got, ref = tr.get_grads(nm), res[key]
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (tag, nm, np.abs(got - ref).max() / np.abs(ref).max())
    np.testing.assert_allclose(tr.log_alpha, orc.log_alpha[0], rtol=0, atol=1e-6, err_msg=tag)
    for nm, ov in (("policy", orc.pi), ("qf1", orc.q1), ("qf2", orc.q2), ("target_qf1", orc.tq1), ("target_qf2", orc.tq2)):
        np.testing.assert_allclose(tr.get_params(nm), ov, rtol=0, atol=5e-5, err_msg=f"{nm} {tag}")


def _check_stats(st, res, tag):
    for k_ref, k_or in (("QF1 Loss", "qf1_loss"), ("QF2 Loss", "qf2_loss"), ("Policy Loss", "policy_loss"), ("Alpha Loss", "alpha_loss")):
        np.testing.assert_allclose(st[k_ref], res[k_or], rtol=2e-4, atol=2e-6, err_msg=f"{k_ref} {tag}")
    # means of O(1e-2) quantities after several chained optimiser steps: the parameters themselves carry 5e-5 (below)
    np.testing.assert_allclose(st["Q1 Predictions Mean"], res["q1_pred"].mean(), rtol=1e-4, atol=2e-5, err_msg=tag)
    np.testing.assert_allclose(st["Log Pis Mean"], res["log_pi"].mean(), rtol=1e-4, atol=2e-5, err_msg=tag)


def _relu_margin(orc, batch):
    """Smallest |pre-activation| of the two critics on this batch (float64): a relu unit within fp32 summation noise (~1e-7) of
    its kink is gated differently by two correct fp32 implementations, and Adam turns that one gate into +-lr on a whole row of
    weights.  Seen for real: seed 180 of the (17, 6, 128, 37) case has |z| = 2.3e-8 in Q1's second layer at step 0."""
    from oracle import mlp as omlp
    x = np.concatenate([batch["observations"], batch["actions"]], 1).astype(np.float64)
    m = np.inf
    for flat in (orc.q1, orc.q2):
        lay = omlp.unpack(flat.astype(np.float64), orc.o + orc.a, orc.hidden, 1)
        h = x
        for W, b in lay[:-1]:
            z = h @ W.T + b
            m = min(m, np.abs(z).min())
            h = np.maximum(z, 0)
    return m


@pytest.mark.parametrize("o,a,H,B,n_steps", [(11, 3, 256, 256, 6), (17, 6, 128, 37, 5), (376, 17, 256, 64, 5), (11, 3, 256, 1, 4), (11, 3, 256, 16, 4)])
def test_fused_train_from_replay_matches_oracle(o, a, H, B, n_steps):
    """ONE train_from_replay(rb, n, B) call (graph + deferred tail ON) == n oracle steps on the rebuilt inputs."""
    import ilswiss_amd as ia
    from oracle.sac_alpha import SacAlphaOracle
    hidden, N = [H, H], 5000
    kw = dict(SAC_KW, target_entropy=-4.0) if a > 6 else SAC_KW
    for seed in range(o + H + B, o + H + B + 8):   # first seed whose comparison is well-posed (see _relu_margin)
        rng = np.random.default_rng(seed)
        params = _init(rng, o, a, hidden)
        data = _ring_data(rng, N, o, a)
        ctx = ia.Context(0, seed=4321)
        rb = ia.SimpleReplayBuffer(8192, o, a, random_seed=9, ctx=ctx)
        rb.add_rows(*data)
        tr = _agent(ia, ctx, o, a, hidden, params, kw, B)
        orc = SacAlphaOracle(o, a, hidden, *params, **kw)
        assert tr.rng_step == 0
        # inputs of steps 0..n-1, rebuilt BEFORE the run by the independent kernels
        inputs = [tr.debug_batch(rb, k, B) for k in range(n_steps)]
        margin = np.inf
        res_first = None
        for batch, e1, e2, _ in inputs:
            margin = min(margin, _relu_margin(orc, batch))
            res = orc.train_step(batch, e1, e2)
            res_first = res_first or res
        # a flipped gate is one batch row's share of a unit's gradient: 1/B of it.  It only breaks the 5e-5 bound for small
        # batches (a unit that is live on that one row alone gets +-lr instead of 0); at B = 256 there are ~1.5M pre-activations
        # per test and |z| < 1e-6 somewhere is the norm, without effect at this tolerance
        if margin > 2e-6 or B >= 128:
            break
        ctx.close()
    else:
        pytest.fail("no well-posed seed")
    for k, (batch, e1, e2, idx) in enumerate(inputs):
        assert idx.min() >= 0 and idx.max() < N
        np.testing.assert_array_equal(batch["observations"], data[0][idx])          # gather == host copy of the ring
        np.testing.assert_array_equal(batch["actions"], data[1][idx])
        np.testing.assert_array_equal(batch["rewards"][:, 0], data[2][idx])
        np.testing.assert_array_equal(batch["terminals"][:, 0], data[3][idx].astype(np.float32))
        np.testing.assert_array_equal(batch["next_observations"], data[4][idx])
        assert not np.array_equal(e1, e2)
        if e1.size >= 100:   # a sanity check of the noise, meaningless on the single-row edge case
            assert abs(e1.mean()) < 0.2 and abs(e1.std() - 1.0) < 0.2
        if k:
            assert not np.array_equal(idx, inputs[k - 1][3])
    tr.eval_statistics = None   # statistics of the FIRST step of the call (sac_alpha.py:185-190: the first train_step after end_epoch)
    tr.train_from_replay(rb, n_steps, B)
    assert tr.rng_step == n_steps
    # the rows the fused gather published for the last step are the standalone kernel's rows, bit for bit
    last, eps_cur_used = tr.debug_last_batch(B)
    for key in ("observations", "actions", "rewards", "terminals", "next_observations"):
        np.testing.assert_array_equal(last[key], inputs[-1][0][key], err_msg=key)
    np.testing.assert_array_equal(eps_cur_used, inputs[-1][2])
    _check_stats(tr.get_eval_statistics(), res_first, "first step of the call")
    _check_against_oracle(tr, orc, f"after {n_steps} fused steps", res)
    ctx.close()


def test_fused_calls_chain_and_index_stream_is_the_sample_kernels():
    """Several calls (1 + 3 + 1 steps; the pending tail is flushed and re-armed at every call boundary) against the oracle step by
    step; and the fused draw at counter k is the draw ilsx_replay_sample makes at its counter k."""
    import ilswiss_amd as ia
    from ilswiss_amd import _lib
    from oracle.sac_alpha import SacAlphaOracle
    o, a, H, B, N = 11, 3, 256, 256, 3000
    rng = np.random.default_rng(77)
    hidden = [H, H]
    params = _init(rng, o, a, hidden)
    data = _ring_data(rng, N, o, a)
    ctx = ia.Context(0, seed=11)
    rb = ia.SimpleReplayBuffer(4096, o, a, random_seed=3, ctx=ctx)
    rb.add_rows(*data)
    tr = _agent(ia, ctx, o, a, hidden, params, SAC_KW, B)
    orc = SacAlphaOracle(o, a, hidden, *params, **SAC_KW)
    k = 0
    for n in (1, 3, 1):
        inputs = [tr.debug_batch(rb, k + i, B) for i in range(n)]
        tr.eval_statistics = None
        tr.train_from_replay(rb, n, B)
        res_first = None
        for batch, e1, e2, _ in inputs:
            res = orc.train_step(batch, e1, e2)
            res_first = res_first or res
        _check_stats(tr.get_eval_statistics(), res_first, f"step {k} (the first of a call of {n})")
        k += n
        _check_against_oracle(tr, orc, f"after {k} steps")
    # ilsx_replay_sample's own counter starts at 1: its k-th call draws what fused step k draws
    bufs = [ctx.empty((B, o)), ctx.empty((B, a)), ctx.empty((B,)), ctx.empty((B,)), ctx.empty((B, o)), ctx.empty((B,), np.int64)]
    for call in (1, 2, 3):
        _lib.check(ctx.lib.ilsx_replay_sample(rb.h, B, None, *[b.ptr for b in bufs]))
        np.testing.assert_array_equal(bufs[5].numpy(), tr.debug_batch(rb, call, B)[3])
    ctx.close()


def test_grouped_lockstep_matches_oracle():
    """K = 3 co-resident seeds through ilsx_sac_group (one launch per stage, deferred tail, graph): every agent lands on ITS oracle."""
    import ilswiss_amd as ia
    from oracle.sac_alpha import SacAlphaOracle
    o, a, H, B, K, n_steps, N = 11, 3, 256, 256, 3, 5, 4000
    hidden = [H, H]
    ctx = ia.Context(0, seed=2024)
    agents, rbs, orcs = [], [], []
    for k in range(K):
        rng = np.random.default_rng(100 + k)
        params = _init(rng, o, a, hidden)
        rb = ia.SimpleReplayBuffer(4096, o, a, random_seed=k, ctx=ctx)
        rb.add_rows(*_ring_data(rng, N, o, a))
        agents.append(_agent(ia, ctx, o, a, hidden, params, SAC_KW, B))
        rbs.append(rb)
        orcs.append(SacAlphaOracle(o, a, hidden, *params, **SAC_KW))
    inputs = [[agents[k].debug_batch(rbs[k], s, B) for s in range(n_steps)] for k in range(K)]
    assert not np.array_equal(inputs[0][0][3], inputs[1][0][3])   # the seeds draw different rows
    grp = ia.SoftActorCriticGroup(agents)
    for tr in agents:
        tr.eval_statistics = {}
    grp.train_from_replay(rbs, n_steps - 1, B)
    for tr in agents:
        tr.eval_statistics = None
    grp.train_from_replay(rbs, 1, B)
    for k in range(K):
        for batch, e1, e2, _ in inputs[k]:
            res = orcs[k].train_step(batch, e1, e2)
        last, _ = agents[k].debug_last_batch(B)
        np.testing.assert_array_equal(last["observations"], inputs[k][-1][0]["observations"])
        _check_stats(agents[k].get_eval_statistics(), res, f"agent {k}")
        _check_against_oracle(agents[k], orcs[k], f"agent {k}")
    grp.close()
    ctx.close()


_SPLIT_SCRIPT = r'''
import ctypes as C, hashlib, json, os, sys
import numpy as np
sys.path.insert(0, ".")
import ilswiss_amd as ia
from ilswiss_amd import _lib
o, a, hid, B, N = 11, 3, [256, 256], 256, 5000
rng = np.random.default_rng(5)
ctx = ia.Context(0, seed=77)
if os.environ.get("ILSX_SPLIT_FORCE"):
    ident = (C.c_uint8 * 128)()
    _lib.check(ctx.lib.ilsx_comm_unique_id(ident))
    _lib.check(ctx.lib.ilsx_comm_init(ctx.h, ident, 1, 0))
    n, r = C.c_int(), C.c_int()
    _lib.check(ctx.lib.ilsx_comm_info(ctx.h, C.byref(n), C.byref(r)))
    assert (n.value, r.value) == (1, 0)
    buf = ctx.from_numpy(np.arange(1000, dtype=np.float32))
    _lib.check(ctx.lib.ilsx_comm_allreduce_sum(ctx.h, buf.ptr, 1000))
    assert np.array_equal(buf.numpy(), np.arange(1000, dtype=np.float32))
rb = ia.SimpleReplayBuffer(8192, o, a, random_seed=3, ctx=ctx)
rb.add_rows(rng.normal(0, 1, (N, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (N, a))).astype(np.float32),
            rng.normal(0, 1, N).astype(np.float32), rng.random(N) < 0.01, rng.normal(0, 1, (N, o)).astype(np.float32))
pol = ia.ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=10)
q1, q2 = ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=20), ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=30)
tr = ia.SoftActorCritic(pol, q1, q2, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rb, 6, B)
tr.eval_statistics = None
tr.train_from_replay(rb, 1, B)
h = hashlib.sha256()
for name in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
    h.update(np.ascontiguousarray(tr.get_params(name)).tobytes())
import sys
print(json.dumps(dict(params=h.hexdigest(), log_alpha=repr(tr.log_alpha), qf1=repr(float(tr.eval_statistics["QF1 Loss"])))))
print("PHASE", int(tr.phase_state()["last_window_on_phase"]), file=sys.stderr)
ctx.close()
'''


def test_split_run_phases_with_rccl_on_the_ctx_stream_equal_the_fused_step():
    """The split-run code path — un-fused phases with ncclAllReduce (librccl dlopen'ed by libilsx, enqueued on the ctx's own
    stream, no host sync) between backward and update — on a ONE-rank communicator (all a single-GPU box allows) gives bit for
    bit the parameters of the fused single-run step, with and without replaying the step from a hipGraph."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    # split: the default form — every launch direct; split_segments: three captured segments with the two all-reduces enqueued between them;
    # split_graph: the whole step, collective included, in one capture
    # (round 5: the default split step runs on the merged phase kernels — A, dW, all-reduce, Adam, C, dW, all-reduce, Adam, the tail deferred
    #  and fed the all-reduced alpha gradient; split_nophase keeps one launch per stage with k_sac_stats / k_sac_finish)
    on_phase = {}
    for tag, extra in (("fused", {}), ("split", {"ILSX_SPLIT_FORCE": "1"}), ("split_nophase", {"ILSX_SPLIT_FORCE": "1", "ILSX_SPLIT_NO_PHASE": "1"}),
                       ("split_segments", {"ILSX_SPLIT_FORCE": "1", "ILSX_SPLIT_SEGMENTS": "1"}),
                       ("split_graph", {"ILSX_SPLIT_FORCE": "1", "ILSX_SPLIT_GRAPH": "1"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _SPLIT_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-3000:])
        outs[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])   # RCCL prints its own banner lines
        on_phase[tag] = [l for l in r.stderr.splitlines() if l.startswith("PHASE")][-1] == "PHASE 1"
    assert on_phase["fused"] and on_phase["split"] and on_phase["split_graph"] and not on_phase["split_nophase"] and not on_phase["split_segments"], on_phase
    assert outs["split"] == outs["fused"], outs
    assert outs["split_nophase"] == outs["fused"], outs
    assert outs["split_segments"] == outs["fused"], outs
    assert outs["split_graph"] == outs["fused"], outs


@pytest.mark.parametrize("o,a,H,B,n_steps", [(11, 3, 256, 256, 7), (17, 6, 128, 37, 5), (11, 3, 256, 100, 4), (11, 3, 256, 1, 3)])
def test_phase_kernels_are_bitwise_the_eight_launch_path(o, a, H, B, n_steps):
    """ilsx_sac_train_from_replay runs F1 F2 B1 and F3 B2 B3 as ONE launch each (k_sac_phase_a / _c: the stages hand over through
    per-tile counters in the XCD's L2).  Same stage bodies, same summation order: every parameter, target, optimiser moment and log_alpha
    must equal the one-launch-per-stage path (ILSX_NO_PHASE=1) bit for bit — through several calls, a ragged last tile (B = 37, 100)
    and the 128-wide / 2-slice instantiation."""
    import os
    import ilswiss_amd as ia
    hidden, N = [H, H], 6000
    rng = np.random.default_rng(o + H + B)
    params = _init(rng, o, a, hidden)
    data = _ring_data(rng, N, o, a)
    runs = []
    for no_phase in (False, True):
        if no_phase:
            os.environ["ILSX_NO_PHASE"] = "1"
        else:
            os.environ.pop("ILSX_NO_PHASE", None)
        try:
            ctx = ia.Context(0, seed=2024)
            rb = ia.SimpleReplayBuffer(8192, o, a, random_seed=5, ctx=ctx)
            rb.add_rows(*data)
            tr = _agent(ia, ctx, o, a, hidden, params, SAC_KW, B)
            tr.eval_statistics = {}
            tr.train_from_replay(rb, n_steps, B)
            tr.train_from_replay(rb, 1, B)             # call boundary: pending tail flushed and re-armed
            tr.eval_statistics = None
            tr.train_from_replay(rb, 2, B)             # statistics of the last step
            st = dict(tr.get_eval_statistics())
            snap = tr.get_snapshot()
            runs.append((snap, st, tr.rng_step))
            ctx.close()
        finally:
            os.environ.pop("ILSX_NO_PHASE", None)
    (s0, st0, r0), (s1, st1, r1) = runs
    assert r0 == r1 == n_steps + 3
    for k in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
        np.testing.assert_array_equal(s0[k], s1[k], err_msg=k)
    assert s0["log_alpha"] == s1["log_alpha"]
    for k in ("policy_optimizer", "qf1_optimizer", "qf2_optimizer"):
        np.testing.assert_array_equal(s0[k]["exp_avg"], s1[k]["exp_avg"], err_msg=k)
        np.testing.assert_array_equal(s0[k]["exp_avg_sq"], s1[k]["exp_avg_sq"], err_msg=k)
    for k, v in st0.items():
        assert v == st1[k] or (np.isnan(v) and np.isnan(st1[k])), (k, v, st1[k])


def test_broken_phase_window_is_rolled_back_and_rerun():
    """A window whose phase kernels report a broken hand-off (another process's kernels on the GPU: the reference's launcher runs every
    worker of a sweep on one GPU, run_experiment.py:57-78) must not cost the run: the library checkpoints parameters / optimiser state /
    counters at the start of the window, rolls back and re-runs the same steps on one launch per stage.  ilsx_sac_debug_break_phase makes
    the NEXT window find a time-out mark; since the two paths are bit-identical, the agent must end exactly where an undisturbed one does
    — statistics of the re-run included — and report fallbacks = 1, disabled."""
    import ilswiss_amd as ia
    o, a, H, B = 11, 3, 256, 256
    hidden, N = [H, H], 6000
    rng = np.random.default_rng(99)
    params = _init(rng, o, a, hidden)
    data = _ring_data(rng, N, o, a)
    outs = []
    for broken in (False, True):
        ctx = ia.Context(0, seed=31)
        rb = ia.SimpleReplayBuffer(8192, o, a, random_seed=5, ctx=ctx)
        rb.add_rows(*data)
        tr = _agent(ia, ctx, o, a, hidden, params, SAC_KW, B)
        tr.eval_statistics = {}
        tr.train_from_replay(rb, 5, B)
        st0 = tr.phase_state()
        assert st0["last_window_on_phase"] and not st0["disabled"] and st0["fallbacks"] == 0, st0   # B = 256 / H = 256: the benchmarked shape uses the phase kernels
        assert st0["wgs_per_cu_a"] >= 2 and st0["wgs_per_cu_c"] >= 1, st0   # 16 tiles x 4 tasks x 4 slices + the tail workgroup = 257 resident workgroups
        if broken:
            _check(ctx, ctx.lib.ilsx_sac_debug_break_phase(tr.h))
        tr.eval_statistics = None
        tr.train_from_replay(rb, 4, B)     # broken: rolled back at its end, re-run on the 8-launch path, statistics from the re-run
        st = dict(tr.get_eval_statistics())
        tr.eval_statistics = {}
        tr.train_from_replay(rb, 3, B)     # and the agent goes on (on one launch per stage)
        ps = tr.phase_state()
        assert ps["fallbacks"] == (1 if broken else 0) and ps["disabled"] == broken and ps["last_window_on_phase"] == (not broken), ps
        outs.append((tr.get_snapshot(), st, tr.rng_step))
        ctx.close()
    (s0, st0, r0), (s1, st1, r1) = outs
    assert r0 == r1 == 12
    for k in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
        np.testing.assert_array_equal(s0[k], s1[k], err_msg=k)
    assert s0["log_alpha"] == s1["log_alpha"]
    for k in ("policy_optimizer", "qf1_optimizer", "qf2_optimizer"):
        np.testing.assert_array_equal(s0[k]["exp_avg"], s1[k]["exp_avg"], err_msg=k)
        np.testing.assert_array_equal(s0[k]["exp_avg_sq"], s1[k]["exp_avg_sq"], err_msg=k)
    for k, v in st0.items():
        assert v == st1[k] or (np.isnan(v) and np.isnan(st1[k])), (k, v, st1[k])


def _check(ctx, rc):
    from ilswiss_amd import _lib
    _lib.check(rc)


_SHARED_GPU_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, ".")
import ilswiss_amd as ia
from oracle import mlp as omlp
seed = int(sys.argv[1])
o, a, hidden, B, N = 11, 3, [256, 256], 256, 4000
rng = np.random.default_rng(seed)
params = (omlp.init_mlp(rng, o, hidden, a, init_w=1e-3, n_heads=2), omlp.init_mlp(rng, o + a, hidden, 1), omlp.init_mlp(rng, o + a, hidden, 1))
data = (rng.normal(0, 1, (N, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (N, a))).astype(np.float32),
        rng.normal(0, 1, N).astype(np.float32), (rng.random(N) < 0.05).astype(np.uint8), rng.normal(0, 1, (N, o)).astype(np.float32))
ctx = ia.Context(0, seed=1000 + seed)
rb = ia.SimpleReplayBuffer(8192, o, a, random_seed=seed, ctx=ctx)
rb.add_rows(*data)
pol = ia.ReparamTanhMultivariateGaussianPolicy(hidden, o, a, ctx=ctx)
q1, q2 = ia.FlattenMlp(hidden, 1, o + a, ctx=ctx), ia.FlattenMlp(hidden, 1, o + a, ctx=ctx)
pol.set_flat_params(params[0]), q1.set_flat_params(params[1]), q2.set_flat_params(params[2])
tr = ia.SoftActorCritic(pol, q1, q2, max_batch=B, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005)
tr.eval_statistics = {}
for _ in range(int(sys.argv[2])):
    tr.train_from_replay(rb, 400, B)
ctx.sync()
h = hashlib.sha256()
for k in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
    h.update(np.ascontiguousarray(tr.get_params(k)).tobytes())
print(json.dumps(dict(digest=h.hexdigest(), log_alpha=tr.log_alpha, steps=tr.rng_step, phase=tr.phase_state())))
"""


def test_two_default_processes_share_one_gpu():
    """The reference's sweep mode (run_experiment.py:57-78: `num_workers` children on the SAME GPU).  Two processes with the DEFAULT
    environment — phase kernels on — train side by side on one GPU; whatever the phase kernels' hand-offs ran into, each process must
    finish and land bit-exactly where the same run lands alone on the one-launch-per-stage path (ILSX_NO_PHASE=1), because a disturbed
    window is rolled back and re-run and the two paths are bit-identical."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "ILSX_NO_PHASE"}
    ref = {}
    for seed in (1, 2):
        out = subprocess.run([sys.executable, "-c", _SHARED_GPU_SCRIPT, str(seed), "6"], env=dict(env, ILSX_NO_PHASE="1"), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        ref[seed] = json.loads(out.stdout.strip().splitlines()[-1])
    procs = {seed: subprocess.Popen([sys.executable, "-c", _SHARED_GPU_SCRIPT, str(seed), "6"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for seed in (1, 2)}
    for seed, p in procs.items():
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-2000:]
        got = json.loads(so.strip().splitlines()[-1])
        assert got["steps"] == ref[seed]["steps"] == 2400
        assert got["digest"] == ref[seed]["digest"] and got["log_alpha"] == ref[seed]["log_alpha"], (seed, got, ref[seed])


def test_dw_tile_shapes_are_bitwise():
    """k_mlp_bwd_dw<GRP, NH, KT>: the output tile of a workgroup (32 x 64, 16 x 32, 16 x 16) decides how many workgroups a launch has,
    not what a wave computes or in which order an output element is summed — parameters, targets and Adam moments after several fused
    steps must not depend on it (launch_bwd_dw picks 16 x 16 for the SAC step; ILSX_DW_TILE forces a shape)."""
    import os
    import ilswiss_amd as ia
    o, a, H, B = 11, 3, 256, 100
    hidden, N = [H, H], 6000
    rng = np.random.default_rng(77)
    params = _init(rng, o, a, hidden)
    data = _ring_data(rng, N, o, a)
    snaps = []
    for shape in (None, "24", "12"):
        if shape:
            os.environ["ILSX_DW_TILE"] = shape
        try:
            ctx = ia.Context(0, seed=11)
            rb = ia.SimpleReplayBuffer(8192, o, a, random_seed=5, ctx=ctx)
            rb.add_rows(*data)
            tr = _agent(ia, ctx, o, a, hidden, params, SAC_KW, B)
            tr.eval_statistics = {}
            tr.train_from_replay(rb, 6, B)
            snaps.append(tr.get_snapshot())
            ctx.close()
        finally:
            os.environ.pop("ILSX_DW_TILE", None)
    for s in snaps[1:]:
        for k in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
            np.testing.assert_array_equal(snaps[0][k], s[k], err_msg=k)
        for k in ("policy_optimizer", "qf1_optimizer", "qf2_optimizer"):
            np.testing.assert_array_equal(snaps[0][k]["exp_avg"], s[k]["exp_avg"], err_msg=k)
            np.testing.assert_array_equal(snaps[0][k]["exp_avg_sq"], s[k]["exp_avg_sq"], err_msg=k)



_SPLIT_PPO_DISC_SCRIPT = r'''
import ctypes as C, hashlib, json, os, sys
import numpy as np
sys.path.insert(0, ".")
import ilswiss_amd as ia
from ilswiss_amd import _lib
from ilswiss_amd.adv_irl import MLPDisc
from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
G = int(os.environ.get("TEST_GRAD_WORLD", "1"))
np.random.seed(12)      # MLPDisc initialises from numpy's global generator, like torch's modules from torch's
ctx = ia.Context(0, seed=77)
if os.environ.get("ILSX_SPLIT_FORCE"):
    ident = (C.c_uint8 * 128)()
    _lib.check(ctx.lib.ilsx_comm_unique_id(ident))
    _lib.check(ctx.lib.ilsx_comm_init(ctx.h, ident, 1, 0))
out = {}
sha = lambda *xs: hashlib.sha256(b"".join(np.ascontiguousarray(x).tobytes() for x in xs)).hexdigest()
# ---- PPO: 3 epochs of ragged minibatches over five trajectories, both log-std forms
rng = np.random.default_rng(99)
o, a, hid = 17, 6, [128, 128]
trajs = [dict(observations=rng.normal(0, 1, (L, o)).astype(np.float32), actions=rng.normal(0, 0.7, (L, a)).astype(np.float32),
              rewards=rng.normal(0.5, 1.0, (L, 1)).astype(np.float32)) for L in (2, 31, 100, 64, 9)]
N = sum(t["rewards"].shape[0] for t in trajs)
perms = np.stack([rng.permutation(N) for _ in range(3)])
for cond in (False, True):
    vf = ia.FlattenMlp(hid, 1, o, hidden_activation="tanh", ctx=ctx, seed=3)
    pol = ReparamMultivariateGaussianPolicy(hid, o, a, conditioned_std=cond, hidden_activation="tanh", ctx=ctx, seed=4)
    tr = PPO(pol, vf, mini_batch_size=48, update_epoch=3, gae_tau=0.9, value_l2_reg=1e-3, use_value_clip=True, max_samples=512, grad_world=G)
    tr.train_step(trajs, perms)
    gn = C.c_float()
    _lib.check(ctx.lib.ilsx_ppo_debug_grad_norm(tr.h, C.byref(gn)))
    out[f"ppo_cond{int(cond)}"] = sha(tr.get_flat_params(0), tr.get_flat_params(1))
    # one more single-minibatch update from a fresh trainer: the norm clip_grad_norm_ saw on the FIRST policy minibatch (parameters still equal)
    vf2 = ia.FlattenMlp(hid, 1, o, hidden_activation="tanh", ctx=ctx, seed=3)
    pol2 = ReparamMultivariateGaussianPolicy(hid, o, a, conditioned_std=cond, hidden_activation="tanh", ctx=ctx, seed=4)
    tr2 = PPO(pol2, vf2, mini_batch_size=N, update_epoch=1, gae_tau=0.9, value_l2_reg=0.0, max_samples=512, grad_world=G)
    tr2.train_step(trajs, perms[:1])
    _lib.check(ctx.lib.ilsx_ppo_debug_grad_norm(tr2.h, C.byref(gn)))
    out[f"ppo_norm_cond{int(cond)}"] = float(gn.value)
# ---- discriminator: fused two-block kernel, any-depth chain, without penalty
D = o + a
for tag, blocks, Hd, B, act, gp in (("d2", 2, 128, 64, "tanh", True), ("d3", 3, 64, 48, "relu", True), ("d2nogp", 2, 128, 37, "tanh", False)):
    rng = np.random.default_rng(7 + blocks)
    disc = MLPDisc(D, num_layer_blocks=blocks, hid_dim=Hd, hid_act=act, use_bn=False, ctx=ctx)
    disc.bind(o, max_batch=B, disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=gp, grad_pen_weight=8.0, grad_world=G)
    grads = []
    for s in range(3):
        xe, xp = rng.normal(0, 1, (B, D)).astype(np.float32), rng.normal(0.2, 1.3, (B, D)).astype(np.float32)
        disc.train_step(xe[:, :o], xe[:, o:], xp[:, :o], xp[:, o:], eps=rng.random((B, 1)).astype(np.float32))
        if s == 0:
            grads = disc.get_flat_grads()
    out[tag] = sha(disc.get_flat_params())
    out[tag + "_g0"] = sha(grads * np.float32(G))      # the first step's gradient (parameters still equal): this rank's share x G
print(json.dumps(out))
ctx.close()
'''


def test_ppo_and_discriminator_split_paths_on_a_one_rank_communicator():
    """SURVEY section 8e "PPO split: same, per-minibatch grads.  Disc: same" on the device (VERDICT r5 missing 6).  (1) The split code path — weight
    gradients without the fused optimiser, ncclAllReduce of the arena on the ctx stream, Adam (+ L2, + norm clip) as launches of their own —
    on a ONE-rank communicator ends bit for bit where the fused single-rank step ends: PPO with both log-std forms (the action_log_std
    gradient rides in the policy arena), the two-block discriminator kernel, the any-depth chain, with and without gradient penalty.
    (2) grad_world = 2 on the same one-rank communicator: every mean-loss gradient is exactly half the single-rank one (a power-of-two
    scale is exact through every product and sum) — the discriminator's first-step arena, and the norm PPO's clip sees.  The two-rank sum
    itself is tests/test_parallel_gloo.py's (numpy engine)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, extra in (("fused", {}), ("split1", {"ILSX_SPLIT_FORCE": "1"}), ("split2", {"ILSX_SPLIT_FORCE": "1", "TEST_GRAD_WORLD": "2"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _SPLIT_PPO_DISC_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-3000:])
        outs[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert outs["split1"] == outs["fused"], (outs["split1"], outs["fused"])
    for k in ("d2_g0", "d3_g0", "d2nogp_g0"):
        assert outs["split2"][k] == outs["fused"][k], k
    for k in ("ppo_norm_cond0", "ppo_norm_cond1"):
        assert outs["split2"][k] == 0.5 * outs["fused"][k] and outs["fused"][k] > 0, (k, outs["split2"][k], outs["fused"][k])
