"""SAC-alpha gradient-step time of one run at given observation / action widths:  python tools/sac_dims_rate.py <obs_dim> <act_dim>  (ILSX_NO_SPLIT=1: generic kernels instead of the column-split ones)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import ilswiss_amd as ia
o, a, H, B, CAP = int(sys.argv[1]), int(sys.argv[2]), 256, 256, 50000
ctx = ia.Context(0, seed=0)
rng = np.random.default_rng(0)
rb = ia.SimpleReplayBuffer(CAP, o, a, ctx=ctx)
rb.add_rows(rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32), rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=ctx, seed=1), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=2), ia.FlattenMlp([H, H], 1, o + a, ctx=ctx, seed=3), policy_lr=3e-4, qf_lr=3e-4, max_batch=B)
tr.eval_statistics = {}
tr.train_from_replay(rb, 300, B); ctx.sync()
t0 = time.perf_counter(); tr.train_from_replay(rb, 2000, B); ctx.sync()
print(sys.argv[1:], "us/step", 1e6 * (time.perf_counter() - t0) / 2000)
# per-kernel-class averages (library instrumentation: dispatch-stamped events, direct launches)
import ctypes as C, os
from ilswiss_amd import _lib
if os.environ.get("ILSX_NO_GRAPH"):
    lib = ctx.lib
    _lib.check(lib.ilsx_prof_reset(ctx.h)); _lib.check(lib.ilsx_prof_enable(ctx.h, 1))
    tr.train_from_replay(rb, 200, B); ctx.sync()
    _lib.check(lib.ilsx_prof_enable(ctx.h, 0))
    for kid, name in ((0, "mlp_fwd"), (1, "mlp_bwd_dx"), (13, "sac_phase_a"), (14, "sac_phase_c"), (2, "mlp_bwd_dw"), (8, "sac_finish")):
        nl, ms = C.c_uint64(), C.c_double()
        _lib.check(lib.ilsx_prof_read(ctx.h, kid, C.byref(nl), C.byref(ms)))
        if nl.value:
            print(f"   {name}: {nl.value / 200:.1f} launches/step, {1e3 * ms.value / nl.value:.2f} us each, {1e3 * ms.value / 200:.1f} us/step")
