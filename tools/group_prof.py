"""Per-kernel HIP-event times of the grouped SAC step (ilsx_sac_group) for K co-resident agents."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
from ilswiss_amd import _lib
from ilswiss_amd.replay import SimpleReplayBuffer
o, a, H, B, CAP = 11, 3, 256, 256, 100_000
rng = np.random.default_rng(0)
data = (rng.normal(0, 1, (CAP, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (CAP, a))).astype(np.float32),
        rng.normal(0, 1, CAP).astype(np.float32), rng.random(CAP) < 1e-3, rng.normal(0, 1, (CAP, o)).astype(np.float32))
for K in [int(x) for x in sys.argv[1:]] or [1, 4, 16]:
    c = ia.Context(0, seed=7)
    rbs, trs = [], []
    for k in range(K):
        rb = SimpleReplayBuffer(CAP, o, a, random_seed=k, ctx=c); rb.add_rows(*data)
        tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy([H, H], o, a, ctx=c, seed=k), ia.FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 1),
                                ia.FlattenMlp([H, H], 1, o + a, ctx=c, seed=k + 2), policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
        tr.eval_statistics = {}
        rbs.append(rb); trs.append(tr)
    grp = ia.SoftActorCriticGroup(trs)
    grp.train_from_replay(rbs, 50, B); c.sync()
    _lib.check(c.lib.ilsx_prof_reset(c.h)); _lib.check(c.lib.ilsx_prof_enable(c.h, 1))
    n = 200
    grp.train_from_replay(rbs, n, B)
    _lib.check(c.lib.ilsx_prof_enable(c.h, 0))
    out = []
    for kid in range(16):
        nl, ms = C.c_uint64(), C.c_double()
        _lib.check(c.lib.ilsx_prof_read(c.h, kid, C.byref(nl), C.byref(ms)))
        if nl.value:
            out.append(f"{c.lib.ilsx_kernel_name(kid).decode()} {nl.value // n}x{1e3 * ms.value / nl.value:.1f}us")
    print(f"K={K}: " + "  ".join(out), flush=True)
    grp.close(); c.close()
