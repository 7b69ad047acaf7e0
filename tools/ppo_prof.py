"""Per-kernel HIP-event times of one PPO train step at BASELINE config-4 sizes (library instrumentation)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
from ilswiss_amd import _lib
from ilswiss_amd.networks import FlattenMlp
from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
ctx = ia.Context(0, seed=0)
o, a, H, n_env, T = 11, 3, 256, 8192, 128
N = n_env * T
rng = np.random.default_rng(0)
pol = ReparamMultivariateGaussianPolicy([H, H], o, a, conditioned_std=False, hidden_activation="tanh", ctx=ctx, seed=1)
vf = FlattenMlp([H, H], 1, o, hidden_activation="tanh", ctx=ctx, seed=2)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
tr = PPO(pol, vf, mini_batch_size=mb, update_epoch=1, gae_tau=0.95, max_samples=N)
obs = ctx.from_numpy(rng.normal(0, 1, (N, o)).astype(np.float32)); act = ctx.from_numpy(rng.normal(0, 0.5, (N, a)).astype(np.float32))
rew = ctx.from_numpy(rng.normal(1, 1, (N,)).astype(np.float32)); offs = (np.arange(n_env + 1) * T).astype(np.int32)
call = lambda: _lib.check(ctx.lib.ilsx_ppo_train(tr.h, obs.ptr, act.ptr, rew.ptr, offs.ctypes.data_as(C.c_void_p), n_env, None, None))
call(); ctx.sync()
_lib.check(ctx.lib.ilsx_prof_reset(ctx.h)); _lib.check(ctx.lib.ilsx_prof_enable(ctx.h, 1))
call()
_lib.check(ctx.lib.ilsx_prof_enable(ctx.h, 0))
for kid in range(16):
    nl, ms = C.c_uint64(), C.c_double()
    _lib.check(ctx.lib.ilsx_prof_read(ctx.h, kid, C.byref(nl), C.byref(ms)))
    if nl.value:
        print(f"{ctx.lib.ilsx_kernel_name(kid).decode():22s} launches {nl.value:5d} total {ms.value:9.3f} ms avg {1e3*ms.value/nl.value:9.1f} us")
