"""Debug: phase timestamps of the column-split kernels inside a SAC step (Hopper dims, B=256)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
from ilswiss_amd import _lib
ctx = ia.Context(0)
buf = ctx.from_numpy(np.zeros(16, np.int64), np.int64)
o, a, H, B = 11, 3, [256, 256], 256
pol = ia.ReparamTanhMultivariateGaussianPolicy(H, o, a, ctx=ctx, seed=2)
q1, q2 = ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=3), ia.FlattenMlp(H, 1, o + a, ctx=ctx, seed=4)
tr = ia.SoftActorCritic(pol, q1, q2, max_batch=B)
rng = np.random.default_rng(0)
batch = dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32), actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
             rewards=rng.normal(0, 1, (B, 1)).astype(np.float32), terminals=np.zeros((B, 1), np.float32),
             next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))
_lib.check(ctx.lib.ilsx_debug_set_stamp_buffer(ctx.h, buf.ptr))
for rep in range(4):
    tr.train_step(batch); ctx.sync()
    t = buf.numpy()
    print("   dW (policy launch, tile 5): loads+mfma=%d lds=%d reduce+adam+store=%d total=%d" % (*np.diff(t[4:8]), t[7] - t[4]))
    f = np.diff(t[[0, 1, 2, 3, 7]]); b = np.diff(t[[8, 9, 10, 11, 12]])
    print(rep, "fwd(last launch: policy bwd? no: Q1n/Q2n) x=%d layer0=%d layer1=%d head=%d total=%d | bwd(last: policy) dout=%d delta1=%d delta0=%d dx=%d total=%d"
          % (*f, t[7] - t[0], *b, t[12] - t[8]))
