"""Debug: phase timestamps of k_mlp_fwd workgroup (0,0) for a Hopper-shaped critic forward (B=256)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
from ilswiss_amd import _lib
ctx = ia.Context(0)
buf = ctx.from_numpy(np.zeros(16, np.int64), np.int64)
net = ia.FlattenMlp([256, 256], 1, 14, ctx=ctx, seed=1)
pol = ia.ReparamTanhMultivariateGaussianPolicy([256, 256], 11, 3, ctx=ctx, seed=2)
x = ctx.from_numpy(np.random.randn(256, 14).astype(np.float32))
o = ctx.from_numpy(np.random.randn(256, 11).astype(np.float32))
y = ctx.empty((256, 1)); act = ctx.empty((256, 3)); lp = ctx.empty((256,))
_lib.check(ctx.lib.ilsx_debug_set_stamp_buffer(ctx.h, buf.ptr))
for name, fn in (("critic fwd", lambda: ctx.lib.ilsx_mlp_forward(net.h, x.ptr, 256, y.ptr)),
                 ("policy act", lambda: ctx.lib.ilsx_policy_act(pol.h, o.ptr, 256, 0, None, act.ptr, lp.ptr))):
    for rep in range(3):
        _lib.check(fn()); ctx.sync()
        t = buf.numpy()
        d = np.diff(t[[0, 1, 2, 3, 6, 7]])
        print(name, rep, "cycles: stage_x=%d layer0=%d layer1=%d heads=%d epilogue=%d total=%d" % (*d, t[7] - t[0]))
