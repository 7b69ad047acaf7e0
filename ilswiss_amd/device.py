"""Device context + device arrays for the ctypes layer.

Mirrors the role of rlkit/torch/utils/pytorch_util.py:50-92 (device globals, from_numpy, get_numpy):
`set_gpu_mode` picks the GPU, `from_numpy` / `get_numpy` move fp32 data across PCIe.  Storage is
plain HIP memory owned by the ilsx context; DevArray exposes `__cuda_array_interface__` so a torch
tensor can alias it zero-copy (used for the RCCL gradient all-reduce), torch is never on the hot path.
"""
import ctypes as C

import numpy as np

from . import _lib

_default_ctx = None
_gpu_id = 0


class Context:
    """One libilsx context: device, HIP stream, Philox key (`seed`) and the per-object stream-id counter.
    `stream`: an existing hipStream_t to enqueue on (include/ilsx.h ilsx_ctx_create).  `Context.sibling(seed)`: the context of another run
    in the same process (run_experiment.py --group): same device, a stream, key and object counter of its own — exactly what the run would
    have in a process of its own, so its random streams do not depend on its company; grouped launches are fenced against the runs' streams
    by the library (ilsx_sac_group_train_from_replay)."""

    def __init__(self, device=0, seed=0, stream=None):
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.ilsx_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.c_uint64(int(seed) & (2**64 - 1)), C.byref(h)))
        self.lib, self.h, self.device, self.seed = lib, h, int(device), int(seed)
        self.parent = None

    def sibling(self, seed, share_stream=False):
        c = Context(self.device, seed, stream=self.stream if share_stream else None)
        c.parent = self     # a shared stream's owner must outlive it
        return c

    def rng_stream_cursor(self, set_to=0):
        """the Philox stream id the next random-drawing object of this context takes (include/ilsx.h ilsx_ctx_rng_stream_cursor); set_to > 0 moves it"""
        cur = C.c_uint32()
        _lib.check(self.lib.ilsx_ctx_rng_stream_cursor(self.h, int(set_to), C.byref(cur)))
        return cur.value

    def sync(self):
        _lib.check(self.lib.ilsx_ctx_sync(self.h))

    @property
    def stream(self):
        return self.lib.ilsx_ctx_stream(self.h)

    def empty(self, shape, dtype=np.float32):
        return DevArray(self, shape, dtype)

    def from_numpy(self, arr, dtype=np.float32):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = DevArray(self, arr.shape, dtype)
        if arr.size:
            _lib.check(self.lib.ilsx_memcpy_h2d(self.h, d.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return d

    def close(self):
        if self.h:
            self.lib.ilsx_ctx_destroy(self.h)
            self.h = None


class DevArray:
    """Contiguous device array (fp32 / int64 / uint8) owned by a Context."""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.check(ctx.lib.ilsx_ctx_alloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self._p = p

    @property
    def ptr(self):
        return self._p

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if out.size:
            _lib.check(self.ctx.lib.ilsx_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self._p, out.nbytes))
        return out

    def copy_from(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.shape == self.shape, (arr.shape, self.shape)
        _lib.check(self.ctx.lib.ilsx_memcpy_h2d(self.ctx.h, self._p, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    @property
    def __cuda_array_interface__(self):
        return dict(shape=self.shape, typestr=self.dtype.str, data=(int(self._p.value), False), version=2)

    def free(self):
        if self._p is not None and self.ctx.h:
            self.ctx.lib.ilsx_ctx_free(self.ctx.h, self._p)
        self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class RawView:
    """Non-owning device view (e.g. the gradient arena) with __cuda_array_interface__."""

    def __init__(self, ptr, n, dtype=np.float32):
        self.ptr, self.shape, self.dtype = ptr, (int(n),), np.dtype(dtype)

    @property
    def __cuda_array_interface__(self):
        return dict(shape=self.shape, typestr=self.dtype.str, data=(int(self.ptr), False), version=2)


def set_gpu_mode(mode=True, gpu_id=0, seed=0):
    """pytorch_util.py:55-66 equivalent: select the GPU the default context lives on."""
    global _default_ctx, _gpu_id
    if not mode:
        raise RuntimeError("ilswiss_amd runs on an MI355X only; there is no CPU mode")
    _gpu_id = int(gpu_id)
    _default_ctx = Context(_gpu_id, seed)
    return _default_ctx


def set_default_context(ctx):
    """Make `ctx` the context objects built without an explicit `ctx=` land in (grouped runs: one sibling context per run)."""
    global _default_ctx
    _default_ctx = ctx
    return ctx


def get_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(_gpu_id, 0)
    return _default_ctx


def as_dev(ctx, x, dtype=np.float32):
    """numpy / torch(cpu or cuda) / DevArray -> (keepalive, ctypes pointer)."""
    if isinstance(x, DevArray):
        assert x.dtype == np.dtype(dtype), (x.dtype, dtype)
        return x, x.ptr
    if hasattr(x, "is_cuda"):  # torch tensor
        import torch
        if x.is_cuda:
            tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int64): torch.int64,
                   np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
            x = x.detach().to(tdt).contiguous()
            torch.cuda.current_stream(x.device).synchronize()  # torch's stream -> ours
            return x, C.c_void_p(x.data_ptr())
        x = x.detach().cpu().numpy()
    d = ctx.from_numpy(np.asarray(x), dtype)
    return d, d.ptr
