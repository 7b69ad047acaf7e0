"""rlkit/envs/terminals.py:6-117 on the device: the batched terminal predicates `is_terminal(obs, act, next_obs) -> bool[n,1]`
(model-based rollouts label imagined transitions with them), same class names and the same `get_terminal_func` name rule.
The arithmetic is libilsx's `ilsx_is_terminal`; numpy arrays are uploaded / downloaded around it, device arrays
(`ilswiss_amd.device.DevArray`) are used in place."""
import ctypes as C

import numpy as np

from .. import _lib
from ..device import DevArray, as_dev, get_context

KINDS = dict(InvertedPendulum=0, InvertedDoublePendulum=1, Hopper=2, Walker2d=3, Halfcheetah=4, Humanoid=5, Ant=6)


def _is_terminal(kind, obs, act, next_obs, ctx=None):
    ctx = ctx or get_context()
    on_device = isinstance(next_obs, DevArray)
    shape = next_obs.shape
    assert len(shape) == 2, "is_terminal expects [n, obs_dim] arrays (terminals.py asserts ndim == 2)"
    n, o = int(shape[0]), int(shape[1])
    keep, ptr = (next_obs, next_obs.ptr) if on_device else as_dev(ctx, np.ascontiguousarray(next_obs, np.float32))
    done = ctx.empty((n,), np.uint8)
    _lib.check(ctx.lib.ilsx_is_terminal(ctx.h, kind, ptr, n, o, done.ptr))
    del keep
    return done if on_device else done.numpy().astype(bool)[:, None]


class TerminalFunc:
    kind = None

    @classmethod
    def is_terminal(cls, obs, act, next_obs, ctx=None):
        if cls.kind is None:
            raise NotImplementedError
        return _is_terminal(cls.kind, obs, act, next_obs, ctx)


for _name, _kind in KINDS.items():
    globals()[_name + "TerminalFunc"] = type(_name + "TerminalFunc", (TerminalFunc,), dict(kind=_kind))


def get_terminal_func(env_name):
    """terminals.py:6-11: 'inverted_double_pendulum' -> InvertedDoublePendulumTerminalFunc.is_terminal"""
    cls_name = "".join(s[0].upper() + s[1:] for s in env_name.split("_")) + "TerminalFunc"
    return globals()[cls_name].is_terminal
