"""Import harness for the read-only reference checkout (survey container ONLY).

Used exclusively by tools/make_golden.py to (1) validate our restatement in oracle/ and
(2) emit golden input/output vectors under tests/golden/.  Nothing from /root/reference is
copied; only plain arrays produced by running it are stored.  /root/reference does not exist
on the GPU box, so nothing under tests/, bench.py or __graft_entry__.py imports this module.

Recipe documented in SURVEY.md Appendix C.
"""
import builtins
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("seaborn", set=lambda *a, **k: None)  # rlkit/core/vistools.py:9
    stub("gtimer")  # rlkit/core/base_algorithm.py:5

    class Box:  # minimal gym.spaces.Box used by env_replay_buffer.get_dim
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low, dtype=np.float64)
            self.high = np.asarray(high, dtype=np.float64)
            self.shape = self.low.shape if shape is None else tuple(shape)
            self.dtype = dtype

        @property
        def size(self):
            return int(np.prod(self.shape))

    sp = stub(
        "gym.spaces",
        Box=Box,
        **{n: type(n, (), {}) for n in ("Discrete", "Tuple", "Dict")},
    )
    stub("gym", spaces=sp, Env=type("Env", (), {}), Space=object)
    stub("torch.utils.tensorboard", SummaryWriter=object)  # rlkit/core/logger.py:28
    builtins.torch = torch  # rlkit/torch/algorithms/torch_base_algorithm.py:24 (missing import)


class NoiseInjector:
    """Replaces torch.randn / torch.rand / torch.normal so the reference consumes noise we chose.

    Reference draw sites: distributions.py:24 (torch.randn), policies.py:182 (torch.normal),
    adv_irl.py / pytorch_util.py:119-122 (torch.rand).
    """

    def __init__(self):
        self.queue = []
        self._orig = {}

    def push(self, arr):
        self.queue.append(torch.as_tensor(np.asarray(arr)))

    def _pop(self, shape, what):
        assert self.queue, f"reference asked for {what}{tuple(shape)} but no noise queued"
        t = self.queue.pop(0)
        assert tuple(t.shape) == tuple(shape), (what, tuple(t.shape), tuple(shape))
        return t.clone()

    def __enter__(self):
        self._orig = dict(randn=torch.randn, rand=torch.rand, normal=torch.normal)

        def _shape(args):
            if len(args) == 1 and not isinstance(args[0], int):
                return tuple(args[0])
            return tuple(args)

        def randn(*size, **kw):
            return self._pop(_shape(size), "randn")

        def rand(*size, **kw):
            return self._pop(_shape(size), "rand")

        def normal(mean, *a, **kw):
            return self._pop(tuple(mean.shape), "normal")

        torch.randn, torch.rand, torch.normal = randn, rand, normal
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig["randn"]
        torch.rand = self._orig["rand"]
        torch.normal = self._orig["normal"]
        assert not self.queue, "unused injected noise"
