"""Mlp / FlattenMlp and the tanh-Gaussian policy: the reference's model classes
(rlkit/torch/common/networks.py:23-115, rlkit/torch/common/policies.py:19-36,191-345,
rlkit/policies/base.py:4-24) re-exposed over libilsx handles.  Same constructor argument names;
arithmetic happens in HIP (csrc/kernels.h), these classes only move data.
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import DevArray, as_dev, get_context

_ACT = {"relu": 0, "tanh": 1}
_WIDTHS = (64, 128, 256)   # hidden widths the kernels are instantiated for


def _act_code(fn):
    if isinstance(fn, str):
        return _ACT[fn]
    name = getattr(fn, "__name__", str(fn))
    if name in _ACT:
        return _ACT[name]
    raise ValueError(f"hidden_activation {fn!r}: libilsx implements relu and tanh")


class Mlp:
    """networks.py:23-101.  `parameters()` order == flat layout of include/ilsx.h."""

    n_heads = 1

    def __init__(self, hidden_sizes, output_size, input_size, init_w=3e-3, hidden_activation="relu",
                 b_init_value=0.1, layer_norm=False, batch_norm=False, ctx=None, seed=None, **kwargs):
        if layer_norm or batch_norm:
            raise NotImplementedError("layer_norm / batch_norm are off in every hot-path config (SURVEY §2 #15)")
        # the rest of Mlp's keywords (networks.py:31-38): their defaults are what libilsx computes; another value is refused, not ignored.
        # (`output_activation` is consumed by MlpGaussianNoisePolicy before it gets here — the one class that applies it.)
        oa = kwargs.pop("output_activation", None)
        if oa is not None and getattr(oa, "__name__", oa) != "identity":
            raise NotImplementedError(f"Mlp(output_activation={getattr(oa, '__name__', oa)}): the head is linear (identity) here")
        hi = kwargs.pop("hidden_init", None)
        if hi is not None and getattr(hi, "__name__", hi) != "fanin_init":
            raise NotImplementedError("Mlp(hidden_init=...): libilsx initialises hidden layers with ptu.fanin_init (pytorch_util.py:20-29)")
        if kwargs.pop("batch_norm_before_output_activation", False):
            raise NotImplementedError("batch_norm_before_output_activation")
        kwargs.pop("layer_norm_kwargs", None)   # only read when layer_norm is on
        if kwargs:
            raise TypeError(f"{type(self).__name__}: unexpected keyword arguments {sorted(kwargs)}")
        hidden_sizes = [int(h) for h in hidden_sizes]
        # networks.py:23-60 takes any list of widths.  The kernels run at 64 / 128 / 256: a narrower (or unequal) layer is embedded as structural
        # zeros by the library (include/ilsx.h ilsx_mlp_cfg::hidden_sizes) — the padded network IS the hidden_sizes network, and every flat
        # parameter / gradient / optimiser vector keeps the logical sizes.  Wider than 256 has no kernel.
        if not 1 <= len(hidden_sizes) <= 3 or min(hidden_sizes) < 1:
            raise NotImplementedError(f"hidden_sizes={hidden_sizes}: libilsx runs 1..3 hidden layers")
        if max(hidden_sizes) > _WIDTHS[-1]:
            raise NotImplementedError(f"hidden_sizes={hidden_sizes}: the widest kernel instantiation is {_WIDTHS[-1]}")
        self.ctx = ctx or get_context()
        self.hidden_sizes, self.input_size, self.output_size = hidden_sizes, int(input_size), int(output_size)
        self.kernel_width = next(w for w in _WIDTHS if w >= max(hidden_sizes))
        self.init_w, self.b_init_value = float(init_w), float(b_init_value)
        self.act = _act_code(hidden_activation)
        cfg = self._cfg()
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_net_create(self.ctx.h, C.byref(cfg), C.byref(self.h)))
        n = C.c_size_t()
        _lib.check(self.ctx.lib.ilsx_net_num_params(self.h, C.byref(n)))
        self.num_params = n.value
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1))  # like torch's global RNG, follows np.random.seed / set_seed
        _lib.check(self.ctx.lib.ilsx_net_init(self.h, C.c_uint64(seed), self.init_w, self.b_init_value))

    def _cfg(self):
        hs = (C.c_int32 * 3)(*(self.hidden_sizes + [0] * (3 - len(self.hidden_sizes))))
        return _lib.MlpCfg(self.input_size, len(self.hidden_sizes), self.kernel_width, self.output_size, self.n_heads, self.act, hs)

    # -- parameters
    def get_flat_params(self):
        out = np.empty(self.num_params, np.float32)
        _lib.check(self.ctx.lib.ilsx_net_get_params(self.h, out.ctypes.data_as(C.c_void_p), out.size, 0))
        return out

    def set_flat_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        _lib.check(self.ctx.lib.ilsx_net_set_params(self.h, flat.ctypes.data_as(C.c_void_p), flat.size, 0))

    def copy(self, ctx=None):  # PyTorchModule.copy (rlkit/torch/core.py:32-35); ctx: the copy lives in another context of the same device
        c = type(self).__new__(type(self))
        c.__dict__.update({k: v for k, v in self.__dict__.items() if k != "h"})
        if ctx is not None:
            c.ctx = ctx
        cfg = self._cfg()
        c.h = C.c_void_p()
        _lib.check(c.ctx.lib.ilsx_net_create(c.ctx.h, C.byref(cfg), C.byref(c.h)))
        c.set_flat_params(self.get_flat_params())
        return c

    def train(self, mode=True):
        return self

    def to(self, device=None):
        return self

    # -- forward
    def forward_dev(self, x_dev_ptr, rows):
        out = self.ctx.empty((rows, self.n_heads * self.output_size))
        _lib.check(self.ctx.lib.ilsx_mlp_forward(self.h, x_dev_ptr, rows, out.ptr))
        return out

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32)
        keep, p = as_dev(self.ctx, x)
        return self.forward_dev(p, x.shape[0]).numpy()

    __call__ = forward


class FlattenMlp(Mlp):
    """networks.py:108-115: cat(inputs, dim=1) then Mlp."""

    def forward(self, *inputs):
        return super().forward(np.concatenate([np.asarray(i, np.float32) for i in inputs], axis=1))

    __call__ = forward


class ReparamTanhMultivariateGaussianPolicy(Mlp):
    """policies.py:191-345.  forward() returns the reference's 8-tuple (numpy)."""

    n_heads = 2

    def __init__(self, hidden_sizes, obs_dim, action_dim, init_w=1e-3, max_act=1.0, conditioned_std=True, **kwargs):
        if not conditioned_std:
            raise NotImplementedError("conditioned_std=False is the PPO policy (ReparamMultivariateGaussianPolicy)")
        super().__init__(hidden_sizes, input_size=obs_dim, output_size=action_dim, init_w=init_w, **kwargs)
        self.max_act, self.obs_dim, self.action_dim = max_act, int(obs_dim), int(action_dim)

    def get_action(self, obs_np, deterministic=False):  # policies.py:241-243
        return self.get_actions(np.asarray(obs_np)[None], deterministic=deterministic)[0, :], {}

    def get_actions(self, obs_np, deterministic=False):  # policies.py:245-246
        obs_np = np.ascontiguousarray(obs_np, np.float32)
        keep, p = as_dev(self.ctx, obs_np)
        return self.get_actions_dev(p, obs_np.shape[0], deterministic).numpy()

    def get_actions_dev(self, obs_ptr, n, deterministic=False, eps_ptr=None, logp=None):
        act = self.ctx.empty((n, self.action_dim))
        _lib.check(self.ctx.lib.ilsx_policy_act(self.h, obs_ptr, n, int(bool(deterministic)), eps_ptr, act.ptr,
                                                logp.ptr if logp is not None else None))
        return act

    def set_num_steps_total(self, t):
        pass

    def forward(self, obs, deterministic=False, return_log_prob=False, eps=None):
        """eps: optional explicit N(0,1) [n,a] (parity mode); otherwise Philox."""
        obs = np.ascontiguousarray(obs, np.float32)
        n = obs.shape[0]
        keep, p = as_dev(self.ctx, obs)
        raw = self.forward_dev(p, n).numpy()
        mean, log_std = raw[:, : self.action_dim], np.clip(raw[:, self.action_dim:], -20.0, 2.0)
        logp = self.ctx.empty((n,)) if return_log_prob and not deterministic else None
        ek = ep = None
        if eps is not None:
            ek, ep = as_dev(self.ctx, np.ascontiguousarray(eps, np.float32))
        act = self.get_actions_dev(p, n, deterministic, ep, logp).numpy()
        log_prob = logp.numpy().reshape(n, 1) if logp is not None else None
        return (act, mean, log_std, log_prob, None, np.exp(log_std), None, None)

    __call__ = forward

    def get_log_prob(self, obs, acts, return_normal_params=False):  # policies.py:329-345
        obs = np.ascontiguousarray(obs, np.float32)
        acts = np.ascontiguousarray(acts, np.float32)
        n = obs.shape[0]
        ko, po = as_dev(self.ctx, obs)
        ka, pa = as_dev(self.ctx, acts)
        lp = self.ctx.empty((n,))
        _lib.check(self.ctx.lib.ilsx_policy_log_prob(self.h, po, pa, n, lp.ptr))
        out = lp.numpy().reshape(n, 1)
        if return_normal_params:
            raw = self.forward_dev(po, n).numpy()
            return out, raw[:, : self.action_dim], np.clip(raw[:, self.action_dim:], -20.0, 2.0)
        return out


class MakeDeterministic:
    """policies.py:19-36."""

    def __init__(self, stochastic_policy):
        self.stochastic_policy = stochastic_policy

    def get_action(self, observation):
        return self.stochastic_policy.get_action(observation, deterministic=True)

    def get_actions(self, observations):
        return self.stochastic_policy.get_actions(observations, deterministic=True)

    def train(self, mode):
        pass

    def set_num_steps_total(self, num):
        pass

    def to(self, device):
        pass
