"""PPO over libilsx: the reference's `PPO` trainer (rlkit/torch/algorithms/ppo/ppo.py:11-190) and its Gaussian
policy `ReparamMultivariateGaussianPolicy` (rlkit/torch/common/policies.py:348-478, both `conditioned_std` settings).
Constructor kwargs are the YAML `ppo_params` keys (exp_specs/ppo/ppo_hopper.yaml:41-50); unknown keys are
swallowed like the reference's **kwargs.  GAE, the fixed log-probs and every minibatch update run on the device
(csrc/ilsx_ppo.hip); this module only moves trajectories to HBM and builds the trajectory offset table.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .device import as_dev
from .networks import Mlp
from .sac import Trainer, check_swallowed_kwargs


class ReparamMultivariateGaussianPolicy(Mlp):
    """policies.py:348-478: mean network whose last layer is scaled by 0.1 (bias 0) after the usual init (:378-379), and either a
    state-independent `action_log_std` parameter (zeros; conditioned_std=False, what the reference's PPO scripts pass) or a second
    head `last_fc_log_std` clamped to [LOG_SIG_MIN, LOG_SIG_MAX] (conditioned_std=True, the class default, :368-374,401-405)."""

    def __init__(self, hidden_sizes, obs_dim, action_dim, conditioned_std=True, init_w=1e-3,
                 hidden_activation="relu", **kwargs):
        # the defaults are the reference's (policies.py:354, networks.py:30); its PPO scripts pass conditioned_std=False and
        # hidden_activation=torch.tanh (ppo_exp_script.py:90-96)
        self.conditioned_std = bool(conditioned_std)
        if self.conditioned_std:
            self.n_heads = 2   # fc.. | last_fc | last_fc_log_std: torch's parameters() order of the reference module
        super().__init__(hidden_sizes, input_size=obs_dim, output_size=action_dim, init_w=init_w,
                         hidden_activation=hidden_activation, **kwargs)
        self.obs_dim, self.action_dim = int(obs_dim), int(action_dim)
        flat = self.get_flat_params()
        nl, a = self.hidden_sizes[-1] * self.action_dim, self.action_dim
        tail = (nl + a) if self.conditioned_std else 0   # the log-std head sits behind last_fc and keeps its U(+-init_w) init
        flat[flat.size - tail - (nl + a):flat.size - tail - a] *= np.float32(0.1)
        flat[flat.size - tail - a:flat.size - tail] = 0.0
        self.set_flat_params(flat)
        self.action_log_std = None if self.conditioned_std else np.zeros(self.action_dim, np.float32)
        self._ppo = None   # set by PPO: from then on the trainer's device copy is the live one

    def ppo_flat(self):  # mean net | action_log_std   (conditioned_std: the two-head net alone)
        flat = self.get_flat_params()
        return flat if self.conditioned_std else np.concatenate([flat, self.action_log_std])

    def get_actions(self, obs_np, deterministic=False):  # policies.py:392-395
        if self._ppo is None:
            raise RuntimeError("bind the policy to a PPO trainer first (PPO(policy=..., vf=...))")
        return self._ppo.policy_act(obs_np, deterministic)[0]

    def get_action(self, obs_np, deterministic=False):
        return self.get_actions(np.asarray(obs_np)[None], deterministic)[0], {}


class PPO(Trainer):
    def __init__(self, policy, vf, mini_batch_size=64, clip_eps=0.2, reward_scale=1.0, discount=0.99, policy_lr=3e-4,
                 value_lr=3e-4, gae_tau=0.9, value_l2_reg=1e-3, use_value_clip=False, update_epoch=10,
                 lambda_entropy_policy=0.0, max_samples=16384, grad_world=1, **kwargs):
        # grad_world (an ilswiss_amd key): this process is one rank of a run split over G GPUs — mini_batch_size / max_samples are this rank's
        # rows, the value and policy gradient arenas are all-reduced by the library before their optimiser steps (ilsx_ppo_cfg.grad_world)
        check_swallowed_kwargs(kwargs, "PPO")
        self.on_policy = True  # ppo.py:30
        self.policy, self.vf, self.ctx = policy, vf, policy.ctx
        if vf.act != 1 or policy.act != 1:
            raise ValueError("PPO networks use tanh hidden units (ppo_exp_script.py:82-96)")
        if vf.hidden_sizes != policy.hidden_sizes:
            raise ValueError("policy and value net share net_size / num_hidden_layers (ppo_exp_script.py:79-80)")
        # any list of widths up to 256 (networks.py:23-60): narrower / unequal layers are structural zeros of the kernel width, as in every other
        # trainer (include/ilsx.h ilsx_mlp_cfg::hidden_sizes); flat parameter vectors keep the logical sizes
        self.mini_batch_size, self.update_epoch, self.max_samples = int(mini_batch_size), int(update_epoch), int(max_samples)
        self.o, self.a = policy.obs_dim, policy.action_dim
        hs = (C.c_int32 * 3)(*(list(policy.hidden_sizes) + [0] * (3 - len(policy.hidden_sizes))))
        cfg = _lib.PpoCfg(self.o, self.a, len(policy.hidden_sizes), policy.kernel_width, reward_scale, discount,
                          clip_eps, policy_lr, value_lr, gae_tau, value_l2_reg, self.mini_batch_size,
                          self.update_epoch, self.max_samples, int(bool(use_value_clip)), int(bool(getattr(policy, "conditioned_std", False))), hs, int(grad_world))
        self.grad_world = int(grad_world)
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_ppo_create(self.ctx.h, C.byref(cfg), C.byref(self.h)))
        self.set_flat_params(policy.ppo_flat(), vf.get_flat_params())
        policy._ppo = self
        self.eval_statistics = None

    # ---- parameters (0 = policy: mean net | action_log_std ; 1 = value net)
    def _num(self, which):
        n = C.c_size_t()
        _lib.check(self.ctx.lib.ilsx_ppo_num_params(self.h, which, C.byref(n)))
        return n.value

    def set_flat_params(self, pi_flat=None, vf_flat=None):
        for which, flat in ((0, pi_flat), (1, vf_flat)):
            if flat is not None:
                flat = np.ascontiguousarray(flat, np.float32)
                _lib.check(self.ctx.lib.ilsx_ppo_set_params(self.h, which, flat.ctypes.data_as(C.c_void_p), flat.size))

    def get_flat_params(self, which):
        out = np.empty(self._num(which), np.float32)
        _lib.check(self.ctx.lib.ilsx_ppo_get_params(self.h, which, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    # ---- data plumbing
    def _upload(self, trajs):
        lens = [int(np.shape(t["rewards"])[0]) for t in trajs]
        offs = np.zeros(len(lens) + 1, np.int32)
        offs[1:] = np.cumsum(lens)
        cat = lambda k, w: np.concatenate([np.asarray(t[k], np.float32).reshape(-1, w) for t in trajs])  # noqa: E731
        keep, ptrs = [], []
        for arr in (cat("observations", self.o), cat("actions", self.a), cat("rewards", 1)):
            k, p = as_dev(self.ctx, arr)
            keep.append(k)
            ptrs.append(p)
        return keep, ptrs, offs

    def calc_adv(self, trajs):
        """ppo.py:57-100 -> (returns, advantages, values, fixed log-probs), numpy [N,1]."""
        keep, (po, pa, pr), offs = self._upload(trajs)
        N = int(offs[-1])
        outs = [self.ctx.empty((N,)) for _ in range(4)]
        _lib.check(self.ctx.lib.ilsx_ppo_gae(self.h, po, pa, pr, offs.ctypes.data_as(C.c_void_p), len(offs) - 1, None,
                                             *[x.ptr for x in outs]))
        return tuple(x.numpy().reshape(N, 1) for x in outs)

    def train_step(self, trajs, perms=None):
        """ppo.py:102-170.  trajs: list of dicts with observations / actions / rewards.  perms (update_epoch x N)
        replaces the library's own shuffles (parity tests)."""
        keep, (po, pa, pr), offs = self._upload(trajs)
        pp = None
        if perms is not None:
            perms = np.ascontiguousarray(perms, np.int32)
            assert perms.shape == (self.update_epoch, int(offs[-1]))
            pp = perms.ctypes.data_as(C.c_void_p)
        _lib.check(self.ctx.lib.ilsx_ppo_train(self.h, po, pa, pr, offs.ctypes.data_as(C.c_void_p), len(offs) - 1, None, pp))
        if self.eval_statistics is None:
            self.eval_statistics = OrderedDict()

    def train_from_rollout(self, env, horizon, max_path_length=1000, bootstrap=True):
        """One on-policy iteration on the device: `horizon` vec steps of all envs (ilsx_ppo_rollout), then train_step on
        the collected segments.  A segment is one env's samples up to an episode end or the end of the rollout.
        bootstrap=False is the reference: V after a segment's last sample is 0 everywhere (ppo.py:74; it only ever sees
        whole episodes).  bootstrap=True uses vf(next observation) for segments cut by the end of the rollout
        (fixed 8192 x 128 rollouts need it)."""
        ctx, n, T = self.ctx, len(env), int(horizon)
        N = n * T
        if N > self.max_samples:
            raise ValueError(f"{n} envs x {T} steps exceed max_samples={self.max_samples}")
        if getattr(self, "_roll", None) is None or self._roll[0] != (n, T):
            self._roll = ((n, T), ctx.empty((N, self.o)), ctx.empty((N, self.a)), ctx.empty((N,)), ctx.empty((N,), np.uint8),
                          ctx.empty((n,)))
        _, obs, act, rew, ends, lastv = self._roll
        _lib.check(ctx.lib.ilsx_ppo_rollout(self.h, env.h, T, int(max_path_length), obs.ptr, act.ptr, rew.ptr, ends.ptr, lastv.ptr))
        e = ends.numpy().reshape(n, T).astype(bool)
        cut = e.copy()
        cut[:, -1] = True                                   # the rollout's end closes every env's last segment
        offs = np.concatenate([[0], np.flatnonzero(cut.ravel()) + 1]).astype(np.int32)
        b = np.zeros(offs.size - 1, np.float32)   # always given: also arms the one-sample-segment guard of k_ppo_gae
        if bootstrap:
            open_env = np.flatnonzero(~e[:, -1])            # last segment still running: bootstrap with V(s_T)
            seg_of_last = np.searchsorted(offs, (open_env + 1) * T, side="left") - 1
            b[seg_of_last] = lastv.numpy()[open_env]
        boot = ctx.from_numpy(b)
        _lib.check(ctx.lib.ilsx_ppo_train(self.h, obs.ptr, act.ptr, rew.ptr, offs.ctypes.data_as(C.c_void_p), offs.size - 1,
                                          boot.ptr, None))
        if self.eval_statistics is None:
            self.eval_statistics = OrderedDict([("PPO Segments", float(offs.size - 1)), ("PPO Samples", float(N))])
        return N

    def policy_act(self, obs, deterministic=False, eps=None):
        obs = np.ascontiguousarray(obs, np.float32)
        n = obs.shape[0]
        ko, po = as_dev(self.ctx, obs)
        ke, pe = as_dev(self.ctx, np.ascontiguousarray(eps, np.float32)) if eps is not None else (None, None)
        act, lp = self.ctx.empty((n, self.a)), self.ctx.empty((n,))
        _lib.check(self.ctx.lib.ilsx_ppo_policy_act(self.h, po, n, int(bool(deterministic)), pe, act.ptr,
                                                    None if deterministic else lp.ptr))
        return act.numpy(), (None if deterministic else lp.numpy().reshape(n, 1))

    # ---- Trainer API
    @property
    def networks(self):
        return [self.policy, self.vf]

    def get_snapshot(self):   # ppo.py:178-185 (+ both optimisers' Adam state; the env's obs_rms is added by the algorithm)
        from .snapshot import get_opt
        snap = dict(policy=self.get_flat_params(0), vf=self.get_flat_params(1))
        snap["policy_optimizer"] = get_opt(self.ctx.lib, "ppo", self.h, snap["policy"].size, 0)
        snap["vf_optimizer"] = get_opt(self.ctx.lib, "ppo", self.h, snap["vf"].size, 1)
        return snap

    def load_snapshot(self, snap):
        from .snapshot import set_opt
        self.set_flat_params(snap["policy"], snap["vf"])
        if "policy_optimizer" in snap:
            set_opt(self.ctx.lib, "ppo", self.h, snap["policy_optimizer"], 0)
            set_opt(self.ctx.lib, "ppo", self.h, snap["vf_optimizer"], 1)

    def get_eval_statistics(self):
        return self.eval_statistics

    def end_epoch(self):
        self.eval_statistics = None

    def to(self, device=None):   # the networks already live on the library's device
        return self
