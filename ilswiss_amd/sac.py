"""SoftActorCritic trainer (twin Q, auto alpha) over libilsx — the reference's
rlkit/torch/algorithms/sac/sac_alpha.py:13-284 interface (Trainer ABC: rlkit/core/trainer.py:4-28).
Constructor kwargs are the YAML `sac_params` keys (exp_specs/sac/sac_hopper.yaml:36-47); unknown keys
are swallowed like the reference's **kwargs (sac_alpha.py:39).
"""
import abc
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .device import RawView, as_dev, get_context


def check_swallowed_kwargs(kwargs, who):
    """The reference's trainers take **kwargs and ignore what they do not know; two of the keys they DO read select something libilsx
    does not have — another optimiser than Adam (`optimizer_class`, e.g. sac_alpha.py:35) or another critic criterion than MSE
    (`qf_criterion`, td3.py:36): those fail loudly, everything else is swallowed as in the reference."""
    oc = kwargs.get("optimizer_class")
    if oc is not None and getattr(oc, "__name__", oc) != "Adam":
        raise NotImplementedError(f"{who}(optimizer_class={getattr(oc, '__name__', oc)}): libilsx implements torch.optim.Adam")
    qc = kwargs.get("qf_criterion")
    if qc is not None:   # an nn.MSELoss instance, the class itself, or its name; anything else (L1Loss, "huber", ...) is refused, not trained as MSE
        name = qc if isinstance(qc, str) else (getattr(qc, "__name__", None) or type(qc).__name__)
        if name.lower() not in ("mseloss", "mse"):
            raise NotImplementedError(f"{who}(qf_criterion={qc!r}): libilsx implements the MSE criterion")


class Trainer(metaclass=abc.ABCMeta):  # rlkit/core/trainer.py:4-28
    @abc.abstractmethod
    def train_step(self, batch):
        pass

    def get_eval_statistics(self):
        return None

    def get_snapshot(self):
        return {}

    def end_epoch(self):
        pass

    @property
    @abc.abstractmethod
    def networks(self):
        pass


def create_stats_ordered_dict(name, data):  # rlkit/core/eval_util.py:create_stats_ordered_dict (Mean/Std/Max/Min)
    data = np.asarray(data, dtype=np.float64)
    if data.size == 1:
        return OrderedDict({name: float(data.ravel()[0])})
    return OrderedDict([(name + " Mean", np.mean(data)), (name + " Std", np.std(data)),
                        (name + " Max", np.max(data)), (name + " Min", np.min(data))])


class SoftActorCritic(Trainer):
    WHICH = dict(policy=0, qf1=1, qf2=2, target_qf1=3, target_qf2=4)

    def __init__(self, policy, qf1, qf2, reward_scale=1.0, discount=0.99, policy_lr=1e-3, qf_lr=1e-3,
                 alpha_lr=3e-4, soft_target_tau=1e-2, alpha=0.2, train_alpha=True,
                 policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3, beta_1=0.9, target_entropy=None,
                 max_batch=1024, grad_world=1, **kwargs):
        check_swallowed_kwargs(kwargs, type(self).__name__)
        self.policy, self.qf1, self.qf2 = policy, qf1, qf2
        self.ctx = policy.ctx
        self.reward_scale, self.discount, self.soft_target_tau = reward_scale, discount, soft_target_tau
        self.train_alpha = train_alpha
        if target_entropy is None and "env" in kwargs:  # sac_alpha.py:55-58
            target_entropy = -np.prod(kwargs["env"].action_space.shape) / 2.0
        cfg = _lib.SacCfg(reward_scale, discount, policy_lr, qf_lr, alpha_lr, soft_target_tau, alpha,
                          int(bool(train_alpha)), policy_mean_reg_weight, policy_std_reg_weight, beta_1,
                          int(target_entropy is not None), float(target_entropy or 0.0), int(max_batch),
                          int(grad_world))
        self.target_entropy = target_entropy if target_entropy is not None else -policy.action_dim / 2.0
        self.max_batch, self.grad_world = int(max_batch), int(grad_world)
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_sac_create(self.ctx.h, C.byref(cfg), policy.h, qf1.h, qf2.h, C.byref(self.h)))
        self.eval_statistics = None
        self._stats = _lib.SacStats()

    # ---- Trainer API
    def train_step(self, batch, eps_next=None, eps_cur=None):
        """batch: dict with observations, actions, rewards, terminals, next_observations (numpy, torch or
        DevArray, fp32 [B,.]).  eps_*: explicit N(0,1) [B,a] draws (parity mode), else Philox."""
        ctx, keep = self.ctx, []

        def dev(x):
            k, p = as_dev(ctx, x)
            keep.append(k)
            return p
        obs = batch["observations"]
        B = int(obs.shape[0])
        flat = lambda v: v.reshape(B) if hasattr(v, "reshape") else v  # noqa: E731
        p = [dev(obs), dev(batch["actions"]), dev(flat(batch["rewards"])), dev(flat(batch["terminals"])),
             dev(batch["next_observations"])]
        e1 = dev(eps_next) if eps_next is not None else None
        e2 = dev(eps_cur) if eps_cur is not None else None
        want = self.eval_statistics is None
        _lib.check(ctx.lib.ilsx_sac_train_step(self.h, *p, B, e1, e2, C.byref(self._stats) if want else None))
        if want:
            self._fill_stats()
        if not want:
            ctx.sync()  # `keep` buffers must outlive the asynchronous step

    def train_from_replay(self, replay_buffer, n_steps, batch_size):
        """TorchRLAlgorithm._do_training (torch_rl_algorithm.py:28-34) with on-device sampling."""
        want = self.eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_sac_train_from_replay(self.h, replay_buffer.h, int(n_steps), int(batch_size),
                                                           C.byref(self._stats) if want else None))
        if want:
            self._fill_stats()

    def phase_state(self):
        """How the merged phase kernels of this agent's train windows are doing (include/ilsx.h ilsx_sac_phase_state): a shared GPU can
        break their in-launch hand-offs; the library then rolls the window back, re-runs it on one launch per stage and stays there."""
        v = [C.c_int() for _ in range(5)]
        _lib.check(self.ctx.lib.ilsx_sac_phase_state(self.h, *[C.byref(x) for x in v]))
        return dict(fallbacks=v[0].value, disabled=bool(v[1].value), last_window_on_phase=bool(v[2].value),
                    wgs_per_cu_a=v[3].value, wgs_per_cu_c=v[4].value)

    def _fill_stats(self):  # sac_alpha.py:186-233
        s, st = self._stats, OrderedDict()
        st["Reward Scale"] = self.reward_scale
        st["QF1 Loss"], st["QF2 Loss"] = s.qf1_loss, s.qf2_loss
        if self.train_alpha:
            st["Alpha Loss"] = s.alpha_loss
        st["Policy Loss"] = s.policy_loss

        def block(name, mean, i):  # create_stats_ordered_dict: Mean / Std / Max / Min
            st[name + " Mean"], st[name + " Std"], st[name + " Max"], st[name + " Min"] = mean, s.ext_std[i], s.ext_max[i], s.ext_min[i]
        block("Q1 Predictions", s.q1_mean, 0)
        block("Q2 Predictions", s.q2_mean, 1)
        st["Alpha"] = s.alpha
        block("Log Pis", s.log_pi_mean, 2)
        block("Policy mu", s.policy_mu_mean, 3)
        block("Policy log std", s.policy_log_std_mean, 4)
        self.eval_statistics = st

    def get_eval_statistics(self):
        return self.eval_statistics

    def end_epoch(self):
        self.eval_statistics = None

    @property
    def networks(self):
        return [self.policy, self.qf1, self.qf2]

    def to(self, device=None):
        return self

    # ---- parameter / optimiser access (snapshots, parity tests)
    def _n(self, which):
        return self.policy.num_params if which == 0 else self.qf1.num_params

    def get_params(self, name):
        w = self.WHICH[name]
        out = np.empty(self._n(w), np.float32)
        _lib.check(self.ctx.lib.ilsx_sac_get_params(self.h, w, out.ctypes.data_as(C.c_void_p), out.size, 0))
        return out

    def set_params(self, name, flat):
        w = self.WHICH[name]
        flat = np.ascontiguousarray(flat, np.float32)
        _lib.check(self.ctx.lib.ilsx_sac_set_params(self.h, w, flat.ctypes.data_as(C.c_void_p), flat.size, 0))

    def get_grads(self, name):
        w = self.WHICH[name]
        out = np.empty(self._n(w), np.float32)
        _lib.check(self.ctx.lib.ilsx_sac_get_grads(self.h, w, out.ctypes.data_as(C.c_void_p), out.size, 0))
        return out

    @property
    def log_alpha(self):
        v = C.c_double()
        _lib.check(self.ctx.lib.ilsx_sac_get_log_alpha(self.h, C.byref(v)))
        return v.value

    @log_alpha.setter
    def log_alpha(self, v):
        _lib.check(self.ctx.lib.ilsx_sac_set_log_alpha(self.h, float(v)))

    def grad_tensor(self, segment):
        """torch tensor aliasing gradient-arena segment `segment` (SplitRunStep's trainer interface)."""
        import torch
        return torch.as_tensor(self.grads_view(segment), device=f"cuda:{self.ctx.device}")

    def grads_view(self, segment):
        """Non-owning device view of the gradient arena (0 = critics, 1 = actor + alpha slot) for the
        RCCL all-reduce: `torch.as_tensor(view, device='cuda')` aliases it."""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self.ctx.lib.ilsx_sac_grads_ptr(self.h, segment, C.byref(p), C.byref(n)))
        return RawView(p.value, n.value)

    def get_snapshot(self):  # sac_alpha.py:249-261, as plain arrays
        snap = {k: self.get_params(k) for k in self.WHICH}
        snap["log_alpha"] = self.log_alpha
        for k, w in (("policy", 0), ("qf1", 1), ("qf2", 2)):
            m, v = np.empty(self._n(w), np.float32), np.empty(self._n(w), np.float32)
            t = C.c_int64()
            _lib.check(self.ctx.lib.ilsx_sac_get_adam(self.h, w, m.ctypes.data_as(C.c_void_p),
                                                      v.ctypes.data_as(C.c_void_p), m.size, C.byref(t)))
            snap[k + "_optimizer"] = dict(exp_avg=m, exp_avg_sq=v, step=t.value)
        m, v, t, r = C.c_double(), C.c_double(), C.c_int64(), C.c_uint64()
        _lib.check(self.ctx.lib.ilsx_sac_get_alpha_opt(self.h, C.byref(m), C.byref(v), C.byref(t), C.byref(r)))
        snap["alpha_optimizer"] = dict(exp_avg=m.value, exp_avg_sq=v.value, step=t.value, rng_step=r.value)
        return snap

    def load_snapshot(self, snap):  # sac_alpha.py:263-273
        for k in self.WHICH:
            self.set_params(k, snap[k])
        self.log_alpha = snap["log_alpha"]
        for k, w in (("policy", 0), ("qf1", 1), ("qf2", 2)):
            o = snap[k + "_optimizer"]
            m, v = np.ascontiguousarray(o["exp_avg"], np.float32), np.ascontiguousarray(o["exp_avg_sq"], np.float32)
            _lib.check(self.ctx.lib.ilsx_sac_set_adam(self.h, w, m.ctypes.data_as(C.c_void_p),
                                                      v.ctypes.data_as(C.c_void_p), m.size, int(o["step"])))
        o = snap["alpha_optimizer"]
        _lib.check(self.ctx.lib.ilsx_sac_set_alpha_opt(self.h, o["exp_avg"], o["exp_avg_sq"], int(o["step"]),
                                                       int(o["rng_step"])))

    # ---- parity aids (tests): the inputs of a fused step, rebuilt by independent kernels
    @property
    def rng_step(self):
        """Philox counter of the NEXT gradient step (== gradient steps taken so far for a fresh agent)."""
        r = C.c_uint64()
        _lib.check(self.ctx.lib.ilsx_sac_get_alpha_opt(self.h, None, None, None, C.byref(r)))
        return r.value

    def debug_batch(self, replay_buffer, step, batch_size):
        """Batch dict + (eps_next, eps_cur, idx) that gradient step number `step` of train_from_replay(replay_buffer, ...) uses."""
        ctx, B, o, a = self.ctx, int(batch_size), self.policy.obs_dim, self.policy.action_dim
        bufs = [ctx.empty((B, o)), ctx.empty((B, a)), ctx.empty((B,)), ctx.empty((B,)), ctx.empty((B, o)),
                ctx.empty((B, a)), ctx.empty((B, a)), ctx.empty((B,), np.int64)]
        _lib.check(ctx.lib.ilsx_sac_debug_batch(self.h, replay_buffer.h, C.c_uint64(int(step)), B, *[b.ptr for b in bufs]))
        obs, act, rew, done, nobs, e1, e2, idx = [b.numpy() for b in bufs]
        batch = dict(observations=obs, actions=act, rewards=rew.reshape(B, 1), terminals=done.reshape(B, 1),
                     next_observations=nobs)
        return batch, e1, e2, idx

    def debug_last_batch(self, batch_size):
        """Rows the last fused step gathered (as published by its first launch) and the eps_cur its policy head consumed."""
        ctx, B, o, a = self.ctx, int(batch_size), self.policy.obs_dim, self.policy.action_dim
        bufs = [ctx.empty((B, o)), ctx.empty((B, a)), ctx.empty((B,)), ctx.empty((B,)), ctx.empty((B, o)), ctx.empty((B, a))]
        _lib.check(ctx.lib.ilsx_sac_debug_last_batch(self.h, B, *[b.ptr for b in bufs]))
        obs, act, rew, done, nobs, e2 = [b.numpy() for b in bufs]
        return dict(observations=obs, actions=act, rewards=rew.reshape(B, 1), terminals=done.reshape(B, 1),
                    next_observations=nobs), e2

    # ---- split-run phases (multi-GPU)
    def set_batch(self, batch, eps_next=None, eps_cur=None):
        ctx, keep = self.ctx, []

        def dev(x):
            k, p = as_dev(ctx, x)
            keep.append(k)
            return p
        B = int(batch["observations"].shape[0])
        flat = lambda v: v.reshape(B)  # noqa: E731
        _lib.check(ctx.lib.ilsx_sac_set_batch(
            self.h, dev(batch["observations"]), dev(batch["actions"]), dev(flat(batch["rewards"])),
            dev(flat(batch["terminals"])), dev(batch["next_observations"]), B,
            dev(eps_next) if eps_next is not None else None, dev(eps_cur) if eps_cur is not None else None))
        ctx.sync()

    def critic_backward(self):
        _lib.check(self.ctx.lib.ilsx_sac_critic_backward(self.h))

    def critic_update(self):
        _lib.check(self.ctx.lib.ilsx_sac_critic_update(self.h))

    def actor_backward(self):
        _lib.check(self.ctx.lib.ilsx_sac_actor_backward(self.h))

    def actor_update(self):
        _lib.check(self.ctx.lib.ilsx_sac_actor_update(self.h))


class SoftActorCriticGroup:
    """K independent SoftActorCritic runs (seeds) of identical shape stepped in lock-step on one GPU, every stage of the
    step being ONE launch for all of them (ilsx_sac_group).  The trainers stay ordinary objects."""

    def __init__(self, trainers, ctx=None):
        # ctx: the context whose stream the grouped launches go on; the trainers live in it or in siblings of it (Context.sibling)
        self.trainers, self.ctx = list(trainers), ctx or trainers[0].ctx
        arr = (C.c_void_p * len(self.trainers))(*[t.h for t in self.trainers])
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_sac_group_create(self.ctx.h, arr, len(self.trainers), C.byref(self.h)))

    def train_from_replay(self, replay_buffers, n_steps, batch_size):
        """TorchRLAlgorithm._do_training of every run; the statistics (first batch of the call, sac_alpha.py:185-190) go to the trainers
        that have none since their last end_epoch."""
        want = [t for t in self.trainers if t.eval_statistics is None]
        arr = (C.c_void_p * len(self.trainers))(*[rb.h for rb in replay_buffers])
        _lib.check(self.ctx.lib.ilsx_sac_group_train_from_replay(self.h, arr, int(n_steps), int(batch_size), int(bool(want))))
        for t in want:
            _lib.check(self.ctx.lib.ilsx_sac_last_stats(t.h, C.byref(t._stats)))
            t._fill_stats()

    def close(self):
        if self.h:
            self.ctx.lib.ilsx_sac_group_destroy(self.h)
            self.h = None
