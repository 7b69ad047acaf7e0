"""Adversarial IRL (GAIL / AIRL / FAIRL / "gail2") over libilsx: the reference's `MLPDisc`
(rlkit/torch/algorithms/adv_irl/disc_models/simple_disc_models.py:8-48) and the two inner steps of `AdvIRL`
(rlkit/torch/algorithms/adv_irl/adv_irl.py:126-131 loop, :133-216 discriminator step, :238-314 reward
relabel + policy step).  Constructor kwargs are the YAML keys of exp_specs/gail/gail_walker.yaml.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .device import as_dev, get_context

_MODES = dict(airl=0, gail=1, gail2=2, fairl=3)
_ACT = dict(relu=0, tanh=1)


class MLPDisc:
    """MLPDisc(input_dim, num_layer_blocks=2, hid_dim, hid_act, use_bn=False, clamp_magnitude) + its optimiser
    (torch.optim.Adam(lr=disc_lr, betas=(disc_momentum, 0.999)), adv_irl.py:75-77)."""

    def __init__(self, obs_dim, act_dim, num_layer_blocks=2, hid_dim=128, hid_act="tanh", use_bn=False,
                 clamp_magnitude=10.0, disc_lr=3e-4, disc_momentum=0.9, use_grad_pen=True, grad_pen_weight=10.0,
                 max_batch=256, ctx=None, seed=None):
        if use_bn or num_layer_blocks != 2:
            raise NotImplementedError("hot-path configs use 2 blocks without batch norm (gail_walker.yaml:24-28)")
        self.ctx = ctx or get_context()
        self.obs_dim, self.act_dim, self.hid_dim = int(obs_dim), int(act_dim), int(hid_dim)
        self.clamp_magnitude, self.grad_pen_weight, self.use_grad_pen = clamp_magnitude, grad_pen_weight, use_grad_pen
        cfg = _lib.DiscCfg(self.obs_dim, self.act_dim, self.hid_dim, _ACT[hid_act], int(bool(use_grad_pen)),
                           clamp_magnitude, disc_lr, disc_momentum, grad_pen_weight, int(max_batch))
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_disc_create(self.ctx.h, C.byref(cfg), C.byref(self.h)))
        n = C.c_size_t()
        _lib.check(self.ctx.lib.ilsx_disc_num_params(self.h, C.byref(n)))
        self.num_params = n.value
        # torch nn.Linear default init: W, b ~ U(+-1/sqrt(fan_in))
        rng = np.random.default_rng(np.random.randint(0, 2**31 - 1) if seed is None else seed)
        D, H = self.obs_dim + self.act_dim, self.hid_dim
        parts = []
        for fan_in, out in ((D, H), (H, H), (H, 1)):
            b = 1.0 / np.sqrt(fan_in)
            parts += [rng.uniform(-b, b, (out, fan_in)).ravel(), rng.uniform(-b, b, out)]
        self.set_flat_params(np.concatenate(parts).astype(np.float32))
        self._stats = _lib.DiscStats()

    def set_flat_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        _lib.check(self.ctx.lib.ilsx_disc_set_params(self.h, flat.ctypes.data_as(C.c_void_p), flat.size))

    def get_flat_params(self):
        out = np.empty(self.num_params, np.float32)
        _lib.check(self.ctx.lib.ilsx_disc_get_params(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def get_flat_grads(self):
        out = np.empty(self.num_params, np.float32)
        _lib.check(self.ctx.lib.ilsx_disc_get_grads(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def train_step(self, expert_obs, expert_act, policy_obs, policy_act, eps=None):
        """AdvIRL._do_reward_training on explicit batches; returns the reference's statistics dict."""
        ctx, keep = self.ctx, []

        def dev(x):
            k, p = as_dev(ctx, x)
            keep.append(k)
            return p
        B = int(np.shape(expert_obs)[0])
        pe = dev(np.asarray(eps, np.float32).reshape(B)) if eps is not None else None
        _lib.check(ctx.lib.ilsx_disc_train_step(self.h, dev(expert_obs), dev(expert_act), dev(policy_obs), dev(policy_act),
                                                B, pe, C.byref(self._stats)))
        s = self._stats
        st = OrderedDict([("Disc CE Loss", s.ce_loss), ("Disc Acc", s.accuracy)])
        if self.use_grad_pen:
            st["Grad Pen"], st["Grad Pen W"] = s.grad_pen, self.grad_pen_weight
        return st

    def reward_dev(self, obs_ptr, act_ptr, n, mode, rew_clip_min=None, rew_clip_max=None, rew=None, logits=None):
        _lib.check(self.ctx.lib.ilsx_disc_reward(
            self.h, obs_ptr, act_ptr, n, _MODES[mode], int(rew_clip_min is not None), float(rew_clip_min or 0.0),
            int(rew_clip_max is not None), float(rew_clip_max or 0.0), rew.ptr if rew is not None else None,
            logits.ptr if logits is not None else None))

    def rewards(self, obs, act, mode="gail2", rew_clip_min=None, rew_clip_max=None):
        obs, act = np.ascontiguousarray(obs, np.float32), np.ascontiguousarray(act, np.float32)
        n = obs.shape[0]
        ko, po = as_dev(self.ctx, obs)
        ka, pa = as_dev(self.ctx, act)
        rew, lg = self.ctx.empty((n,)), self.ctx.empty((n,))
        self.reward_dev(po, pa, n, mode, rew_clip_min, rew_clip_max, rew, lg)
        return rew.numpy().reshape(n, 1), lg.numpy().reshape(n, 1)

    def __call__(self, x):  # clamped logits of cat(obs, act) rows
        x = np.ascontiguousarray(x, np.float32)
        return self.rewards(x[:, : self.obs_dim], x[:, self.obs_dim:], "airl")[1]


class AdvIRLTrainer:
    """`AdvIRL._do_training` (adv_irl.py:126-131): per update loop, k discriminator steps then m policy steps
    whose rewards are relabelled by the discriminator (adv_irl.py:256-301).  Batches are drawn on the device
    from the two HBM replay buffers; nothing crosses PCIe."""

    def __init__(self, mode, discriminator, policy_trainer, expert_replay_buffer, replay_buffer,
                 disc_optim_batch_size=256, policy_optim_batch_size=256, num_update_loops_per_train_call=1,
                 num_disc_updates_per_loop_iter=1, num_policy_updates_per_loop_iter=1, rew_clip_min=None,
                 rew_clip_max=None, state_only=False, **kwargs):
        assert mode in _MODES, "Invalid adversarial irl algorithm!"
        if state_only:
            raise NotImplementedError("state_only discriminators are not on the hot path (gail_walker.yaml)")
        self.mode, self.disc, self.policy_trainer = mode, discriminator, policy_trainer
        self.expert_rb, self.rb = expert_replay_buffer, replay_buffer
        self.Bd, self.Bp = int(disc_optim_batch_size), int(policy_optim_batch_size)
        self.loops, self.k, self.m = num_update_loops_per_train_call, num_disc_updates_per_loop_iter, num_policy_updates_per_loop_iter
        self.rew_clip_min, self.rew_clip_max = rew_clip_min, rew_clip_max
        ctx = self.ctx = discriminator.ctx
        o, a, B = discriminator.obs_dim, discriminator.act_dim, max(self.Bd, self.Bp)
        mk = lambda: [ctx.empty((B, o)), ctx.empty((B, a)), ctx.empty((B,)), ctx.empty((B,)), ctx.empty((B, o))]  # noqa
        self._e, self._p = mk(), mk()
        self.disc_eval_statistics = None

    def _sample(self, rb, bufs, B):
        _lib.check(self.ctx.lib.ilsx_replay_sample(rb.h, B, None, *[b.ptr for b in bufs], None))

    def _do_reward_training(self):
        self._sample(self.expert_rb, self._e, self.Bd)
        self._sample(self.rb, self._p, self.Bd)
        want = self.disc_eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_disc_train_step(self.disc.h, self._e[0].ptr, self._e[1].ptr, self._p[0].ptr,
                                                     self._p[1].ptr, self.Bd, None,
                                                     C.byref(self.disc._stats) if want else None))
        if want:
            s = self.disc._stats
            self.disc_eval_statistics = OrderedDict([("Disc CE Loss", s.ce_loss), ("Disc Acc", s.accuracy),
                                                     ("Grad Pen", s.grad_pen), ("Grad Pen W", self.disc.grad_pen_weight)])

    def _do_policy_training(self):
        obs, act, rew, done, nobs = self._p
        self._sample(self.rb, self._p, self.Bp)
        self.disc.reward_dev(obs.ptr, act.ptr, self.Bp, self.mode, self.rew_clip_min, self.rew_clip_max, rew=rew)
        tr = self.policy_trainer
        want = tr.eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_sac_train_step(tr.h, obs.ptr, act.ptr, rew.ptr, done.ptr, nobs.ptr, self.Bp, None, None,
                                                    C.byref(tr._stats) if want else None))
        if want:
            tr._fill_stats()
            r = rew.numpy()[: self.Bp]
            self.disc_eval_statistics.update({"Disc Rew Mean": float(r.mean()), "Disc Rew Std": float(r.std()),
                                              "Disc Rew Max": float(r.max()), "Disc Rew Min": float(r.min())})

    def train(self, n_calls=1):
        """n_calls x `num_update_loops_per_train_call` loop iterations in ONE library call (ilsx_advirl_train)."""
        want_d, tr = self.disc_eval_statistics is None, self.policy_trainer
        want_p = tr.eval_statistics is None
        rs = (C.c_float * 4)()
        _lib.check(self.ctx.lib.ilsx_advirl_train(
            self.disc.h, tr.h, self.expert_rb.h, self.rb.h, int(n_calls * self.loops), int(self.k), int(self.m), self.Bd, self.Bp,
            _MODES[self.mode], int(self.rew_clip_min is not None), float(self.rew_clip_min or 0.0),
            int(self.rew_clip_max is not None), float(self.rew_clip_max or 0.0),
            C.byref(self.disc._stats) if want_d else None, C.byref(tr._stats) if want_p else None, rs if want_d else None))
        if want_d:
            s = self.disc._stats
            self.disc_eval_statistics = OrderedDict([("Disc CE Loss", s.ce_loss), ("Disc Acc", s.accuracy)])
            if self.disc.use_grad_pen:
                self.disc_eval_statistics.update({"Grad Pen": s.grad_pen, "Grad Pen W": self.disc.grad_pen_weight})
            self.disc_eval_statistics.update({"Disc Rew Mean": rs[0], "Disc Rew Std": rs[1], "Disc Rew Max": rs[2],
                                              "Disc Rew Min": rs[3]})
        if want_p:
            tr._fill_stats()

    # ---- what DeviceRLAlgorithm asks of a trainer
    def train_from_replay(self, replay_buffer, n_loops, batch_size):
        """One train call of AdvIRL (adv_irl.py:126-131): n_loops = num_update_loops_per_train_call."""
        self.rb, loops = replay_buffer, self.loops
        self.loops = int(n_loops)
        try:
            self.train(1)
        finally:
            self.loops = loops

    @property
    def policy(self):
        return self.policy_trainer.policy

    @property
    def networks(self):
        return self.policy_trainer.networks + [self.disc]

    def get_snapshot(self):  # adv_irl.py:316-326, as plain arrays (+ disc_optimizer's Adam state)
        from .snapshot import get_opt
        snap = dict(self.policy_trainer.get_snapshot())
        snap["disc"] = self.disc.get_flat_params()
        snap["disc_optimizer"] = get_opt(self.disc.ctx.lib, "disc", self.disc.h, snap["disc"].size)
        return snap

    def load_snapshot(self, snap):
        from .snapshot import set_opt
        self.policy_trainer.load_snapshot(snap)
        self.disc.set_flat_params(snap["disc"])
        if "disc_optimizer" in snap:
            set_opt(self.disc.ctx.lib, "disc", self.disc.h, snap["disc_optimizer"])

    def get_eval_statistics(self):
        st = OrderedDict()
        st.update(self.disc_eval_statistics or {})
        st.update(self.policy_trainer.get_eval_statistics() or {})
        return st

    def end_epoch(self):
        self.policy_trainer.end_epoch()
        self.disc_eval_statistics = None
