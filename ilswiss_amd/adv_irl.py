"""Adversarial IRL (GAIL / AIRL / FAIRL / "gail2") over libilsx: the reference's `MLPDisc`
(rlkit/torch/algorithms/adv_irl/disc_models/simple_disc_models.py:8-48) and the two inner steps of `AdvIRL`
(rlkit/torch/algorithms/adv_irl/adv_irl.py:126-131 loop, :133-216 discriminator step, :238-314 reward
relabel + policy step).  Constructor signatures and DEFAULTS are the reference's (simple_disc_models.py:9-17,
adv_irl.py:34-54): a caller that relies on defaults gets the reference's algorithm or a loud error, never a
different network.
  * `use_bn=True` (the reference default) runs: Linear -> BatchNorm1d -> act blocks in train mode inside the discriminator step (both
    forwards on their own batch statistics, the gradient penalty's double backward THROUGH those statistics, running-statistics
    updates) and in eval mode for the policy's rewards (adv_irl.py:268-274) — a chain of simple launches (csrc/disc_bn.h), pinned by the
    reference-generated g26; every exp_spec of the reference sets `disc_use_bn: false`, which keeps the fused kernels;
  * `num_layer_blocks` outside 1..3 raises (every exp_spec sets 2, which runs the fused kernel; 1 and 3 run the same mathematics as a
    chain of per-layer launches).
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from .device import as_dev, get_context

_MODES = dict(airl=0, gail=1, gail2=2, fairl=3)
_ACT = dict(relu=0, tanh=1)
_WIDTHS = (64, 128, 256)   # hidden widths the MFMA kernels are instantiated for


class MLPDisc:
    """MLPDisc(input_dim, num_layer_blocks=2, hid_dim=100, hid_act='relu', use_bn=True, clamp_magnitude=10.0)
    (simple_disc_models.py:9-17).  The optimiser (Adam(lr=disc_lr, betas=(disc_momentum, 0.999)), adv_irl.py:75-77) and the
    gradient-penalty settings belong to AdvIRL in the reference; here they reach the library through `bind`, which AdvIRLTrainer
    calls with its own kwargs (standalone use: call `bind` yourself).  Until then the parameters live on the host.

    A `hid_dim` the kernels have no instantiation for (the reference default 100) is zero-padded to the next supported width:
    padded units have pre-activation 0, output 0 (tanh / relu), receive gradient 0 and are never moved by Adam, so the padded
    network IS the hid_dim-wide one; flat parameter vectors cross this class in the logical (unpadded) layout."""

    def __init__(self, input_dim, num_layer_blocks=2, hid_dim=100, hid_act="relu", use_bn=True, clamp_magnitude=10.0, *,
                 ctx=None, seed=None):
        # use_bn=True (the reference's default, simple_disc_models.py:15): Linear -> BatchNorm1d -> act blocks, train mode in the discriminator
        # step (batch statistics, the gradient penalty's double backward through them), eval mode for the policy's rewards — a chain of
        # simple launches in the library (csrc/disc_bn.h), any hid_dim, no padding; use_bn=False (every exp_spec) keeps the fused kernels
        self.use_bn = bool(use_bn)
        if num_layer_blocks not in (1, 2, 3):
            raise NotImplementedError("libilsx implements num_layer_blocks 1..3 (gail_walker.yaml:24 uses 2); got %r" % (num_layer_blocks,))
        self.num_layer_blocks = int(num_layer_blocks)
        if hid_act not in _ACT:
            raise NotImplementedError()   # simple_disc_models.py:24-25
        if hid_dim > _WIDTHS[-1] and not self.use_bn:
            raise NotImplementedError("hid_dim > 256")
        self.ctx = ctx or get_context()
        self.input_dim, self.hid_dim, self.hid_act = int(input_dim), int(hid_dim), hid_act
        self._Hp = self.hid_dim if self.use_bn else next(w for w in _WIDTHS if w >= self.hid_dim)
        self.clamp_magnitude = clamp_magnitude
        # torch nn.Linear default init: W, b ~ U(+-1/sqrt(fan_in))
        rng = np.random.default_rng(np.random.randint(0, 2**31 - 1) if seed is None else seed)
        D, H = self.input_dim, self.hid_dim
        parts = []
        layers = [(D, H)] + [(H, H)] * (self.num_layer_blocks - 1) + [(H, 1)]
        for li, (fan_in, out) in enumerate(layers):
            b = 1.0 / np.sqrt(fan_in)
            parts += [rng.uniform(-b, b, (out, fan_in)).ravel(), rng.uniform(-b, b, out)]
            if self.use_bn and li < len(layers) - 1:   # every block, never the output layer (by index: hid_dim == 1 has out == H there too); nn.BatchNorm1d: weight (gamma) = 1, bias (beta) = 0 — parameters() lists them after the Linear's
                parts += [np.ones(H), np.zeros(H)]
        self._flat = np.concatenate(parts).astype(np.float32)
        self.num_params = self._flat.size
        self.h, self._bound = None, None
        self._stats = _lib.DiscStats()
        self.obs_dim = self.act_dim = None
        self.use_grad_pen, self.grad_pen_weight = True, 10.0

    # ---- logical <-> padded flat layouts (fc0.W | fc0.b | fc1.W | fc1.b | out.W | out.b)
    def _pad(self, flat):
        D, H, Hp = self.input_dim, self.hid_dim, self._Hp
        flat = np.ascontiguousarray(flat, np.float32)
        assert flat.size == self.num_params, (flat.size, self.num_params)
        if H == Hp:
            return flat
        o, out = 0, []
        for rows, cols, prow, pcol in self._shapes():
            m = np.zeros((prow, pcol), np.float32)
            m[:rows, :cols] = flat[o:o + rows * cols].reshape(rows, cols)
            out.append(m.ravel())
            o += rows * cols
        return np.concatenate(out)

    def _shapes(self):   # (rows, cols, padded rows, padded cols) of every tensor in flat order: W_0, b_0, ..., w_head, b_head
        D, H, Hp = self.input_dim, self.hid_dim, self._Hp
        sh = [(H, D, Hp, D), (H, 1, Hp, 1)]
        for _ in range(self.num_layer_blocks - 1):
            sh += [(H, H, Hp, Hp), (H, 1, Hp, 1)]
        return sh + [(1, H, 1, Hp), (1, 1, 1, 1)]

    def _unpad(self, phys):
        D, H, Hp = self.input_dim, self.hid_dim, self._Hp
        if H == Hp:
            return phys
        o, out = 0, []
        for rows, cols, prow, pcol in self._shapes():
            out.append(phys[o:o + prow * pcol].reshape(prow, pcol)[:rows, :cols].ravel())
            o += prow * pcol
        return np.concatenate(out)

    @property
    def _nphys(self):
        if self.use_bn:
            return self.num_params
        return sum(pr * pc for _, _, pr, pc in self._shapes())

    def get_bn_stats(self):
        """(running_mean, running_var) [num_layer_blocks, hid_dim] of a use_bn discriminator: module buffers, saved beside the parameters"""
        self._need()
        n = self.num_layer_blocks * self.hid_dim
        rm, rv = np.empty(n, np.float32), np.empty(n, np.float32)
        _lib.check(self.ctx.lib.ilsx_disc_get_bn_stats(self.h, rm.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p), n))
        return rm.reshape(self.num_layer_blocks, self.hid_dim), rv.reshape(self.num_layer_blocks, self.hid_dim)

    def set_bn_stats(self, running_mean, running_var):
        self._need()
        rm, rv = np.ascontiguousarray(running_mean, np.float32).ravel(), np.ascontiguousarray(running_var, np.float32).ravel()
        _lib.check(self.ctx.lib.ilsx_disc_set_bn_stats(self.h, rm.ctypes.data_as(C.c_void_p), rv.ctypes.data_as(C.c_void_p), rm.size))

    def bind(self, obs_dim, second_dim=None, state_only=False, disc_lr=1e-3, disc_momentum=0.0, use_grad_pen=True,
             grad_pen_weight=10.0, max_batch=1024, grad_world=1):
        """Create (or re-create, keeping the parameters) the library object.  The discriminator input is cat(obs, act) —
        or cat(obs, next_obs) when state_only (adv_irl.py:140-162) — so obs_dim + second_dim == input_dim."""
        obs_dim = int(obs_dim)
        second_dim = self.input_dim - obs_dim if second_dim is None else int(second_dim)
        if obs_dim + second_dim != self.input_dim:
            raise ValueError(f"input_dim {self.input_dim} != {obs_dim} + {second_dim}")
        key = (obs_dim, second_dim, bool(state_only), float(disc_lr), float(disc_momentum), bool(use_grad_pen),
               float(grad_pen_weight), int(max_batch), int(grad_world))
        if self._bound == key:
            return self
        opt = bn = None
        if self.h is not None and self.use_bn:
            bn = self.get_bn_stats()
        if self.h is not None:   # re-bind (e.g. a forward on an unbound discriminator bound it with default settings): parameters AND the
            from .snapshot import get_opt   # optimiser's moments / step count move to the new library object
            self._flat = self.get_flat_params()
            opt = get_opt(self.ctx.lib, "disc", self.h, self._nphys)
            _lib.check(self.ctx.lib.ilsx_disc_destroy(self.h))
        cfg = _lib.DiscCfg(obs_dim, second_dim, self._Hp, _ACT[self.hid_act], int(bool(use_grad_pen)), self.clamp_magnitude,
                           disc_lr, disc_momentum, grad_pen_weight, int(max_batch), int(bool(state_only)), self.num_layer_blocks,
                           int(self.use_bn), int(grad_world))   # grad_world: ilsx_disc_cfg (one run split over G ranks)
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_disc_create(self.ctx.h, C.byref(cfg), C.byref(self.h)))
        self._bound = key
        self.obs_dim, self.act_dim = obs_dim, second_dim
        self.use_grad_pen, self.grad_pen_weight, self.state_only = bool(use_grad_pen), grad_pen_weight, bool(state_only)
        self.set_flat_params(self._flat)
        if opt is not None:
            from .snapshot import set_opt
            set_opt(self.ctx.lib, "disc", self.h, opt)
        if bn is not None:
            self.set_bn_stats(*bn)
        return self

    def _need(self):
        if self.h is None:
            raise RuntimeError("MLPDisc is not bound to the library yet: construct the AdvIRLTrainer around it or call "
                               "disc.bind(obs_dim, ...) (the optimiser settings live on AdvIRL, adv_irl.py:75-77)")

    def set_flat_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        if self.h is None:
            assert flat.size == self.num_params
            self._flat = flat.copy()
            return
        phys = self._pad(flat)
        _lib.check(self.ctx.lib.ilsx_disc_set_params(self.h, phys.ctypes.data_as(C.c_void_p), phys.size))

    def _get(self, fn):
        out = np.empty(self._nphys, np.float32)
        _lib.check(fn(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return self._unpad(out)

    def get_flat_params(self):
        if self.h is None:
            return self._flat.copy()
        return self._get(self.ctx.lib.ilsx_disc_get_params)

    def get_flat_grads(self):
        self._need()
        return self._get(self.ctx.lib.ilsx_disc_get_grads)

    def train_step(self, expert_obs, expert_act, policy_obs, policy_act, eps=None):
        """AdvIRL._do_reward_training on explicit batches (the second array of each pair holds next_obs when state_only);
        returns the reference's statistics dict."""
        self._need()
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
        self._need()
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

    def __call__(self, x):  # MLPDisc.forward: clamped logits of disc-input rows (simple_disc_models.py:42-48); use_bn: the EVAL-mode forward
        # (running statistics), which is how the reference calls a trained discriminator outside its training step (adv_irl.py:268-274)
        if self.h is None:
            self.bind(self.input_dim - 1, 1)   # any split of the input will do for a forward
        x = np.ascontiguousarray(x, np.float32)
        return self.rewards(x[:, : self.obs_dim], x[:, self.obs_dim:], "airl")[1]

    forward = __call__


class AdvIRLTrainer:
    """`AdvIRL` (adv_irl.py:34-54 signature and defaults; the base-algorithm kwargs of the reference go to DeviceRLAlgorithm
    instead, `replay_buffer` is the one this class keeps): per update loop, num_disc_updates_per_loop_iter discriminator steps then
    num_policy_updates_per_loop_iter policy steps whose rewards are relabelled by the discriminator (adv_irl.py:126-131,256-301).
    Batches are drawn on the device from the two HBM replay buffers; nothing crosses PCIe."""

    def __init__(self, mode, discriminator, policy_trainer, expert_replay_buffer, state_only=False, disc_optim_batch_size=1024,
                 policy_optim_batch_size=1024, policy_optim_batch_size_from_expert=0, num_update_loops_per_train_call=1,
                 num_disc_updates_per_loop_iter=100, num_policy_updates_per_loop_iter=100, disc_lr=1e-3, disc_momentum=0.0,
                 disc_optimizer_class=None, use_grad_pen=True, grad_pen_weight=10, rew_clip_min=None, rew_clip_max=None,
                 replay_buffer=None, wrap_absorbing=False, grad_world=1, **kwargs):
        # grad_world (an ilswiss_amd key): this process is one rank of a run split over G GPUs — the batch sizes are this rank's rows, the
        # discriminator's and the policy trainer's gradient arenas are all-reduced by the library (include/ilsx.h ilsx_disc_cfg.grad_world)
        assert mode in _MODES, "Invalid adversarial irl algorithm!"
        if disc_optimizer_class is not None and getattr(disc_optimizer_class, "__name__", disc_optimizer_class) != "Adam":
            raise NotImplementedError("the discriminator optimiser is Adam (adv_irl.py:48)")
        if wrap_absorbing:
            raise NotImplementedError("wrap_absorbing discriminator inputs (adv_irl.py:152-170) are not implemented")
        self.mode, self.disc, self.policy_trainer = mode, discriminator, policy_trainer
        self.state_only = bool(state_only)
        self.expert_rb, self.rb = expert_replay_buffer, replay_buffer
        self.Bd, self.Bp = int(disc_optim_batch_size), int(policy_optim_batch_size)
        self.Bpe = int(policy_optim_batch_size_from_expert)
        if not 0 <= self.Bpe <= self.Bp:
            raise ValueError("policy_optim_batch_size_from_expert must be in 0..policy_optim_batch_size")
        self.loops, self.k, self.m = num_update_loops_per_train_call, num_disc_updates_per_loop_iter, num_policy_updates_per_loop_iter
        self.rew_clip_min, self.rew_clip_max = rew_clip_min, rew_clip_max
        ctx = self.ctx = discriminator.ctx
        o = int(expert_replay_buffer._observation_dim)
        a = int(expert_replay_buffer._action_dim)
        self.o, self.a = o, a
        discriminator.bind(o, o if self.state_only else a, state_only=self.state_only, disc_lr=disc_lr, disc_momentum=disc_momentum,
                           use_grad_pen=use_grad_pen, grad_pen_weight=grad_pen_weight, max_batch=max(self.Bd, self.Bp),
                           grad_world=int(grad_world))
        _lib.check(ctx.lib.ilsx_advirl_set_policy_batch_from_expert(discriminator.h, self.Bpe))
        B = max(self.Bd, self.Bp)
        mk = lambda: [ctx.empty((B, o)), ctx.empty((B, a)), ctx.empty((B,)), ctx.empty((B,)), ctx.empty((B, o))]  # noqa
        self._e, self._p = mk(), mk()
        self.disc_eval_statistics = None

    def _sample(self, rb, bufs, B, row0=0):
        ptrs = [C.c_void_p(b.ptr.value + 4 * row0 * (b.shape[1] if len(b.shape) > 1 else 1)) for b in bufs]
        _lib.check(self.ctx.lib.ilsx_replay_sample(rb.h, B, None, *ptrs, None))

    def _second(self, bufs):   # the discriminator's second input segment: actions, or next observations when state_only
        return bufs[4] if self.state_only else bufs[1]

    def _do_reward_training(self):
        self._sample(self.expert_rb, self._e, self.Bd)
        self._sample(self.rb, self._p, self.Bd)
        want = self.disc_eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_disc_train_step(self.disc.h, self._e[0].ptr, self._second(self._e).ptr, self._p[0].ptr,
                                                     self._second(self._p).ptr, self.Bd, None,
                                                     C.byref(self.disc._stats) if want else None))
        if want:
            s = self.disc._stats
            self.disc_eval_statistics = OrderedDict([("Disc CE Loss", s.ce_loss), ("Disc Acc", s.accuracy)])
            if self.disc.use_grad_pen:
                self.disc_eval_statistics.update({"Grad Pen": s.grad_pen, "Grad Pen W": self.disc.grad_pen_weight})

    def _do_policy_training(self):
        obs, act, rew, done, nobs = self._p
        npol = self.Bp - self.Bpe                      # adv_irl.py:239-255: cat([policy-buffer rows, expert-buffer rows])
        if npol > 0:
            self._sample(self.rb, self._p, npol)
        if self.Bpe > 0:
            self._sample(self.expert_rb, self._p, self.Bpe, row0=npol)
        self.disc.reward_dev(obs.ptr, self._second(self._p).ptr, self.Bp, self.mode, self.rew_clip_min, self.rew_clip_max, rew=rew)
        tr = self.policy_trainer
        want = tr.eval_statistics is None
        _lib.check(self.ctx.lib.ilsx_sac_train_step(tr.h, obs.ptr, act.ptr, rew.ptr, done.ptr, nobs.ptr, self.Bp, None, None,
                                                    C.byref(tr._stats) if want else None))
        self._last_rew = "host"     # the relabelled rewards of this (the latest) policy batch stay in self._p: get_eval_statistics reads them
        if want:
            tr._fill_stats()
            r = rew.numpy()[: self.Bp]
            if self.disc_eval_statistics is None:
                self.disc_eval_statistics = OrderedDict()
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
        if n_calls * self.loops > 0 and self.m > 0:
            self._last_rew = "staged"   # the last relabelled batch sits in the agent's batch arrays until the next call

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
    def networks(self):   # adv_irl.py:316-318
        return [self.disc] + self.policy_trainer.networks

    def get_snapshot(self):  # adv_irl.py:320-326, as plain arrays (+ disc_optimizer's Adam state, padded layout)
        from .snapshot import get_opt
        snap = dict(self.policy_trainer.get_snapshot())
        snap["disc"] = self.disc.get_flat_params()
        snap["disc_optimizer"] = get_opt(self.disc.ctx.lib, "disc", self.disc.h, self.disc._nphys)
        if self.disc.use_bn:   # BatchNorm buffers (state_dict entries of the reference's module, not parameters)
            snap["disc_bn_running_mean"], snap["disc_bn_running_var"] = self.disc.get_bn_stats()
        return snap

    def load_snapshot(self, snap):
        from .snapshot import set_opt
        self.policy_trainer.load_snapshot(snap)
        self.disc.set_flat_params(snap["disc"])
        if "disc_optimizer" in snap:
            set_opt(self.disc.ctx.lib, "disc", self.disc.h, snap["disc_optimizer"])
        if self.disc.use_bn and "disc_bn_running_mean" in snap:
            self.disc.set_bn_stats(snap["disc_bn_running_mean"], snap["disc_bn_running_var"])

    def _last_disc_rewards(self):
        """The relabelled rewards of the most recent policy batch (still on the device), or None before the first policy step."""
        src = getattr(self, "_last_rew", None)
        if src is None:
            return None
        if src == "host":
            return self._p[2].numpy()[: self.Bp]
        buf = self.ctx.empty((self.Bp,))
        _lib.check(self.ctx.lib.ilsx_sac_debug_last_batch(self.policy_trainer.h, self.Bp, None, None, buf.ptr, None, None, None))
        return buf.numpy()

    def get_eval_statistics(self):
        st = OrderedDict()
        st.update(self.disc_eval_statistics or {})
        # "Disc Rew *": the reference overwrites the four entries after EVERY policy step (adv_irl.py:303-314), so what an epoch logs are
        # the rewards of its LAST relabelled batch; the discriminator's own entries are those of the epoch's first batch (:205-216)
        r = self._last_disc_rewards() if st else None
        if r is not None and "Disc Rew Mean" in st:
            st.update({"Disc Rew Mean": float(r.mean()), "Disc Rew Std": float(r.std()), "Disc Rew Max": float(r.max()), "Disc Rew Min": float(r.min())})
        st.update(self.policy_trainer.get_eval_statistics() or {})
        return st

    def end_epoch(self):
        self.policy_trainer.end_epoch()
        self.disc_eval_statistics = None

    def to(self, device=None):   # adv_irl.py:328-331: everything already lives on the library's device
        return self


AdvIRL = AdvIRLTrainer   # the reference's class name
