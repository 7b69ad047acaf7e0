"""HipVectorEnv — the reference's BaseVectorEnv protocol (rlkit/envs/vecenvs.py:63-369: `len(env)`,
`reset(id)`, `step(action, id)`, `seed`, list-valued `action_space` / `observation_space`) over the HIP
batched stepper of libilsx, plus the env factory `get_envs` (rlkit/envs/__init__.py:72-132).

Differences a caller can observe are listed in DESIGN.md: physics is this repo's planar engine, not
MuJoCo; env i is seeded `seed + i` through a counter-based generator instead of gym's np_random.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..device import as_dev, get_context
from .models import MODELS
from .models3d import MODELS3D


class Box:  # minimal gym.spaces.Box stand-in (gym is not installed here)
    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, np.float32), np.asarray(high, np.float32)
        self.shape, self.dtype = self.low.shape, np.float32

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(np.float32)


def model_struct(m):
    s = _lib.PlanarModel()
    s.task, s.n_body, s.n_geom, s.frame_skip, s.pgs_iters = m["task"], m["n_body"], m["n_geom"], m["frame_skip"], m["pgs_iters"]
    for b in range(m["n_body"]):
        s.parent[b], s.limited[b] = m["parent"][b], m["limited"][b]
        for k in (0, 1):
            s.anchor[b][k], s.com[b][k], s.range[b][k] = m["anchor"][b][k], m["com"][b][k], m["range"][b][k]
        s.mass[b], s.inertia[b], s.jsign[b] = m["mass"][b], m["inertia"][b], m["jsign"][b]
        s.armature[b], s.damping[b], s.gear[b] = m["armature"][b], m["damping"][b], m["gear"][b]
        s.stiffness[b] = m.get("stiffness", [0.0] * m["n_body"])[b]
    for g in range(m["n_geom"]):
        s.geom_body[g] = m["geom_body"][g]
        for k in (0, 1):
            s.geom_p1[g][k], s.geom_p2[g][k] = m["geom_p1"][g][k], m["geom_p2"][g][k]
        s.geom_radius[g], s.geom_friction[g] = m["geom_radius"][g], m["geom_friction"][g]
    s.timestep, s.gravity, s.reset_noise, s.contact_margin = m["timestep"], m["gravity"], m["reset_noise"], m["contact_margin"]
    for k in (0, 1):
        s.contact_solref[k], s.limit_solref[k] = m["contact_solref"][k], m["limit_solref"][k]
    for k in (0, 1, 2):
        s.contact_solimp[k], s.limit_solimp[k] = m["contact_solimp"][k], m["limit_solimp"][k]
    hl = m["healthy"]
    s.ctrl_cost, s.alive_bonus = m["ctrl_cost"], m["alive_bonus"]
    s.z_min, s.z_max, s.ang_max, s.state_max = hl["z_min"], hl["z_max"], hl["ang"], hl["state"]
    for i, v in enumerate(m["init_qpos"]):
        s.init_qpos[i] = v
    s.reset_noise_vel_std, s.qvel_clip, s.max_rows = m.get("reset_noise_vel_std", 0.0), m.get("qvel_clip", 10.0), m.get("max_rows", 0)
    return s


def spatial_struct(m):
    """models3d dict -> ilsx_spatial_model (include/ilsx.h)."""
    s = _lib.SpatialModel()
    s.task, s.n_link, s.n_act, s.n_contact, s.n_body = m["task"], m["n_link"], m["act_dim"], m["n_contact"], len(m["body_link"])
    s.frame_skip, s.pgs_iters, s.max_rows = m["frame_skip"], m["pgs_iters"], m["max_rows"]
    inertia = np.asarray(m["inertia"], np.float64)
    for l in range(m["n_link"]):
        s.parent[l], s.limited[l] = int(m["parent"][l]), int(m["limited"][l])
        for k in range(3):
            s.anchor[l][k], s.axis[l][k], s.com[l][k] = m["anchor"][l][k], m["axis"][l][k], m["com"][l][k]
        for k in range(4):
            s.quat0[l][k] = m["quat0"][l][k]
        I = inertia[l]
        for k, (i, j) in enumerate(((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))):
            s.inertia[l][k] = I[i, j]
        s.mass[l], s.armature[l], s.damping[l], s.stiffness[l] = m["mass"][l], m["armature"][l], m["damping"][l], m["stiffness"][l]
        s.range[l][0], s.range[l][1], s.gear[l] = m["range"][l][0], m["range"][l][1], m["gear"][l]
    for k, l in enumerate(m["act_links"]):
        s.act_link[k] = int(l)
    for c in range(m["n_contact"]):
        s.contact_link[c], s.contact_radius[c], s.contact_friction[c] = int(m["contact_link"][c]), m["contact_radius"][c], m["contact_friction"][c]
        for k in range(3):
            s.contact_pos[c][k] = m["contact_pos"][c][k]
    for b, l in enumerate(m["body_link"]):
        s.body_link[b] = int(l)
    s.timestep, s.gravity, s.reset_noise, s.reset_noise_vel_std = m["timestep"], m["gravity"], m["reset_noise"], m["reset_noise_vel_std"]
    s.contact_margin, s.ctrl_range = m["contact_margin"], m["ctrl_range"]
    for k in (0, 1):
        s.contact_solref[k], s.limit_solref[k] = m["contact_solref"][k], m["limit_solref"][k]
    for k in (0, 1, 2):
        s.contact_solimp[k], s.limit_solimp[k] = m["contact_solimp"][k], m["limit_solimp"][k]
    s.ctrl_cost, s.alive_bonus, s.vel_weight, s.z_min, s.z_max = m["ctrl_cost"], m["alive_bonus"], m["vel_weight"], m["z_min"], m["z_max"]
    for i, v in enumerate(m["init_qpos"]):
        s.init_qpos[i] = v
    return s


EPS = np.finfo(np.float32).eps.item()   # rlkit/envs/wrappers.py:9


class ProxyEnv:
    """wrappers.py ProxyEnv: the identity wrapper the run scripts pass by default."""
    shift = scale = None

    def __init__(self, **kwargs):
        pass


class ScaledEnv(ProxyEnv):
    """wrappers.py:53-131: obs -> (obs - obs_mean) / (obs_std + EPS).  (The action un-scaling half is unused by every spec:
    acts_mean / acts_std are None in the run scripts.)"""

    def __init__(self, obs_mean=None, obs_std=None, acts_mean=None, acts_std=None, **kwargs):
        if acts_mean is not None or acts_std is not None:
            raise NotImplementedError("action un-scaling is never enabled by the reference's scripts (adv_irl_exp_script.py:59)")
        if obs_mean is not None:
            self.shift, self.scale = np.asarray(obs_mean, np.float64), np.asarray(obs_std, np.float64) + EPS


class MinmaxEnv(ProxyEnv):
    """wrappers.py:134-203: obs -> (obs - obs_min) / (obs_max - obs_min + EPS)."""

    def __init__(self, obs_min=None, obs_max=None, **kwargs):
        if obs_min is not None:
            self.shift = np.asarray(obs_min, np.float64)
            self.scale = np.asarray(obs_max, np.float64) - self.shift + EPS


class DeviceObsRms:
    """`env.obs_rms` (RunningMeanStd, normalizer.py:128-152) living on the device inside a HipVectorEnv: float64
    mean / var / count, read back on attribute access.  Passing one env's obs_rms to another env's constructor
    (ppo_exp_script.py:68-75) copies the statistics at every `sync_from()`."""

    def __init__(self, env):
        self.env = env

    def _get(self):
        e = self.env
        m, v, c = np.empty(e.obs_dim), np.empty(e.obs_dim), C.c_double()
        _lib.check(e.ctx.lib.ilsx_vecenv_get_obs_rms(e.h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.byref(c)))
        return m, v, c.value

    mean = property(lambda self: self._get()[0])
    var = property(lambda self: self._get()[1])
    count = property(lambda self: self._get()[2])

    def set(self, mean, var, count):
        e = self.env
        m, v = np.ascontiguousarray(mean, np.float64), np.ascontiguousarray(var, np.float64)
        _lib.check(e.ctx.lib.ilsx_vecenv_set_obs_rms(e.h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), float(count)))


class HipVectorEnv:
    def __init__(self, env_name, env_num, seed=0, ctx=None, model=None, norm_obs=False, obs_rms=None, update_obs_rms=True,
                 obs_shift=None, obs_scale=None):
        self.ctx = ctx or get_context()
        self.model = model or (MODELS3D[env_name]() if env_name in MODELS3D else MODELS[env_name]())
        self.env_num = int(env_num)
        self.h = C.c_void_p()
        if "n_link" in self.model:      # 3-D engine (Ant / Humanoid)
            ms = spatial_struct(self.model)
            _lib.check(self.ctx.lib.ilsx_vecenv_create_spatial(self.ctx.h, C.byref(ms), self.env_num, C.c_uint64(seed), C.byref(self.h)))
        else:
            ms = model_struct(self.model)
            _lib.check(self.ctx.lib.ilsx_vecenv_create(self.ctx.h, C.byref(ms), self.env_num, C.c_uint64(seed), C.byref(self.h)))
        o, a, n, ne = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.ctx.lib.ilsx_vecenv_dims(self.h, C.byref(o), C.byref(a), C.byref(n), C.byref(ne)))
        self.obs_dim, self.act_dim, self.n_dof = o.value, a.value, n.value
        _lib.check(self.ctx.lib.ilsx_vecenv_state_dims(self.h, C.byref(o), C.byref(a)))
        self.nq, self.nv = o.value, a.value
        ob = Box(-np.inf * np.ones(self.obs_dim), np.inf * np.ones(self.obs_dim))
        ac = Box(-np.ones(self.act_dim), np.ones(self.act_dim))
        self.observation_space, self.action_space = [ob] * self.env_num, [ac] * self.env_num  # vecenvs.py:118-140
        self.single_observation_space, self.single_action_space = ob, ac
        self.obs_shift, self.obs_scale = obs_shift, obs_scale
        if obs_shift is not None:   # ScaledEnv / MinmaxEnv folded into the stepper
            sh, sc = np.ascontiguousarray(obs_shift, np.float64), np.ascontiguousarray(obs_scale, np.float64)
            assert sh.shape == (self.obs_dim,) and sc.shape == (self.obs_dim,)
            _lib.check(self.ctx.lib.ilsx_vecenv_set_obs_affine(self.h, sh.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))
        # vecenvs.py:104-113
        self.norm_obs, self.update_obs_rms = bool(norm_obs), bool(update_obs_rms) and bool(norm_obs)
        self.obs_rms = DeviceObsRms(self) if norm_obs else None
        self._shared_rms = obs_rms if norm_obs else None    # another env's statistics (eval env of ppo_exp_script.py:68-75)
        if norm_obs:
            if self._shared_rms is not None:
                self.sync_obs_rms()
            _lib.check(self.ctx.lib.ilsx_vecenv_obs_norm(self.h, 1, int(self.update_obs_rms)))

    def sync_obs_rms(self):
        """Pull the statistics of the env this one shares its obs_rms with (same object in the reference)."""
        if self._shared_rms is not None:
            self.obs_rms.set(*self._shared_rms._get())

    def get_scaled_obs(self, obs):      # wrappers.py:101-105,176-180
        return obs if self.obs_shift is None else (obs - self.obs_shift) / self.obs_scale

    def get_unscaled_obs(self, obs):    # wrappers.py:95-99,170-174
        return obs if self.obs_shift is None else obs * self.obs_scale + self.obs_shift

    def normalize_obs(self, obs):  # vecenvs.py:299-327
        if not self.norm_obs:
            return obs
        m, v, _ = self.obs_rms._get()
        return np.clip((obs - m) / np.sqrt(v + np.finfo(np.float32).eps.item()), -10.0, 10.0)

    def unnormalize_obs(self, obs):  # vecenvs.py:329-349
        if not self.norm_obs:
            return obs
        m, v, _ = self.obs_rms._get()
        return obs * np.sqrt(v + np.finfo(np.float32).eps.item()) + m

    def __len__(self):
        return self.env_num

    def single_env_view(self):
        """What the run scripts pass as `env` to trainers / EnvReplayBuffer: spaces of ONE env."""
        return type("EnvView", (), dict(observation_space=self.single_observation_space,
                                        action_space=self.single_action_space))()

    def seed(self, seed=None):  # vecenvs.py:259-277 (seed + i per env is the Philox key + env index here)
        return [seed] * self.env_num

    def _ids(self, id):
        if id is None:
            return None, self.env_num
        ids = np.ascontiguousarray(np.atleast_1d(id), np.int32)
        return ids, ids.size

    def reset(self, id=None):
        ids, n = self._ids(id)
        obs = self.ctx.empty((n, self.obs_dim))
        _lib.check(self.ctx.lib.ilsx_vecenv_reset(self.h, ids.ctypes.data_as(C.c_void_p) if ids is not None else None, n, obs.ptr))
        return obs.numpy().astype(np.float64)

    def step(self, action, id=None):
        ids, n = self._ids(id)
        action = np.ascontiguousarray(action, np.float32).reshape(n, self.act_dim)  # sync mode: len(action) == len(id)
        k, pa = as_dev(self.ctx, action)
        obs, rew, done = self.ctx.empty((n, self.obs_dim)), self.ctx.empty((n,)), self.ctx.empty((n,), np.uint8)
        _lib.check(self.ctx.lib.ilsx_vecenv_step(self.h, pa, ids.ctypes.data_as(C.c_void_p) if ids is not None else None, n,
                                                 obs.ptr, rew.ptr, done.ptr))
        env_ids = ids if ids is not None else np.arange(n)
        infos = [{"env_id": int(i)} for i in env_ids]  # vecenvs.py:217-219
        return obs.numpy().astype(np.float64), rew.numpy().astype(np.float64), done.numpy().astype(bool), infos

    # ---- simulator state (tests, snapshots)
    def get_state(self):
        q = np.empty((self.env_num, self.nq)); v = np.empty((self.env_num, self.nv))
        _lib.check(self.ctx.lib.ilsx_vecenv_get_state(self.h, q.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
        return q, v

    def set_state(self, qpos, qvel):
        q, v = np.ascontiguousarray(qpos, np.float64), np.ascontiguousarray(qvel, np.float64)
        assert q.shape == (self.env_num, self.nq) and v.shape == (self.env_num, self.nv)
        _lib.check(self.ctx.lib.ilsx_vecenv_set_state(self.h, q.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))

    # ---- fused device loop (BaseAlgorithm's sampling iteration, base_algorithm.py:183-277)
    def rollout_step(self, policy=None, replay=None, max_path_length=1000, random_actions=False, deterministic=False,
                     no_terminal=False, label_policy=None, begin_only=False):
        """begin_only: only enqueue the step (ilsx_rollout_step_begin); the caller finishes it with rollout_step_end() — lets a lock-step loop
        over several runs overlap their launches (DeviceRLAlgorithmGroup)."""
        if label_policy is not None:      # DAgger: store the expert's action (dagger.py:45-71)
            det = hasattr(label_policy, "stochastic_policy")
            lp = label_policy.stochastic_policy if det else label_policy
            _lib.check(self.ctx.lib.ilsx_rollout_step_relabel(self.h, policy.h, lp.h, int(det), replay.h if replay is not None else None,
                                                              int(max_path_length), int(bool(no_terminal))))
            return
        fn = self.ctx.lib.ilsx_rollout_step_begin if begin_only else self.ctx.lib.ilsx_rollout_step
        _lib.check(fn(self.h, policy.h if policy is not None else None, replay.h if replay is not None else None, int(max_path_length),
                      int(bool(random_actions)), int(bool(deterministic)), int(bool(no_terminal))))

    def rollout_step_end(self):
        _lib.check(self.ctx.lib.ilsx_rollout_step_end(self.h))

    def set_path_mode(self, on=True):
        """Fused rollouts insert whole episodes when they end, contiguously and registered in `_traj_endpoints` (the reference's
        order, base_algorithm.py:509-519) instead of every transition as it happens."""
        _lib.check(self.ctx.lib.ilsx_vecenv_set_path_mode(self.h, int(bool(on))))

    def rollout_stats(self, reset=True):
        e, r = C.c_double(), C.c_double()
        _lib.check(self.ctx.lib.ilsx_rollout_stats(self.h, C.byref(e), C.byref(r), int(reset)))
        return e.value, r.value

    def close(self):
        if self.h:
            self.ctx.lib.ilsx_vecenv_destroy(self.h)
            self.h = None


def get_envs(env_specs, env_wrapper=None, wrapper_kwargs=None, ctx=None, norm_obs=False, obs_rms=None, update_obs_rms=True,
             **kwargs):
    """rlkit/envs/__init__.py:72-132: env_specs{env_name, env_num, training_env_seed, ...} -> vec env.
    NormalizedBoxEnv is folded into the stepper; env_wrapper = ProxyEnv / ScaledEnv / MinmaxEnv with wrapper_kwargs as in
    adv_irl_exp_script.py:79-132; norm_obs / obs_rms / update_obs_rms are BaseVectorEnv's (vecenvs.py:84-113)."""
    if env_specs.get("use_envpool"):   # rlkit/envs/__init__.py:78-84 (wrappers do not apply on this branch there either)
        from .envpool import EnvpoolEnv
        return EnvpoolEnv(env_specs, ctx=ctx, norm_obs=norm_obs, obs_rms=obs_rms, update_obs_rms=update_obs_rms)
    w = env_wrapper(**(wrapper_kwargs or {})) if env_wrapper is not None else ProxyEnv()
    return HipVectorEnv(env_specs["env_name"], env_specs.get("env_num", 1),
                        seed=env_specs.get("training_env_seed", env_specs.get("seed", 0)), ctx=ctx, norm_obs=norm_obs,
                        obs_rms=obs_rms, update_obs_rms=update_obs_rms, obs_shift=w.shift, obs_scale=w.scale)
