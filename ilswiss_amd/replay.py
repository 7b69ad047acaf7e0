"""SimpleReplayBuffer / EnvReplayBuffer over the HBM-resident ring of libilsx
(rlkit/data_management/replay_buffer.py:4-83, simple_replay_buffer.py:17-442, env_replay_buffer.py:7-49).

The ring lives in device memory; this class keeps the reference's method names.  `random_batch`
draws its indices from the same `np.random.RandomState(random_seed).randint(0, size, B)` stream as the
reference (simple_replay_buffer.py:20,242) and hands them to the device gather, so the index stream
is bit-identical; the fused training path (SoftActorCritic.train_from_replay) samples on-device.
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import as_dev, get_context


class SimpleReplayBuffer:
    def __init__(self, max_replay_buffer_size, observation_dim, action_dim, random_seed=1995, ctx=None):
        if not isinstance(observation_dim, (int, np.integer)):
            raise NotImplementedError("dict / image observations are out of scope (SURVEY §2 #18)")
        self.ctx = ctx or get_context()
        self._np_rand_state = np.random.RandomState(random_seed)
        self._observation_dim, self._action_dim = int(observation_dim), int(action_dim)
        self._max_replay_buffer_size = int(max_replay_buffer_size)
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_replay_create(self.ctx.h, self._max_replay_buffer_size, self._observation_dim,
                                                   self._action_dim, C.c_uint64(random_seed), C.byref(self.h)))
        self._trajs = 0

    # ---- cursors (host mirror kept by the library)
    def _cursors(self):
        s, t = C.c_int64(), C.c_int64()
        _lib.check(self.ctx.lib.ilsx_replay_size(self.h, C.byref(s), C.byref(t)))
        return s.value, t.value

    @property
    def _size(self):
        return self._cursors()[0]

    @property
    def _top(self):
        return self._cursors()[1]

    @property
    def _traj_endpoints(self):
        n = C.c_int()
        _lib.check(self.ctx.lib.ilsx_replay_traj_endpoints(self.h, None, None, 0, C.byref(n)))
        st = np.zeros(max(n.value, 1), np.int64)
        en = np.zeros(max(n.value, 1), np.int64)
        _lib.check(self.ctx.lib.ilsx_replay_traj_endpoints(
            self.h, st.ctypes.data_as(_lib.c_i64p), en.ctypes.data_as(_lib.c_i64p), n.value, C.byref(n)))
        return {int(s): int(e) for s, e in zip(st[: n.value], en[: n.value])}

    # ---- inserts
    def add_rows(self, obs, act, rew, term, next_obs, ep_end=None):
        """n x add_sample (+ terminate_episode after flagged rows) in one device copy."""
        n = len(rew)
        obs = np.ascontiguousarray(obs, np.float32).reshape(n, self._observation_dim)
        act = np.ascontiguousarray(act, np.float32).reshape(n, self._action_dim)
        rew = np.ascontiguousarray(rew, np.float32).reshape(n)
        term = np.ascontiguousarray(term).reshape(n).astype(np.uint8)
        next_obs = np.ascontiguousarray(next_obs, np.float32).reshape(n, self._observation_dim)
        ee = None
        if ep_end is not None:
            ee = np.ascontiguousarray(ep_end).astype(np.uint8)
        vp = C.c_void_p
        _lib.check(self.ctx.lib.ilsx_replay_add(
            self.h, obs.ctypes.data_as(vp), act.ctypes.data_as(vp), rew.ctypes.data_as(vp), term.ctypes.data_as(vp),
            next_obs.ctypes.data_as(vp), n, ee.ctypes.data_as(vp) if ee is not None else None, 0))

    def add_sample(self, observation, action, reward, terminal, next_observation, timeout=False, **kwargs):
        self.add_rows(np.asarray(observation)[None], np.asarray(action)[None], [reward], [terminal],
                      np.asarray(next_observation)[None])

    def terminate_episode(self):
        _lib.check(self.ctx.lib.ilsx_replay_terminate_episode(self.h))

    def add_path(self, path, absorbing=False, env=None):  # simple_replay_buffer.py:134-216
        n = len(path["rewards"])
        if not absorbing:
            ep_end = np.zeros(n, np.uint8)
            ep_end[-1] = 1
            self.add_rows(path["observations"], path["actions"], np.asarray(path["rewards"]).reshape(n),
                          np.asarray(path["terminals"]).reshape(n), path["next_observations"], ep_end)
            self._trajs += 1
            return
        # wrap_absorbing (:163-213): stored terminals are all False; a terminal transition is followed by
        # (next_ob -> 0, absorbing [0,1]) and (0 -> 0, absorbing [1,1]) with freshly sampled actions and the same reward
        obs, act, rew, nobs, ab = [], [], [], [], []
        for ob, a_, r_, nob, term in zip(path["observations"], path["actions"], np.asarray(path["rewards"]).reshape(n),
                                         path["next_observations"], np.asarray(path["terminals"]).reshape(n)):
            obs.append(ob), act.append(a_), rew.append(r_), nobs.append(nob), ab.append((0.0, 0.0))
            if term:
                zero = np.zeros_like(nob)
                obs.append(nob), act.append(env.action_space.sample()), rew.append(r_), nobs.append(zero), ab.append((0.0, 1.0))
                obs.append(zero), act.append(env.action_space.sample()), rew.append(r_), nobs.append(zero), ab.append((1.0, 1.0))
        m = len(rew)
        if m > self._max_replay_buffer_size:
            raise ValueError("path longer than the buffer")
        ep_end = np.zeros(m, np.uint8)
        ep_end[-1] = 1
        top = self._top
        self.add_rows(np.asarray(obs), np.asarray(act), np.asarray(rew), np.zeros(m, np.uint8), np.asarray(nobs), ep_end)
        flags = np.ascontiguousarray(ab, np.float32)
        _lib.check(self.ctx.lib.ilsx_replay_set_absorbing(self.h, top, m, flags.ctypes.data_as(C.c_void_p)))
        self._trajs += 1

    def get_traj_num(self):
        return self._trajs

    # ---- sampling
    def num_steps_can_sample(self):
        return self._size

    def _gather(self, indices, keys=None):
        idx = np.ascontiguousarray(indices, np.int64)
        B = idx.size
        o, a, ctx = self._observation_dim, self._action_dim, self.ctx
        want_abs = keys is None or "absorbing" in keys
        if B == 0:   # get_all() / a gather on an empty buffer returns empty arrays in the reference
            ret = dict(observations=np.empty((0, o), np.float32), actions=np.empty((0, a), np.float32), rewards=np.empty((0, 1), np.float32),
                       terminals=np.empty((0, 1), np.uint8), next_observations=np.empty((0, o), np.float32), absorbing=np.empty((0, 2)))
        else:
            k_idx, p_idx = as_dev(ctx, idx, np.int64)
            obs, act, rew, done, nobs = (ctx.empty((B, o)), ctx.empty((B, a)), ctx.empty((B,)), ctx.empty((B,)),
                                         ctx.empty((B, o)))
            _lib.check(ctx.lib.ilsx_replay_sample(self.h, B, p_idx, obs.ptr, act.ptr, rew.ptr, done.ptr, nobs.ptr, None))
            ret = dict(observations=obs.numpy(), actions=act.numpy(), rewards=rew.numpy().reshape(B, 1),
                       terminals=done.numpy().reshape(B, 1).astype(np.uint8), next_observations=nobs.numpy())
            if want_abs:   # one more launch + copy: only when the caller asks for the key
                absb = ctx.empty((B, 2))
                _lib.check(ctx.lib.ilsx_replay_get_absorbing(self.h, p_idx, B, absb.ptr))
                ret["absorbing"] = absb.numpy().astype(np.float64)
        if keys is not None:
            ret = {k: v for k, v in ret.items() if k in keys}
        return ret

    def random_batch(self, batch_size, keys=None, **kwargs):  # :239-253
        indices = self._np_rand_state.randint(0, self._size, batch_size)
        return self._gather(indices, keys)

    def _get_batch_using_indices(self, indices, keys=None, **kwargs):
        return self._gather(list(indices), keys)

    def _get_segment(self, start, end, keys=None):  # :325-332
        cap = self._max_replay_buffer_size
        if start < end or end == 0:
            if end == 0:
                end = cap
            return self._gather(np.arange(start, end), keys)
        return self._gather(np.concatenate([np.arange(start, cap), np.arange(0, end)]), keys)

    def _np_randint(self, *args, **kwargs):  # :70-72
        return self._np_rand_state.randint(*args, **kwargs)

    def _np_choice(self, *args, **kwargs):  # :74-76
        return self._np_rand_state.choice(*args, **kwargs)

    def _get_samples_from_traj(self, start, end, samples_per_traj, keys=None):  # :334-347: subsample a trajectory
        cap = self._max_replay_buffer_size
        if start < end or end == 0:
            inds = range(start, cap if end == 0 else end)
        else:
            inds = list(range(start, cap)) + list(range(0, end))
        inds = self._np_choice(inds, size=samples_per_traj, replace=len(inds) < samples_per_traj)
        return self._gather(inds, keys)

    def sample_trajs(self, num_trajs, keys=None, samples_per_traj=None):  # :349-369
        keys_list = list(self._traj_endpoints.keys())
        ends_of = self._traj_endpoints
        starts = self._np_choice(keys_list, size=num_trajs, replace=len(keys_list) < num_trajs)
        if samples_per_traj is None:
            return [self._get_segment(int(s), ends_of[s], keys) for s in starts]
        return [self._get_samples_from_traj(int(s), ends_of[s], samples_per_traj, keys) for s in starts]

    def sample_all_trajs(self, keys=None, samples_per_traj=None):  # :374-395
        items = list(self._traj_endpoints.items())
        if samples_per_traj is None:
            return [self._get_segment(s, e, keys) for s, e in items]
        return [self._get_samples_from_traj(s, e, samples_per_traj, keys) for s, e in items]

    def get_all(self, keys=None, **kwargs):  # :219-226
        return self._gather(np.arange(self._size), keys)

    def save_data(self, save_name):  # :110-123: the rows [0, _top) as one pickled dict
        import pickle
        n = self._top
        b = self._gather(np.arange(n)) if n else {k: np.zeros((0, 1)) for k in ("observations", "actions", "next_observations", "terminals", "rewards")}
        d = dict(observations=b["observations"], actions=b["actions"], next_observations=b["next_observations"], terminals=b["terminals"],
                 timeouts=np.zeros((n, 1), np.uint8), rewards=b["rewards"], agent_infos=[None] * n, env_infos=[None] * n)
        with open(save_name, "wb") as f:
            pickle.dump(d, f)

    def clear(self):
        _lib.check(self.ctx.lib.ilsx_replay_clear(self.h))
        self._trajs = 0


class EnvReplayBuffer(SimpleReplayBuffer):
    """env_replay_buffer.py:7-20: dims taken from the env's spaces."""

    def __init__(self, max_replay_buffer_size, env, random_seed=1995, ctx=None):
        self._ob_space, self._action_space = env.observation_space, env.action_space
        super().__init__(max_replay_buffer_size, int(np.prod(self._ob_space.shape)),
                         int(np.prod(self._action_space.shape)), random_seed, ctx)
