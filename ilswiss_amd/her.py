"""Hindsight Experience Replay on libilsx: the goal-conditioned trainers of rlkit/torch/algorithms/her/ (td3.py, sac.py), the
relabelling buffer (rlkit/data_management/relabel_replay_buffer.py), the exploration policy the HER scripts use
(policies.py:480-566 `MlpGaussianAndEpsilonPolicy` under `ConditionPolicy`, :568-640) and the loop wrapper (her/her.py).

The gradient steps run on the device (the TD3 / SAC-alpha kernels; `ilsx_td3_cfg.her` switches on the three places where
her/td3.py departs from td3.py).  The buffer keeps the reference's dictionary observations in host memory and relabels there — goals
are a few floats per sample and the reference's specs train one small batch per env step.  `HindsightReplayBuffer` is the host form (any python
`compute_reward`); `DeviceHindsightReplayBuffer` keeps the rows in an HBM ring and relabels in one gather kernel (ilsx_her_gather), so
HER batches do not cross PCIe either — the HER loop uses it whenever the env exposes gym's sparse / dense goal reward rule.  The reference's goal envs are gym's Fetch robots (MuJoCo),
which do not exist here: `PointReachEnv` below is a stand-in with the same dictionary interface, used by the smoke test only.
"""
import random

import numpy as np

from .sac import SoftActorCritic
from .td3 import TD3 as _TD3, MlpGaussianNoisePolicy


def _cat(batch):
    """her/td3.py:95-99, her/sac.py:80-84: networks see observation | desired_goal."""
    if batch.get("_her_cat"):      # DeviceHindsightReplayBuffer: already concatenated on the device by ilsx_her_gather
        return {k: v for k, v in batch.items() if k != "_her_cat"}
    out = dict(batch)
    out["observations"] = np.concatenate([np.asarray(batch["observations"], np.float32), np.asarray(batch["desired_goals"], np.float32)], -1)
    out["next_observations"] = np.concatenate([np.asarray(batch["next_observations"], np.float32),
                                               np.asarray(batch["next_desired_goals"], np.float32)], -1)
    return out


class TD3(_TD3):
    """her/td3.py:13-245.  `policy` is a MlpGaussianAndEpsilonPolicy (its sigma / max_act feed the target-action noise, :104-114)."""

    def __init__(self, policy, qf1, qf2, discount=0.99, clip_return_l=None, clip_return_r=None, **kwargs):
        gamma_sum = 1.0 / (1.0 - discount)                                  # her/td3.py:79-86
        self.clip_return_l = -gamma_sum if clip_return_l is None else clip_return_l
        self.clip_return_r = 0.0 if clip_return_r is None else clip_return_r
        super().__init__(policy, qf1, qf2, discount=discount, her=True, clip_return_l=self.clip_return_l,
                         clip_return_r=self.clip_return_r, **kwargs)

    def train_step(self, batch, eps_target=None):
        super().train_step(_cat(batch), eps_target)


class SAC(SoftActorCritic):
    """her/sac.py:12-251: sac_alpha on concatenated inputs, target entropy -|A| (:52)."""

    def __init__(self, policy, qf1, qf2, **kwargs):
        kwargs.pop("target_entropy", None)
        super().__init__(policy, qf1, qf2, target_entropy=-float(policy.action_dim), **kwargs)

    def train_step(self, batch, eps_next=None, eps_cur=None):
        super().train_step(_cat(batch), eps_next, eps_cur)


class MlpGaussianAndEpsilonPolicy(MlpGaussianNoisePolicy):
    """policies.py:480-566 (+ the dictionary handling of ConditionPolicy, :587-640): deterministic tanh MLP on the device; exploration
    on the host — with probability epsilon a uniform action from the action space, else the action plus N(0, sigma^2) clipped to
    [min_act, max_act], sigma annealed from max_sigma to min_sigma over decay_period env steps."""

    def __init__(self, hidden_sizes, obs_dim, action_dim, action_space=None, condition_dim=0, epsilon=0.3, max_sigma=0.2, min_sigma=0.2,
                 decay_period=1000000, max_act=1.0, min_act=-1.0, observation_key="observation", desired_goal_key="desired_goal", **kwargs):
        # output_activation travels in kwargs to MlpGaussianNoisePolicy (identity unless given; her_td3_exp_script.py:85 passes tanh)
        if min_act != -max_act:
            raise NotImplementedError("the device target-action clamp is symmetric (her/td3.py:111-114 with the reference's defaults)")
        super().__init__(hidden_sizes, obs_dim + condition_dim, action_dim, policy_noise=max_sigma, policy_noise_clip=0.0, max_act=max_act,
                         **kwargs)
        self.sigma, self._max_sigma, self._min_sigma = max_sigma, max_sigma, (max_sigma if min_sigma is None else min_sigma)
        self._epsilon, self._decay_period, self._action_space = epsilon, decay_period, action_space
        self.min_act, self.t = min_act, 0
        self.observation_key, self.desired_goal_key = observation_key, desired_goal_key
        # exploration randomness comes from the GLOBAL `random` / `np.random` streams, which set_seed(variant seed) seeds — as in the
        # reference (policies.py:537-552); a private RandomState here gave every variant seed the same exploration noise

    def set_num_steps_total(self, t):
        self.t = t

    def _flat(self, obs):
        if isinstance(obs, dict):
            return np.concatenate([obs[self.observation_key], obs[self.desired_goal_key]], -1)
        if len(obs) and isinstance(obs[0], dict):
            return np.array([np.concatenate([x[self.observation_key], x[self.desired_goal_key]], -1) for x in obs])
        return np.asarray(obs)

    def get_actions(self, obs_np, deterministic=False):
        obs = np.atleast_2d(self._flat(obs_np)).astype(np.float32)
        action = super().get_actions(obs, deterministic=True)
        if deterministic:
            return action
        if random.random() < self._epsilon:
            return np.array([self._action_space.sample() for _ in range(obs.shape[0])], np.float32)
        self.sigma = self._max_sigma - (self._max_sigma - self._min_sigma) * min(1.0, self.t * 1.0 / self._decay_period)
        return np.clip(action + np.random.normal(size=action.shape) * self.sigma, self.min_act, self.max_act).astype(np.float32)

    def get_action(self, obs_np, deterministic=False):
        return self.get_actions(self._flat(obs_np)[None], deterministic)[0], {}


class ConditionedPolicy:
    """ConditionPolicy (policies.py:568-640) over any device policy: dictionary observations -> observation | desired_goal."""

    def __init__(self, policy, observation_key="observation", desired_goal_key="desired_goal"):
        self.policy, self.observation_key, self.desired_goal_key = policy, observation_key, desired_goal_key

    def set_num_steps_total(self, t):
        pass

    def _flat(self, obs):
        if isinstance(obs, dict):
            return np.concatenate([obs[self.observation_key], obs[self.desired_goal_key]], -1)
        return np.asarray(obs)

    def get_actions(self, obs_np, deterministic=False):
        return self.policy.get_actions(np.atleast_2d(self._flat(obs_np)).astype(np.float32), deterministic=deterministic)

    def get_action(self, obs_np, deterministic=False):
        return self.get_actions(obs_np, deterministic)[0], {}


class HindsightReplayBuffer:
    """relabel_replay_buffer.py:13-163 over the dictionary branch of simple_replay_buffer.py (:36-49, 78-132, 228-237, 255-293):
    host arrays, the reference's cursor logic, the reference's random-number call order (its own RandomState for the trajectory and
    step draws, the GLOBAL numpy stream for the `future` index, :88)."""

    def __init__(self, max_replay_buffer_size, env, random_seed=1995, relabel_type="future", her_ratio=0.8,
                 observation_key="observation", desired_goal_key="desired_goal", achieved_goal_key="achieved_goal"):
        self._np_rand_state = np.random.RandomState(random_seed)
        self._max_replay_buffer_size = cap = int(max_replay_buffer_size)
        spaces = env.observation_space.spaces
        self._action_dim = int(np.prod(env.action_space.shape))
        self.compute_reward = env.compute_reward
        self.her_ratio, self.relabel_type = her_ratio, relabel_type
        self.observation_key, self.desired_goal_key, self.achieved_goal_key = observation_key, desired_goal_key, achieved_goal_key
        self._observations = {k: np.zeros((cap, int(np.prod(sp.shape)))) for k, sp in spaces.items()}
        self._next_obs = {k: np.zeros((cap, int(np.prod(sp.shape)))) for k, sp in spaces.items()}
        self._actions = np.zeros((cap, self._action_dim))
        self._rewards = np.zeros((cap, 1))
        self._terminals = np.zeros((cap, 1), dtype="uint8")
        self._top = self._size = self._cur_start = 0
        self._traj_endpoints = {}

    def add_sample(self, observation, action, reward, terminal, next_observation, **kwargs):
        assert isinstance(observation, dict), "Observation should be dict!"
        t = self._top
        self._actions[t], self._rewards[t], self._terminals[t] = action, reward, terminal
        if terminal:
            nxt = (t + 1) % self._max_replay_buffer_size
            self._traj_endpoints[self._cur_start] = nxt
            self._cur_start = nxt
        for k, v in observation.items():
            self._observations[k][t] = v
        for k, v in next_observation.items():
            self._next_obs[k][t] = v
        if self._top in self._traj_endpoints:       # _advance
            del self._traj_endpoints[self._top]
        self._top = (self._top + 1) % self._max_replay_buffer_size
        if self._size < self._max_replay_buffer_size:
            self._size += 1

    def terminate_episode(self):
        if self._cur_start != self._top:
            self._traj_endpoints[self._cur_start] = self._top
            self._cur_start = self._top

    def num_steps_can_sample(self):
        return self._size

    def _gather(self, indices, with_all=True):
        out = dict(observations={k: v[indices] for k, v in self._observations.items()},
                   next_observations={k: v[indices] for k, v in self._next_obs.items()})
        if with_all:
            out.update(actions=self._actions[indices], rewards=self._rewards[indices], terminals=self._terminals[indices])
        return out

    def random_batch(self, batch_size, keys=None, **kwargs):
        relabel = (self.relabel_type is not None) and (self.her_ratio > 0)
        keys_list = list(self._traj_endpoints.keys())
        starts = self._np_rand_state.choice(keys_list, size=len(keys_list), replace=False)
        ends = [self._traj_endpoints[k] for k in starts]
        traj_indice = self._np_rand_state.randint(0, len(starts), batch_size)
        indices, indices_relabel = [], []
        for i in traj_indice:
            traj_len = (ends[i] - starts[i]) % self._size
            step = (self._np_rand_state.randint(0, traj_len, 1)[0] + starts[i]) % self._size
            indices.append(step)
            if relabel:
                # the reference builds an eager dict literal (relabel_replay_buffer.py:85-88): the `future` draw from the GLOBAL numpy
                # stream happens for every relabelled sample whatever the relabel_type — drawn here too so later global draws agree
                try:
                    fut = np.random.randint(step, traj_len + starts[i])
                except ValueError:      # empty range after ring wrap-around (the reference prints and exits, :89-91); harmless for `final`
                    if self.relabel_type != "final":
                        raise
                    fut = 0
                indices_relabel.append(ends[i] - 1 if self.relabel_type == "final" else fut % self._size)
        b = self._gather(indices)
        if relabel:
            n = int(self.her_ratio * batch_size)
            src = self._gather(indices_relabel, with_all=False)["next_observations"][self.achieved_goal_key]
            b["observations"][self.desired_goal_key][:n] = src[:n].copy()
            b["next_observations"][self.desired_goal_key][:n] = src[:n].copy()
        b["achieved_goals"] = b["observations"][self.achieved_goal_key]
        b["desired_goals"] = b["observations"][self.desired_goal_key]
        b["next_achieved_goals"] = b["next_observations"][self.achieved_goal_key]
        b["next_desired_goals"] = b["next_observations"][self.desired_goal_key]
        b["observations"] = b["observations"][self.observation_key]
        b["next_observations"] = b["next_observations"][self.observation_key]
        if relabel:
            b["rewards"] = np.asarray(self.compute_reward(b["next_achieved_goals"], b["desired_goals"], info=None)).reshape(-1, 1)
        return b


class DeviceHindsightReplayBuffer(HindsightReplayBuffer):
    """The same buffer with its rows in an HBM ring: cursors, `_traj_endpoints` and every random draw stay the
    reference's (host integers, the reference's RandomState call order — g22 pins them), but the transitions live in an `ilsx_replay`
    ring whose observation segment is observation | desired_goal | achieved_goal, and `random_batch` is ONE gather kernel
    (ilsx_her_gather: goal overwrite for the first her_ratio * B rows, reward recompute for all rows, observation | goal concatenation)
    whose outputs feed the trainer's device entry point directly: 2 x B int64 indices go up per batch, no batch comes down.

    The reward rule is gym's GoalEnv one, evaluated on the device: sparse -(|achieved - goal| > distance_threshold) or dense -|.|, read
    from the env (`reward_type`, `distance_threshold`, as gym's Fetch envs expose them)."""

    def __init__(self, max_replay_buffer_size, env, random_seed=1995, relabel_type="future", her_ratio=0.8, observation_key="observation",
                 desired_goal_key="desired_goal", achieved_goal_key="achieved_goal", ctx=None):
        import ctypes as C
        from . import _lib
        from .device import get_context
        self._np_rand_state = np.random.RandomState(random_seed)
        self._max_replay_buffer_size = cap = int(max_replay_buffer_size)
        spaces = env.observation_space.spaces
        self._action_dim = int(np.prod(env.action_space.shape))
        self.her_ratio, self.relabel_type = her_ratio, relabel_type
        self.observation_key, self.desired_goal_key, self.achieved_goal_key = observation_key, desired_goal_key, achieved_goal_key
        self.d_obs, self.d_goal = int(np.prod(spaces[observation_key].shape)), int(np.prod(spaces[desired_goal_key].shape))
        assert int(np.prod(spaces[achieved_goal_key].shape)) == self.d_goal
        self.reward_kind, self.threshold = device_reward_rule(env)
        self.ctx = ctx or get_context()
        self._C, self._lib = C, _lib
        self.h = C.c_void_p()
        _lib.check(self.ctx.lib.ilsx_replay_create(self.ctx.h, cap, self.d_obs + 2 * self.d_goal, self._action_dim, C.c_uint64(random_seed),
                                                   C.byref(self.h)))
        self._top = self._size = self._cur_start = 0
        self._traj_endpoints = {}
        self._out = None

    def _row(self, d):
        return np.concatenate([np.asarray(d[self.observation_key], np.float32).ravel(), np.asarray(d[self.desired_goal_key], np.float32).ravel(),
                               np.asarray(d[self.achieved_goal_key], np.float32).ravel()])[None]

    def add_sample(self, observation, action, reward, terminal, next_observation, **kwargs):
        assert isinstance(observation, dict), "Observation should be dict!"
        C = self._C
        obs, nobs = self._row(observation), self._row(next_observation)
        act = np.asarray(action, np.float32).reshape(1, self._action_dim)
        rew, done = np.asarray([reward], np.float32), np.asarray([1 if terminal else 0], np.uint8)
        p = lambda x: x.ctypes.data_as(C.c_void_p)   # noqa: E731
        self._lib.check(self.ctx.lib.ilsx_replay_add(self.h, p(obs), p(act), p(rew), p(done), p(nobs), 1, None, 0))
        t = self._top                               # host cursors exactly as the parent's (the ring's own cursor moves in step)
        if terminal:
            nxt = (t + 1) % self._max_replay_buffer_size
            self._traj_endpoints[self._cur_start] = nxt
            self._cur_start = nxt
        if self._top in self._traj_endpoints:
            del self._traj_endpoints[self._top]
        self._top = (self._top + 1) % self._max_replay_buffer_size
        if self._size < self._max_replay_buffer_size:
            self._size += 1

    def random_batch(self, batch_size, keys=None, **kwargs):
        """Same draws as the parent (relabel_replay_buffer.py:66-100); the gather, relabel and reward run on the device.  Returns device
        arrays: observations / next_observations are observation | desired_goal rows ready for the goal-conditioned trainers."""
        C, B = self._C, int(batch_size)
        relabel = (self.relabel_type is not None) and (self.her_ratio > 0)
        keys_list = list(self._traj_endpoints.keys())
        starts = self._np_rand_state.choice(keys_list, size=len(keys_list), replace=False)
        ends = [self._traj_endpoints[k] for k in starts]
        traj_indice = self._np_rand_state.randint(0, len(starts), B)
        indices, indices_relabel = [], []
        for i in traj_indice:
            traj_len = (ends[i] - starts[i]) % self._size
            step = (self._np_rand_state.randint(0, traj_len, 1)[0] + starts[i]) % self._size
            indices.append(step)
            if relabel:
                try:
                    fut = np.random.randint(step, traj_len + starts[i])
                except ValueError:
                    if self.relabel_type != "final":
                        raise
                    fut = 0
                indices_relabel.append(ends[i] - 1 if self.relabel_type == "final" else fut % self._size)
        ctx, W, a = self.ctx, self.d_obs + self.d_goal, self._action_dim
        if self._out is None or self._out[0] != B:
            self._out = (B, ctx.empty((B, W)), ctx.empty((B, a)), ctx.empty((B,)), ctx.empty((B,)), ctx.empty((B, W)),
                         ctx.empty((B,), np.int64), ctx.empty((B,), np.int64))
        _, obs, act, rew, done, nobs, di, dr = self._out
        di.copy_from(np.asarray(indices, np.int64))
        if relabel:
            dr.copy_from(np.asarray(indices_relabel, np.int64))
        n = int(self.her_ratio * B) if relabel else 0
        self._lib.check(ctx.lib.ilsx_her_gather(self.h, di.ptr, dr.ptr if relabel else None, B, n, int(relabel), self.d_obs, self.d_goal,
                                                self.reward_kind, self.threshold, obs.ptr, act.ptr, rew.ptr, done.ptr, nobs.ptr))
        return dict(observations=obs, actions=act, rewards=rew, terminals=done, next_observations=nobs, _her_cat=True)

    def numpy_batch(self, batch):
        """A device batch split back into the reference's keys (tests)."""
        o = self.d_obs
        ob, nob = batch["observations"].numpy(), batch["next_observations"].numpy()
        B = ob.shape[0]
        return dict(observations=ob[:, :o], desired_goals=ob[:, o:], next_observations=nob[:, :o], next_desired_goals=nob[:, o:],
                    actions=batch["actions"].numpy(), rewards=batch["rewards"].numpy().reshape(B, 1),
                    terminals=batch["terminals"].numpy().reshape(B, 1).astype(np.uint8))


def device_reward_rule(env):
    """(kind, threshold) of the ONE reward rule the device-side relabel implements (k_her_gather): gym's robotics GoalEnv rule on the
    Euclidean goal distance d — kind 0 'sparse': -(d > distance_threshold), kind 1 'dense': -d.  Raises for anything else: an env whose
    reward_type is neither, or a sparse rule without a threshold attribute (`distance_threshold`, or `tol` of the stand-in env)."""
    rt = getattr(env, "reward_type", "sparse")
    if rt not in ("sparse", "dense"):
        raise NotImplementedError(f"DeviceHindsightReplayBuffer: reward_type={rt!r}; the device rule knows 'sparse' and 'dense' (use HindsightReplayBuffer)")
    thr = getattr(env, "distance_threshold", getattr(env, "tol", None))
    if rt == "sparse" and thr is None:
        raise NotImplementedError("DeviceHindsightReplayBuffer: a sparse goal reward needs env.distance_threshold")
    return (0 if rt == "sparse" else 1), float(thr if thr is not None else 0.0)


def device_reward_rule_matches(env, d_goal, n=512, seed=0):
    """Does env.compute_reward (what the reference calls on every relabelled row, relabel_replay_buffer.py:37-40) equal the device rule?
    Probed on n random goal pairs whose distances straddle the threshold (and some exact copies: d = 0).  False on any mismatch, shape
    surprise or exception — e.g. gym's HandManipulate* tasks, which expose distance_threshold but also test a rotation_threshold."""
    try:
        kind, thr = device_reward_rule(env)
        rs = np.random.RandomState(seed)
        ag = rs.uniform(-1, 1, (n, d_goal)).astype(np.float32)
        scale = (thr if kind == 0 and thr > 0 else 0.5) * rs.uniform(0, 3, (n, 1)) / np.sqrt(d_goal)
        dg = (ag + scale * rs.uniform(-1, 1, (n, d_goal))).astype(np.float32)
        dg[: n // 16] = ag[: n // 16]
        d = np.linalg.norm(ag.astype(np.float64) - dg.astype(np.float64), axis=-1)
        if kind == 0:
            keep = np.abs(d - thr) > 1e-4 * max(thr, 1e-6)     # rows within fp32 rounding of the threshold decide nothing
            want = -(d > thr).astype(np.float64)
        else:
            keep, want = np.ones(n, bool), -d
        got = np.asarray(env.compute_reward(ag, dg, None), np.float64).reshape(-1)
        return got.shape == want.shape and bool(np.allclose(got[keep], want[keep], rtol=1e-5, atol=1e-6))
    except Exception:
        return False


class Box:
    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, np.float32), np.asarray(high, np.float32)
        self.shape = self.low.shape

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(np.float32)


class DictSpace:
    def __init__(self, **spaces):
        self.spaces = spaces


class PointReachEnv:
    """A stand-in goal env with gym's GoalEnv interface (dictionary observations, `compute_reward`): a 2-D point with velocity control
    must come within `tol` of a goal drawn at reset; reward -1 until it does (the sparse reward of the Fetch tasks).  NOT one of the
    reference's environments — those are MuJoCo Fetch robots; this exists so that the HER loop can be exercised end to end."""

    def __init__(self, seed=0, tol=0.1, max_steps=50):
        self.rs, self.tol, self.max_steps = np.random.RandomState(seed), tol, max_steps
        self.observation_space = DictSpace(observation=Box(-np.ones(4), np.ones(4)), desired_goal=Box(-np.ones(2), np.ones(2)),
                                           achieved_goal=Box(-np.ones(2), np.ones(2)))
        self.action_space = Box(-np.ones(2), np.ones(2))

    def compute_reward(self, achieved_goal, desired_goal, info=None):
        return -(np.linalg.norm(np.asarray(achieved_goal) - np.asarray(desired_goal), axis=-1) > self.tol).astype(np.float32)

    def _obs(self):
        return dict(observation=np.concatenate([self.p, self.v]), desired_goal=self.g.copy(), achieved_goal=self.p.copy())

    def reset(self):
        self.p, self.v, self.g, self.k = self.rs.uniform(-0.5, 0.5, 2), np.zeros(2), self.rs.uniform(-0.8, 0.8, 2), 0
        return self._obs()

    def step(self, action):
        self.v = 0.1 * np.clip(action, -1, 1)
        self.p = np.clip(self.p + self.v, -1, 1)
        self.k += 1
        r = float(self.compute_reward(self.p, self.g))
        return self._obs(), r, False, dict(is_success=float(r == 0.0))


class HER:
    """her/her.py:8-42 over the sampling loop of base_algorithm.py:183-291 for ONE host-side goal env: HindsightReplayBuffer by default,
    `exploration_policy.set_num_steps_total` before every action, one train call of `num_train_steps_per_train_call` steps every
    `num_steps_between_train_calls` env steps."""

    def __init__(self, trainer, env, exploration_policy, replay_buffer=None, her_ratio=0.8, relabel_type="future", num_epochs=10,
                 num_steps_per_epoch=1000, num_steps_between_train_calls=1, num_train_steps_per_train_call=1, max_path_length=50,
                 min_steps_before_training=1000, batch_size=128, replay_buffer_size=100000, num_steps_per_eval=500, **kwargs):
        assert max_path_length < replay_buffer_size
        self.trainer, self.env, self.policy = trainer, env, exploration_policy
        # the device-resident buffer (batches never cross PCIe) only when the env's OWN compute_reward is the rule the device implements —
        # checked by evaluating it on a probe batch, not inferred from attribute names; the host buffer (which calls compute_reward on
        # every relabelled row like relabel_replay_buffer.py:37-40) otherwise
        spaces = getattr(getattr(env, "observation_space", None), "spaces", None)
        ok = hasattr(trainer, "ctx") and spaces is not None and "desired_goal" in spaces and \
            device_reward_rule_matches(env, int(np.prod(spaces["desired_goal"].shape)))
        cls = DeviceHindsightReplayBuffer if ok else HindsightReplayBuffer
        kw = dict(ctx=trainer.ctx) if cls is DeviceHindsightReplayBuffer else {}
        self.replay_buffer = replay_buffer or cls(replay_buffer_size, env, random_seed=np.random.randint(10000),
                                                  relabel_type=relabel_type, her_ratio=her_ratio, **kw)
        self.num_epochs, self.num_steps_per_epoch = num_epochs, num_steps_per_epoch
        self.between, self.per_call, self.max_path_length = num_steps_between_train_calls, num_train_steps_per_train_call, max_path_length
        self.min_steps, self.batch_size, self.num_steps_per_eval = min_steps_before_training, batch_size, num_steps_per_eval
        self._n_env_steps_total = self._n_train_steps_total = 0

    def evaluate(self):
        succ, n, obs, k = [], 0, self.env.reset(), 0
        while n < self.num_steps_per_eval:
            a, _ = self.policy.get_action(obs, deterministic=True)
            obs, r, d, info = self.env.step(a)
            n, k = n + 1, k + 1
            if d or k >= self.max_path_length:
                succ.append(info["is_success"])
                obs, k = self.env.reset(), 0
        return float(np.mean(succ)) if succ else 0.0

    def train(self):
        history, obs, k, since = [], self.env.reset(), 0, 0
        for epoch in range(self.num_epochs):
            for _ in range(self.num_steps_per_epoch):
                self.policy.set_num_steps_total(self._n_env_steps_total)          # her.py:33-42
                a = self.policy.get_actions(obs)[0]
                nobs, r, d, info = self.env.step(a)
                self.replay_buffer.add_sample(obs, a, r, d, nobs)
                self._n_env_steps_total, k, since, obs = self._n_env_steps_total + 1, k + 1, since + 1, nobs
                if d or k >= self.max_path_length:
                    self.replay_buffer.terminate_episode()
                    obs, k = self.env.reset(), 0
                if since >= self.between and self.replay_buffer.num_steps_can_sample() >= self.min_steps and self.replay_buffer._traj_endpoints:
                    since = 0
                    for _ in range(self.per_call):
                        self.trainer.train_step(self.replay_buffer.random_batch(self.batch_size))
                        self._n_train_steps_total += 1
            self.trainer.end_epoch()
            history.append(self.evaluate())
        return history
