"""Evaluation sampler + path statistics: the reference's `VecPathSampler.obtain_samples` / `rollout`
(rlkit/samplers/vec_sampler.py:5-142), `PathBuilder` (rlkit/data_management/path_builder.py:4-60) and
`eval_util.get_generic_path_information` / `get_average_returns` (rlkit/core/eval_util.py:15-142).

Semantics kept: reset every env, step the policy for up to `max_path_length`; an env that terminates is
dropped from the ready set (no auto-reset), so one call yields exactly one episode per env; repeat until
at least `num_steps` transitions were collected.  The column names produced by
`get_generic_path_information` are the `progress.csv` schema curves are compared on.
"""
from collections import OrderedDict

import numpy as np


class PathBuilder(dict):
    def __init__(self):
        super().__init__()
        self._path_length = 0

    def add_all(self, **key_to_value):
        for k, v in key_to_value.items():
            self.setdefault(k, []).append(v)
        self._path_length += 1

    def __len__(self):
        return self._path_length


def rollout(env, policy, max_path_length, no_terminal=False):
    n = len(env)
    paths = [PathBuilder() for _ in range(n)]
    ready = np.arange(n)
    obs = env.reset(ready)
    for _ in range(max_path_length):
        actions = policy.get_actions(obs)
        next_obs, rewards, terminals, infos = env.step(actions, ready)
        if no_terminal:
            terminals = np.zeros(len(ready), dtype=bool)
        for i, e in enumerate(ready):
            paths[e].add_all(observations=obs[i], actions=actions[i], rewards=np.array([rewards[i]]),
                             next_observations=next_obs[i], terminals=np.array([terminals[i]]),
                             absorbings=np.array([0.0, 0.0]), env_infos=infos[i])
        terminals = np.asarray(terminals, dtype=bool)
        obs = next_obs[~terminals]
        if terminals.any():
            ready = ready[~terminals]
            if len(ready) == 0:
                break
    return paths


class VecPathSampler:
    def __init__(self, env, policy, num_steps, max_path_length, no_terminal=False, **kwargs):
        self.env, self.policy = env, policy
        self.num_steps, self.max_path_length, self.no_terminal = num_steps, max_path_length, no_terminal

    def obtain_samples(self, num_steps=None):
        paths, total = [], 0
        num_steps = self.num_steps if num_steps is None else num_steps
        while total < num_steps:
            new = rollout(self.env, self.policy, self.max_path_length, no_terminal=self.no_terminal)
            paths.extend(new)
            total += sum(len(p) for p in new)
        return paths


def create_stats_ordered_dict(name, data, stat_prefix=None, always_show_all_stats=False, exclude_max_min=False):
    """rlkit/core/eval_util.py:92-142: `<prefix> <name> Mean/Std/Max/Min`; a Number passes through, a tuple fans out into
    `<name>_<i>` entries (computed without the prefix-less flags, as there), a list of arrays is concatenated, a size-1 array
    collapses to its value unless `always_show_all_stats`."""
    from numbers import Number
    if stat_prefix is not None:
        name = "{} {}".format(stat_prefix, name)
    if isinstance(data, Number):
        return OrderedDict({name: data})
    if len(data) == 0:
        return OrderedDict()
    if isinstance(data, tuple):
        out = OrderedDict()
        for number, d in enumerate(data):
            out.update(create_stats_ordered_dict("{0}_{1}".format(name, number), d))
        return out
    if isinstance(data, list):
        try:
            iter(data[0])
        except TypeError:
            pass
        else:
            data = np.concatenate([np.asarray(d) for d in data])
    data = np.asarray(data)
    if data.size == 1 and not always_show_all_stats:
        return OrderedDict({name: float(data.reshape(-1)[0])})
    stats = OrderedDict([(name + " Mean", np.mean(data)), (name + " Std", np.std(data))])
    if not exclude_max_min:
        stats[name + " Max"] = np.max(data)
        stats[name + " Min"] = np.min(data)
    return stats


def get_generic_path_information(paths, stat_prefix=""):
    """rlkit/core/eval_util.py:15-81, incl. the `is_success` branch (Success Num / Traj Num / Success Rate)."""
    st = OrderedDict()
    returns = [np.sum(p["rewards"], axis=0).reshape(-1) for p in paths]     # `sum(path["rewards"])`: one (1,) array per path
    rewards = np.concatenate([np.asarray(p["rewards"]).reshape(-1) for p in paths])
    st.update(create_stats_ordered_dict("Rewards", rewards, stat_prefix, True))
    st.update(create_stats_ordered_dict("Returns", returns, stat_prefix, True))
    infos0 = paths[0].get("env_infos")
    if infos0 is not None and len(infos0) and "is_success" in infos0[0]:
        acc_sum = [float(np.sum([x["is_success"] for x in p["env_infos"]]) > 0) for p in paths]
        st.update(create_stats_ordered_dict("Success Num", float(np.sum(acc_sum)), stat_prefix, True))
        st.update(create_stats_ordered_dict("Traj Num", len(paths), stat_prefix, True))
        st.update(create_stats_ordered_dict("Success Rate", float(np.sum(acc_sum)) / len(paths), stat_prefix, True))
    st.update(create_stats_ordered_dict("Actions", [np.asarray(p["actions"]) for p in paths], stat_prefix, True))
    st.update(create_stats_ordered_dict("Ep. Len.", np.array([len(p["terminals"]) for p in paths]), stat_prefix, True))
    st["Num Paths"] = len(paths)
    return st


def get_average_returns(paths, std=False):   # eval_util.py:84-89
    returns = [float(np.sum(p["rewards"])) for p in paths]
    if std:
        return float(np.mean(returns)), float(np.std(returns))
    return float(np.mean(returns))


class DeviceEvalSampler:
    """VecPathSampler.obtain_samples + get_generic_path_information without leaving the device (ilsx_eval_rollout): every
    call of the library plays one episode per env; calls repeat until `num_steps` samples are in (vec_sampler.py:124-142).
    Returns the statistics dict directly (there are no host-side paths)."""

    def __init__(self, env, policy, num_steps, max_path_length, **kwargs):
        self.env, self.policy, self.num_steps, self.max_path_length = env, policy, num_steps, max_path_length

    def _handles(self):
        pol, det = self.policy, False
        if hasattr(pol, "stochastic_policy"):        # MakeDeterministic (policies.py:19-36)
            pol, det = pol.stochastic_policy, True
        ppo = getattr(pol, "_ppo", None)
        return (None, ppo.h, det) if ppo is not None else (pol.h, None, det)

    def obtain_statistics(self, stat_prefix="Test"):
        import ctypes as C

        from . import _lib
        pi, ppo, det = self._handles()
        st = (C.c_double * 18)()
        first = True
        while first or st[1] < self.num_steps:
            _lib.check(self.env.ctx.lib.ilsx_eval_rollout(self.env.h, pi, ppo, int(self.max_path_length), int(det), int(first), st))
            first = False
        return self.stats_dict(st, stat_prefix)

    def stats_dict(self, st, stat_prefix="Test"):
        """The path statistics (get_generic_path_information's keys, eval_util.py:15-80) from ilsx_eval_rollout's 18 sums / extrema."""
        n_paths, n_steps = st[0], st[1]
        out = OrderedDict()

        def block(name, i, n):
            mean = st[i] / n
            out[f"{stat_prefix} {name} Mean"] = mean
            out[f"{stat_prefix} {name} Std"] = float(np.sqrt(max(st[i + 1] / n - mean * mean, 0.0)))
            out[f"{stat_prefix} {name} Max"], out[f"{stat_prefix} {name} Min"] = st[i + 2], st[i + 3]
        block("Rewards", 10, n_steps)
        block("Returns", 2, n_paths)
        block("Actions", 14, n_steps * self.env.act_dim)
        block("Ep. Len.", 6, n_paths)
        out["Num Paths"] = int(n_paths)
        out["AverageReturn"] = st[2] / n_paths
        return out
