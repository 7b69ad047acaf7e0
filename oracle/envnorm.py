"""Observation normalisation + action map, restating rlkit/data_management/normalizer.py:128-152
(RunningMeanStd), rlkit/envs/vecenvs.py:299-327 (normalize_obs, eps = np.finfo(float32).eps,
vecenvs.py:107) and rlkit/envs/wrappers.py:342-346 (NormalizedBoxEnv action map).  numpy float64
like the reference.  Test infrastructure."""
import numpy as np

EPS = np.finfo(np.float32).eps.item()


class RunningMeanStd:
    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.var = mean, std  # normalizer.py:133-135: `var` is initialised from `std`
        self.count = 0

    def update(self, x):
        batch_mean, batch_var = np.mean(x, axis=0), np.var(x, axis=0)
        batch_count = len(x)
        delta = batch_mean - self.mean
        total = self.count + batch_count
        new_mean = self.mean + delta * batch_count / total
        m2 = self.var * self.count + batch_var * batch_count + delta ** 2 * self.count * batch_count / total
        self.mean, self.var, self.count = new_mean, m2 / total, total


def normalize_obs(obs, rms, clip_max=10.0):
    return np.clip((obs - rms.mean) / np.sqrt(rms.var + EPS), -clip_max, clip_max)


def unnormalize_obs(obs, rms):
    return obs * np.sqrt(rms.var + EPS) + rms.mean


def action_map(action, lb, ub):
    return np.clip(lb + (action + 1.0) * 0.5 * (ub - lb), lb, ub)
