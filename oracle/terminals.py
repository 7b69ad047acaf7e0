"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — numpy restatement of the batched terminal predicates of
rlkit/envs/terminals.py:14-117 (what MBPO's FakeEnv and the envpool adapter use): done[n,1] = f(next_obs[n,o]).

Pinned by tests/golden/g14_terminals.npz (the reference's own classes run on seeded inputs).  Quirks kept on purpose:
Hopper's state bound is `abs(next_obs[:, 1:] < 100)` (:61) — the abs of a boolean, i.e. only an UPPER bound of 100 —
and NaN comparisons are False, so a NaN height ends a Walker2d episode but not a Humanoid one.
"""
import numpy as np

KINDS = ("inverted_pendulum", "inverted_double_pendulum", "hopper", "walker2d", "halfcheetah", "humanoid", "ant")


def is_terminal(kind, next_obs):
    x = np.asarray(next_obs)
    assert x.ndim == 2
    if kind == "inverted_pendulum":            # terminals.py:23-33
        done = ~(np.all(np.isfinite(x), axis=-1) & (np.abs(x[:, 1]) <= 0.2))
    elif kind == "inverted_double_pendulum":   # :36-50
        th1, th2 = np.arctan2(x[:, 1], x[:, 3]), np.arctan2(x[:, 2], x[:, 4])
        done = 0.6 * (x[:, 3] + np.cos(th1 + th2)) <= 1
    elif kind == "hopper":                     # :53-69
        done = ~(np.all(np.isfinite(x), axis=-1) & np.all(x[:, 1:] < 100, axis=-1) & (x[:, 0] > 0.7) & (np.abs(x[:, 1]) < 0.2))
    elif kind == "walker2d":                   # :72-83
        done = ~((x[:, 0] > 0.8) & (x[:, 0] < 2.0) & (x[:, 1] > -1.0) & (x[:, 1] < 1.0))
    elif kind == "halfcheetah":                # :86-94
        done = np.zeros(len(x), bool)
    elif kind == "humanoid":                   # :97-106
        done = (x[:, 0] < 1.0) | (x[:, 0] > 2.0)
    elif kind == "ant":                        # :109-117
        done = ~(np.all(np.isfinite(x), axis=-1) & (x[:, 0] >= 0.2) & (x[:, 0] <= 1.0))
    else:
        raise KeyError(kind)
    return done[:, None]
