"""Snapshots and resume: the optimiser halves of the trainers' get_snapshot / load_snapshot pairs and the
`load_params` path of the run scripts (rlkit/core/logger.py:31-49 `load_from_file`; run_scripts/sac_alpha_exp_script.py:
106-108,142-146; BaseAlgorithm.get_extra_data_to_save / set_steps, base_algorithm.py:560-597).

A snapshot here is a dict of plain numpy arrays / scalars (the reference pickles whole nn.Modules and optimizers; on-disk
interchange of module pickles with the reference is not attempted — `params.pkl` keeps the reference's keys, the values are
flat parameter vectors in torch `parameters()` order)."""
import ctypes as C
import os
import pickle

import numpy as np

from . import _lib


def get_opt(lib, name, h, n, which=None):
    """Adam state of one parameter block through ilsx_<name>_get_opt -> dict(exp_avg, exp_avg_sq, step, rng_step, n_train_steps)."""
    m, v, meta = np.empty(n, np.float32), np.empty(n, np.float32), _lib.OptMeta()
    fn = getattr(lib, f"ilsx_{name}_get_opt")
    args = ([h] if which is None else [h, int(which)]) + [m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), n, C.byref(meta)]
    _lib.check(fn(*args))
    return dict(exp_avg=m, exp_avg_sq=v, step=int(meta.t), rng_step=int(meta.rng_step), n_train_steps=int(meta.n_train_steps))


def set_opt(lib, name, h, state, which=None):
    m = np.ascontiguousarray(state["exp_avg"], np.float32)
    v = np.ascontiguousarray(state["exp_avg_sq"], np.float32)
    meta = _lib.OptMeta(int(state["step"]), int(state.get("rng_step", 0)), int(state.get("n_train_steps", 0)))
    fn = getattr(lib, f"ilsx_{name}_set_opt")
    args = ([h] if which is None else [h, int(which)]) + [m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), m.size, C.byref(meta)]
    _lib.check(fn(*args))


def dump_replay(rb):
    """The ring's rows in insertion order (oldest first) as plain arrays: what `save_replay_buffer: true` puts into
    extra_data.pkl (base_algorithm.py:574-576 pickles the buffer object).  Re-adding them reproduces the sampleable set;
    trajectory end points of device-inserted rows are rebuilt from the `ep_end` flags."""
    size, top = rb._cursors()
    cap = rb._max_replay_buffer_size
    if size == 0:
        o, a = rb._observation_dim, rb._action_dim
        return dict(observations=np.empty((0, o), np.float32), actions=np.empty((0, a), np.float32), rewards=np.empty(0, np.float32),
                    terminals=np.empty(0, np.uint8), next_observations=np.empty((0, o), np.float32), ep_end=np.empty(0, np.uint8),
                    absorbing=np.empty((0, 2), np.float32), capacity=cap)
    order = np.arange(size) if size < cap else (np.arange(cap) + top) % cap
    b = rb._gather(order)
    ends = np.zeros(size, np.uint8)
    pos = {int(s): i for i, s in enumerate(order)}
    starts = set()
    for s, e in rb._traj_endpoints.items():
        last = (e - 1) % cap
        if last in pos:
            ends[pos[last]] = 1
        starts.add(int(s))
    # a trajectory whose start was overwritten in the ring leaves tail rows that belong to no registered trajectory: close that tail
    # at the row before the oldest surviving trajectory start, so that restore does not merge it into the next trajectory
    if starts:
        first = min(pos[s] for s in starts if s in pos) if any(s in pos for s in starts) else None
        if first is not None and first > 0:
            ends[first - 1] = 1
    return dict(observations=b["observations"], actions=b["actions"], rewards=b["rewards"][:, 0], terminals=b["terminals"][:, 0],
                next_observations=b["next_observations"], ep_end=ends, absorbing=b["absorbing"].astype(np.float32), capacity=cap)


def restore_replay(rb, dump):
    rb.clear()
    n = len(dump["rewards"])
    for i in range(0, n, 262144):
        sl = slice(i, min(n, i + 262144))
        slot0 = rb._top
        rb.add_rows(dump["observations"][sl], dump["actions"][sl], dump["rewards"][sl], dump["terminals"][sl],
                    dump["next_observations"][sl], dump["ep_end"][sl])
        ab = dump.get("absorbing")
        if ab is not None and len(ab) and np.any(ab[sl]):     # the wrap_absorbing flags travel with the rows (simple_replay_buffer.py:66-67)
            flags = np.ascontiguousarray(ab[sl], np.float32)
            _lib.check(rb.ctx.lib.ilsx_replay_set_absorbing(rb.h, int(slot0), int(flags.shape[0]), flags.ctypes.data_as(C.c_void_p)))


def load_from_file(algorithm, load_replay_buffer=False, load_model=True, load_path=None):
    """rlkit/core/logger.py:31-49 with the same keyword names (the YAML's `load_params` dict is splatted into it):
    <load_path>/params.pkl -> algorithm.load_snapshot, <load_path>/extra_data.pkl -> counters (+ replay buffer); returns
    (algorithm, epoch to start from)."""
    epoch = 0
    if load_path:
        with open(os.path.join(load_path, "extra_data.pkl"), "rb") as f:
            extra = pickle.load(f)
        with open(os.path.join(load_path, "params.pkl"), "rb") as f:
            model = pickle.load(f)
        if load_replay_buffer:
            if "replay_buffer" not in extra:
                raise KeyError("extra_data.pkl holds no replay buffer (run with save_replay_buffer: true)")
            print(f"LOAD BUFFER from {load_path}")
            restore_replay(algorithm.replay_buffer, extra["replay_buffer"])
        if load_model:
            print(f"LOAD MODELS from {load_path}")
            algorithm.load_snapshot(model)
        algorithm.set_steps(extra)
        epoch = int(extra["epoch"]) + 1
    return algorithm, epoch
