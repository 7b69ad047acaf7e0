"""Batched terminal predicates (rlkit/envs/terminals.py) and the envpool-shaped adapter (rlkit/envs/envpool.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g14_terminals.npz")
NAMES = dict(inverted_pendulum="inverted_pendulum", inverted_double_pendulum="inverted_double_pendulum", hopper="hopper",
             walker2d="walker2d", halfcheetah="halfcheetah", humanoid="humanoid", ant="ant")


def test_oracle_matches_reference_vectors():
    from oracle import terminals as oterm
    g = np.load(GOLD)
    for kind in oterm.KINDS:
        x, done = g[kind + "_x"], g[kind + "_done"]
        assert done.shape == (len(x), 1) and 0 < done.sum() < done.size or kind == "halfcheetah"
        np.testing.assert_array_equal(oterm.is_terminal(kind, x), done, err_msg=kind)
    # the Hopper quirk the vectors pin: -150 in a state column is NOT terminal, +150 is (terminals.py:61)
    x = np.zeros((2, 11), np.float32); x[:, 0] = 1.0
    x[0, 4], x[1, 4] = -150.0, 150.0
    assert oterm.is_terminal("hopper", x).ravel().tolist() == [False, True]


def test_name_rule():
    from ilswiss_amd.envs import terminals as T
    assert T.get_terminal_func("inverted_double_pendulum").__self__ is T.InvertedDoublePendulumTerminalFunc
    assert T.get_terminal_func("walker2d").__self__.kind == 3
    with pytest.raises(KeyError):
        T.get_terminal_func("swimmer")
    with pytest.raises(NotImplementedError):
        T.TerminalFunc.is_terminal(None, None, None)


def test_envpool_task_names():
    from ilswiss_amd.envs.envpool import _model_name
    assert _model_name("Hopper-v3") == "hopper" and _model_name("Walker2d-v4") == "walker2d" and _model_name("HalfCheetah-v2") == "halfcheetah"
    assert _model_name("Ant-v3") == "ant" and _model_name("Humanoid-v4") == "humanoid"    # the 3-D steppers (csrc/env3d_wave.h)
    with pytest.raises(KeyError):
        _model_name("Swimmer-v3")


def test_envpool_adapter_info_conversion_without_a_device():
    """EnvpoolEnv.step (rlkit/envs/envpool.py:17-27): dict of arrays -> list of dicts, the nested "players" entry dropped;
    attribute access and len() forward to the pool."""
    from ilswiss_amd.envs.envpool import EnvpoolEnv

    class FakePool:
        marker = "pool"

        def __len__(self):
            return 3

        def step(self, actions, env_id=None):
            n = len(actions)
            return (np.zeros((n, 2)), np.ones(n), np.zeros(n, bool),
                    dict(env_id=np.arange(n) + 5, elapsed_step=np.full(n, 7), players=dict(env_id=np.arange(n))))

    env = object.__new__(EnvpoolEnv)
    env._envs = FakePool()
    obs, rew, done, info = env.step(np.zeros((3, 1)))
    assert isinstance(info, list) and len(info) == 3 and info[2] == dict(env_id=7, elapsed_step=7)
    assert len(env) == 3 and env.marker == "pool" and env.envs is env._envs


@pytest.mark.gpu
def test_hip_matches_reference_vectors(ctx):
    from ilswiss_amd.envs.terminals import get_terminal_func
    g = np.load(GOLD)
    for kind, name in NAMES.items():
        x, done = g[kind + "_x"], g[kind + "_done"]
        f = get_terminal_func(name)
        got = f(x, np.zeros((len(x), 1), np.float32), x, ctx=ctx)
        assert got.shape == done.shape and got.dtype == bool
        np.testing.assert_array_equal(got, done, err_msg=kind)
        # device arrays stay on the device
        dx = ctx.from_numpy(x)
        np.testing.assert_array_equal(f(dx, None, dx, ctx=ctx).numpy().astype(bool)[:, None], done)
    # argument errors are reported, not executed
    import ctypes as C
    from ilswiss_amd import _lib
    dx = ctx.from_numpy(np.zeros((4, 3), np.float32)); dd = ctx.empty((4,), np.uint8)
    with pytest.raises(RuntimeError, match="unknown kind"):
        _lib.check(ctx.lib.ilsx_is_terminal(ctx.h, 9, dx.ptr, 4, 3, dd.ptr))
    with pytest.raises(RuntimeError, match="observation columns"):
        _lib.check(ctx.lib.ilsx_is_terminal(ctx.h, 1, dx.ptr, 4, 3, dd.ptr))   # the double pendulum reads 5 columns
    # ragged / degenerate sizes
    assert get_terminal_func("hopper")(np.zeros((0, 11), np.float32), None, np.zeros((0, 11), np.float32), ctx=ctx).shape == (0, 1)
    big = np.random.default_rng(0).normal(1.2, 0.4, (4099, 376)).astype(np.float32)
    from oracle import terminals as oterm
    np.testing.assert_array_equal(get_terminal_func("humanoid")(big, None, big, ctx=ctx), oterm.is_terminal("humanoid", big))
    np.testing.assert_array_equal(get_terminal_func("ant")(big, None, big, ctx=ctx), oterm.is_terminal("ant", big))


@pytest.mark.gpu
def test_stepper_done_flag_agrees_with_terminal_func(ctx):
    from ilswiss_amd.envs.terminals import get_terminal_func
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    rng = np.random.default_rng(3)
    for name in ("hopper", "walker2d", "halfcheetah"):
        env = HipVectorEnv(name, 256, seed=2, ctx=ctx)
        env.reset()
        seen = 0
        for _ in range(30):
            obs, _, done, _ = env.step(rng.uniform(-1, 1, (256, env.act_dim)).astype(np.float32))
            np.testing.assert_array_equal(get_terminal_func(name)(obs, None, obs.astype(np.float32), ctx=ctx).ravel(), done)
            seen += int(done.sum())
        assert (seen > 0) == (name != "halfcheetah")
        env.close()


@pytest.mark.gpu
def test_envpool_adapter_3d_tasks(ctx):
    """envpool_name Ant-* / Humanoid-* reach the 3-D steppers (the reference hands any task id to envpool.make, rlkit/envs/envpool.py:4-11)."""
    from ilswiss_amd.envs import get_envs
    for name, o, a in (("Ant-v3", 111, 8), ("Humanoid-v4", 376, 17)):
        env = get_envs(dict(use_envpool=True, envpool_name=name, env_type="gym", env_name="x", env_kwargs={}, env_num=4, training_env_seed=1), ctx=ctx)
        assert env.reset().shape == (4, o)
        ob, rew, done, info = env.step(np.zeros((4, a), np.float32))
        assert ob.shape == (4, o) and np.all(np.isfinite(ob)) and np.all(np.isfinite(rew)) and [i["env_id"] for i in info] == [0, 1, 2, 3]
        env.close()


@pytest.mark.gpu
def test_envpool_adapter(ctx):
    from ilswiss_amd.envs import EnvpoolEnv, get_envs
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    spec = dict(use_envpool=True, envpool_name="Hopper-v3", env_type="gym", env_name="hopper", env_kwargs={}, env_num=8,
                eval_env_seed=0, training_env_seed=4)   # sac_hopper_envpool.yaml:49-57
    env = get_envs(spec, ctx=ctx)
    plain = HipVectorEnv("hopper", 8, seed=4, ctx=ctx)
    assert isinstance(env, EnvpoolEnv) and len(env) == 8 and env.action_space[0].shape == (3,)
    o0 = env.reset()
    plain.reset()
    plain.set_state(*env.get_state())   # every vec env draws from its own Philox stream: copy the state to get a twin
    assert o0.shape == (8, 11)
    ids = np.array([1, 5, 6])
    act = np.random.default_rng(0).uniform(-1, 1, (3, 3)).astype(np.float32)
    o1, r1, d1, info = env.step(act, ids)
    o2, r2, d2, _ = plain.step(act, ids)
    np.testing.assert_array_equal(o1, o2); np.testing.assert_array_equal(r1, r2); np.testing.assert_array_equal(d1, d2)
    assert isinstance(info, list) and [i["env_id"] for i in info] == [1, 5, 6] and all(i["elapsed_step"] == 1 for i in info)
    assert all("players" not in i for i in info)
    env.step(act, ids)
    _, _, _, info = env.step(np.zeros((8, 3), np.float32))
    assert [int(i["elapsed_step"]) for i in info] == [1, 3, 1, 1, 1, 3, 3, 1]
    env.reset(np.array([5]))
    _, _, _, info = env.step(np.zeros((8, 3), np.float32))
    assert int(info[5]["elapsed_step"]) == 1 and int(info[1]["elapsed_step"]) == 4
    with pytest.raises(KeyError):
        get_envs(dict(spec, envpool_name="Swimmer-v3"), ctx=ctx)   # (Ant / Humanoid reach the 3-D steppers: test_envpool_adapter_3d_tasks)
    env.close(); plain.close()
