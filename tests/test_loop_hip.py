"""GPU suite: the outer loop, eval sampler and run script end to end (config C1 plumbing: the reference's
sac_hopper.yaml keys, env_num 4) + a short learning check on the HIP Hopper."""
import csv
import os

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vec_path_sampler_one_episode_per_env(ctx):
    import ilswiss_amd as ia
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from ilswiss_amd.samplers import VecPathSampler, get_generic_path_information
    env = HipVectorEnv("hopper", 6, seed=2, ctx=ctx)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], 11, 3, ctx=ctx, seed=4)
    s = VecPathSampler(env, ia.MakeDeterministic(pol), num_steps=60, max_path_length=50)
    paths = s.obtain_samples()
    assert len(paths) % 6 == 0 and sum(len(p) for p in paths) >= 60
    for p in paths:
        t = np.concatenate(p["terminals"])
        assert len(p) <= 50 and not t[:-1].any()          # a path ends at its first terminal or at the horizon
        assert np.asarray(p["observations"]).shape == (len(p), 11)
        np.testing.assert_array_equal(np.asarray(p["observations"])[1:], np.asarray(p["next_observations"])[:-1])
    st = get_generic_path_information(paths, stat_prefix="Test")
    for k in ("Test Returns Mean", "Test Rewards Mean", "Test Ep. Len. Mean", "Test Actions Max", "Num Paths"):
        assert k in st


def test_run_script_c1_plumbing(tmp_path):
    """exp_specs keys of the reference's sac_hopper.yaml (env_num 4, batch 512), two tiny epochs."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    import sac_alpha_exp_script as script
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "sac", "sac_hopper_hip.yaml")))
    v = script.flatten_spec(spec)
    v["env_specs"]["env_num"] = 4
    v["rl_alg_params"].update(num_epochs=1, num_steps_per_epoch=400, num_steps_between_train_calls=100,
                              num_train_steps_per_train_call=10, num_steps_per_eval=100, max_path_length=100,
                              min_steps_before_training=100, batch_size=512, replay_buffer_size=5000, freq_saving=1)
    v["sac_params"]["vf_lr"] = 3e-4  # ignored key, swallowed like sac_alpha.py:39
    alg = script.experiment(v, 0, str(tmp_path))
    rows = list(csv.DictReader(open(tmp_path / "progress.csv")))
    assert len(rows) == 2 and rows[-1]["Epoch"] == "1"
    for k in ("Test Returns Mean", "AverageReturn", "QF1 Loss", "QF2 Loss", "Policy Loss", "Alpha Loss", "Alpha",
              "Q1 Predictions Mean", "Log Pis Mean", "Train Time (s)", "Sample Time (s)", "Epoch Time (s)",
              "Total Train Time (s)", "Number of env steps total", "Number of train steps total"):
        assert k in rows[-1], k
    assert float(rows[-1]["Number of env steps total"]) == 800
    assert os.path.exists(tmp_path / "params.pkl") and os.path.exists(tmp_path / "best.pkl")
    assert alg.replay_buffer.num_steps_can_sample() == 800


def test_sac_learns_on_hip_hopper():
    """A few thousand gradient steps must lift the deterministic policy well above the random policy."""
    import ilswiss_amd as ia
    from ilswiss_amd.algorithm import DeviceRLAlgorithm
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    np.random.seed(0)
    ctx = ia.Context(0, seed=0)
    tr_env, ev_env = HipVectorEnv("hopper", 1024, seed=0, ctx=ctx), HipVectorEnv("hopper", 16, seed=99, ctx=ctx)
    env = tr_env.single_env_view()
    pol = ia.ReparamTanhMultivariateGaussianPolicy([256, 256], 11, 3, ctx=ctx)
    q1, q2 = ia.FlattenMlp([256, 256], 1, 14, ctx=ctx), ia.FlattenMlp([256, 256], 1, 14, ctx=ctx)
    tr = ia.SoftActorCritic(pol, q1, q2, env=env, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=256)
    alg = DeviceRLAlgorithm(tr, env, tr_env, ev_env, pol, num_epochs=7, num_steps_per_epoch=20480,
                            num_steps_between_train_calls=1024, num_train_steps_per_train_call=250, num_steps_per_eval=1000,
                            max_path_length=500, min_steps_before_training=4096, batch_size=256, replay_buffer_size=200000)
    rets = []
    alg.evaluate = (lambda orig: (lambda *a: rets.append(orig(*a)["AverageReturn"])))(alg.evaluate)
    alg.train()
    ctx.close()
    assert max(rets[-3:]) > 5 * max(rets[0], 15.0), rets
