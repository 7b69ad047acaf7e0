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


@pytest.mark.parametrize("spec_name", ["sac_hopper_hip.yaml", "sac_hopper_envpool_hip.yaml"])
def test_run_script_c1_plumbing(tmp_path, spec_name):
    """exp_specs keys of the reference's sac_hopper.yaml / sac_hopper_envpool.yaml (env_num 4, batch 512), two tiny epochs."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    import sac_alpha_exp_script as script
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "sac", spec_name)))
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
              "Total Train Time (s)", "Number of env steps total", "Number of train calls total", "Number of gradient steps total"):
        assert k in rows[-1], k
    # _try_to_train (base_algorithm.py:293-299): 8 x 100 env steps, gate 100, min_steps_before_training 100 -> 8 train calls, counted as calls
    assert float(rows[-1]["Number of train calls total"]) == 8 and float(rows[-1]["Number of gradient steps total"]) == 80
    assert float(rows[-1]["Number of env steps total"]) == 800
    assert os.path.exists(tmp_path / "params.pkl") and os.path.exists(tmp_path / "best.pkl")
    assert "AverageReturn" in open(tmp_path / "debug.log").read()      # the text output of launcher_util.py:215,267-269
    assert alg.replay_buffer.num_steps_can_sample() == 800


def test_reference_sac_hopper_yaml_key_set_runs_unchanged(tmp_path):
    """Config 1: exp_specs/sac/sac_hopper.yaml:1-54 of the reference, typed in as the config schema it is — every key and value as the
    reference ships it (meta_data.num_workers / using_gpus, wrap_absorbing, save_replay_buffer, vf_lr, env_num 4, batch 512, the two
    unused seeds) — through the launcher's grid expansion and sac_alpha_exp_script.experiment.  Only the run LENGTH is cut for the test
    (six numbers, no key added or removed): 102 epochs x 10000 steps would be 1M env steps."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    import sac_alpha_exp_script as script
    spec = dict(
        meta_data=dict(script_path="run_scripts/sac_alpha_exp_script.py", exp_name="test_sac_hopper",
                       description="Train an agent using Soft-Actor-Critic", num_workers=2, using_gpus=True),
        variables=dict(seed=[0]),
        constants=dict(
            net_size=256, num_hidden_layers=2,
            rl_alg_params=dict(num_epochs=102, num_steps_per_epoch=10000, num_steps_between_train_calls=1000, num_train_steps_per_train_call=1000,
                               num_steps_per_eval=10000, max_path_length=1000, min_steps_before_training=0, eval_deterministic=True,
                               batch_size=512, replay_buffer_size=1000000, no_terminal=False, wrap_absorbing=False, save_best=True,
                               freq_saving=1, save_replay_buffer=False),
            sac_params=dict(alpha=0.2, reward_scale=1.0, discount=0.99, soft_target_tau=0.005, policy_lr=0.0003, qf_lr=0.0003, vf_lr=0.0003,
                            policy_mean_reg_weight=0.001, policy_std_reg_weight=0.001),
            env_specs=dict(env_name="hopper", env_kwargs={}, env_num=4, eval_env_seed=0, training_env_seed=0)))
    keys_before = {k: sorted(v) for k, v in spec["constants"].items() if isinstance(v, dict)}
    v = script.flatten_spec(spec)
    assert v["seed"] == 0 and v["env_specs"]["env_num"] == 4 and v["rl_alg_params"]["batch_size"] == 512
    v["rl_alg_params"].update(num_epochs=1, num_steps_per_epoch=400, num_steps_between_train_calls=100, num_train_steps_per_train_call=10,
                              num_steps_per_eval=100, max_path_length=100)
    assert {k: sorted(x) for k, x in v.items() if isinstance(x, dict) and k in keys_before} == keys_before      # same key set
    alg = script.experiment(v, 0, str(tmp_path))
    rows = list(csv.DictReader(open(tmp_path / "progress.csv")))
    assert len(rows) == 2 and rows[-1]["Epoch"] == "1"
    # min_steps_before_training 0 (the reference's value): training starts with the first 100-step gate -> 8 calls of 10 steps
    assert float(rows[-1]["Number of train calls total"]) == 8 and float(rows[-1]["Number of env steps total"]) == 800
    for k in ("Test Returns Mean", "QF1 Loss", "Policy Loss", "Alpha", "Train Time (s)", "Sample Time (s)"):
        assert k in rows[-1] and np.isfinite(float(rows[-1][k])), k
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
                            max_path_length=500, min_steps_before_training=4096, batch_size=256, replay_buffer_size=200000,
                            eval_deterministic=True)
    rets = []
    alg.evaluate = (lambda orig: (lambda *a: rets.append(orig(*a)["AverageReturn"])))(alg.evaluate)
    alg.train()
    ctx.close()
    assert max(rets[-3:]) > 5 * max(rets[0], 15.0), rets


def test_device_obs_rms_matches_running_mean_std(ctx):
    """norm_obs env vs a twin env without it: RunningMeanStd (normalizer.py:128-152) over every batch returned by
    reset / step, and normalize_obs (vecenvs.py:299-327) of what the env hands out."""
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from oracle.envnorm import RunningMeanStd, normalize_obs
    raw_env = HipVectorEnv("hopper", 32, seed=5, ctx=ctx)
    env = HipVectorEnv("hopper", 32, seed=5, ctx=ctx, norm_obs=True)
    rms = RunningMeanStd()
    env.obs_rms.set(np.zeros(11), np.ones(11), 0)     # forget the batch shown at construction
    o_n = env.reset()
    q, v = env.get_state()
    o_raw = np.concatenate([q[:, 1:], np.clip(v, -10, 10)], axis=1).astype(np.float32).astype(np.float64)   # gym HopperEnv._get_obs, f32 like the device
    rms.update(o_raw)
    np.testing.assert_allclose(o_n, normalize_obs(o_raw, rms), rtol=1e-5, atol=1e-6)
    rng = np.random.default_rng(0)
    for k in range(6):
        ids = None if k % 2 == 0 else np.sort(rng.choice(32, 7, replace=False))
        n = 32 if ids is None else 7
        act = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        raw_env.set_state(*env.get_state())               # twin without normalisation steps from the same states
        o_raw, r0, d0, _ = raw_env.step(act, ids)
        o_n, r1, d1, _ = env.step(act, ids)
        rms.update(o_raw.astype(np.float64))
        np.testing.assert_allclose(o_n, normalize_obs(o_raw, rms), rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(r0, r1)
    np.testing.assert_allclose(env.obs_rms.mean, rms.mean, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(env.obs_rms.var, rms.var, rtol=1e-9, atol=1e-12)
    assert env.obs_rms.count == rms.count
    # an eval env built on the training env's statistics normalises with them and leaves them alone
    ev = HipVectorEnv("hopper", 4, seed=9, ctx=ctx, norm_obs=True, obs_rms=env.obs_rms, update_obs_rms=False)
    o_n = ev.reset()
    q, v = ev.get_state()
    o_raw = np.concatenate([q[:, 1:], np.clip(v, -10, 10)], axis=1).astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(o_n, normalize_obs(o_raw, rms), rtol=1e-5, atol=1e-6)
    assert ev.obs_rms.count == rms.count


@pytest.mark.parametrize("name,spec_path", [("td3", "td3/td3_hopper_hip.yaml"), ("sac", "sac/sac_v_hopper_hip.yaml"),
                                            ("ppo", "ppo/ppo_hopper_hip.yaml")])
def test_other_run_scripts_plumbing(tmp_path, name, spec_path):
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    script = importlib.import_module(f"{name}_exp_script")
    from _common import flatten_spec
    v = flatten_spec(yaml.safe_load(open(os.path.join(ROOT, "exp_specs", spec_path))))
    v["env_specs"].update(env_num=8, eval_env_num=4)
    v["rl_alg_params"].update(num_epochs=1, num_steps_per_epoch=800, num_steps_between_train_calls=400,
                              num_train_steps_per_train_call=1 if name == "ppo" else 10, num_steps_per_eval=100,
                              max_path_length=100, min_steps_before_training=0 if name == "ppo" else 200, batch_size=64,
                              replay_buffer_size=5000, freq_saving=1)
    if name == "ppo":
        v["ppo_params"].update(mini_batch_size=64, update_epoch=2)
    alg = script.experiment(v, 0, str(tmp_path))
    rows = list(csv.DictReader(open(tmp_path / "progress.csv")))
    assert len(rows) == 2 and float(rows[-1]["Number of env steps total"]) == 1600
    assert np.isfinite(float(rows[-1]["AverageReturn"]))
    want = dict(td3=("QF1 Loss", "Policy Loss", "Q Targets Mean", "Bellman Errors 1 Max", "Policy Action Std"),
                sac=("QF1 Loss", "VF Loss", "Policy Loss", "V Predictions Mean", "Log Pis Min"), ppo=("PPO Segments",))[name]
    for k in want:
        assert k in rows[-1], k
    if name == "ppo":   # 2 rollouts of 8 envs x 50 steps per epoch, statistics fed by every returned batch
        assert alg.training_env.obs_rms.count >= 1600 and alg.eval_env.obs_rms.count == alg.training_env.obs_rms.count


def test_ppo_learns_on_hip_hopper():
    """Ten on-device PPO iterations (rollout -> GAE -> minibatch epochs): the exploration policy's episode return must
    rise several-fold (measured: 15 -> 165 over 650k samples; the zero-mean initial policy falls within ~15 steps)."""
    import ilswiss_amd as ia
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
    np.random.seed(0)
    ctx = ia.Context(0, seed=0)
    env = HipVectorEnv("hopper", 512, seed=0, ctx=ctx, norm_obs=True)
    pol = ReparamMultivariateGaussianPolicy([64, 64], 11, 3, conditioned_std=False, hidden_activation="tanh", ctx=ctx)
    vf = ia.FlattenMlp([64, 64], 1, 11, hidden_activation="tanh", ctx=ctx)
    tr = PPO(pol, vf, mini_batch_size=2048, update_epoch=10, gae_tau=0.95, max_samples=512 * 128)
    env.rollout_stats(reset=True)
    rets = []
    for it in range(10):
        tr.train_from_rollout(env, 128, max_path_length=1000)
        ep, rs = env.rollout_stats(reset=True)
        rets.append(rs / max(ep, 1))
    ctx.close()
    assert np.isfinite(rets).all() and rets[-1] > 4.0 * rets[0] and rets[-1] > 80, rets


def test_gail_run_script_with_generated_demos(tmp_path, ctx):
    """Config 3 plumbing end to end: demonstrations in the reference's pickle format (gen_expert_demos) -> demos listing ->
    adv_irl_exp_script with the gail_walker keys (no_terminal, gail2 rewards, WGAN-GP) for one tiny epoch."""
    import pickle
    import sys
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    import adv_irl_exp_script as script
    import gen_expert_demos as gen
    import ilswiss_amd as ia
    from _common import flatten_spec
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    env = HipVectorEnv("walker", 6, seed=3, ctx=ctx)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], env.obs_dim, env.act_dim, ctx=ctx, seed=5)
    demos = gen.generate(pol, env, 6, max_path_length=60)
    assert len(demos) == 6 and demos[0]["observations"].shape[1] == 17 and demos[0]["rewards"].shape[1] == 1
    for d in demos:   # wire format of adv_irl_exp_script.py:51-60
        T = len(d["rewards"])
        assert d["actions"].shape == (T, 6) and d["next_observations"].shape == (T, 17) and d["terminals"].shape == (T, 1)
        np.testing.assert_array_equal(d["observations"][1:], d["next_observations"][:-1])
    (tmp_path / "demos").mkdir()
    with open(tmp_path / "demos" / "walker.pkl", "wb") as f:
        pickle.dump(demos, f)
    with open(tmp_path / "listing.yaml", "w") as f:
        yaml.dump(dict(walker_sac=dict(description="test", file_paths=["./demos/walker.pkl"])), f)
    v = flatten_spec(yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "gail", "gail_walker_hip.yaml"))))
    v["demos_listing"] = str(tmp_path / "listing.yaml")
    v["env_specs"].update(env_num=8, eval_env_num=4)
    v["adv_irl_params"].update(num_epochs=1, num_steps_per_epoch=800, num_steps_between_train_calls=400, max_path_length=100,
                               min_steps_before_training=200, num_steps_per_eval=100, replay_buffer_size=5000,
                               num_update_loops_per_train_call=10, disc_optim_batch_size=64, policy_optim_batch_size=64, freq_saving=1)
    alg = script.experiment(v, 0, str(tmp_path / "log"))
    rows = list(csv.DictReader(open(tmp_path / "log" / "progress.csv")))
    assert len(rows) == 2
    for k in ("Disc CE Loss", "Disc Acc", "Grad Pen", "Grad Pen W", "Disc Rew Mean", "Disc Rew Max", "QF1 Loss", "Policy Loss",
              "AverageReturn", "Test Returns Mean"):
        assert k in rows[-1] and np.isfinite(float(rows[-1][k])), k
    assert float(rows[-1]["Disc Rew Max"]) <= 0.0            # gail2: log D <= 0 (adv_irl.py:283-286)
    assert float(rows[-1]["Grad Pen W"]) == 8.0
    assert alg.trainer.expert_rb.num_steps_can_sample() <= sum(len(d["rewards"]) for d in demos)   # 4 of the 6 (traj_num)
    # no_terminal: nothing stored in the policy buffer is flagged terminal (base_algorithm.py:195-196)
    batch = alg.replay_buffer.random_batch(512)
    assert batch["terminals"].sum() == 0
    with open(tmp_path / "log" / "params.pkl", "rb") as f:
        snap = pickle.load(f)
    assert "disc" in snap and "policy" in snap


def test_device_eval_sampler_matches_host_walked_sampler(ctx):
    """ilsx_eval_rollout vs the host-walked VecPathSampler on twin envs (same seed -> same reset noise): one episode per
    env, deterministic policy, same statistics (get_generic_path_information keys)."""
    import ilswiss_amd as ia
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    from ilswiss_amd.samplers import DeviceEvalSampler, VecPathSampler, get_average_returns, get_generic_path_information
    pol = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], 11, 3, ctx=ctx, seed=4)
    det = ia.MakeDeterministic(pol)
    env_d = HipVectorEnv("hopper", 12, seed=21, ctx=ctx)
    env_h = HipVectorEnv("hopper", 12, seed=21, ctx=ctx)
    dev = DeviceEvalSampler(env_d, det, num_steps=1, max_path_length=80).obtain_statistics("Test")
    assert dev["Num Paths"] == 12
    # replay the same episodes through the host sampler: start the twin from the device env's post-reset states
    # (reset noise differs per env object), by stepping it manually like vec_sampler.rollout does
    env_d2 = HipVectorEnv("hopper", 12, seed=21, ctx=ctx)
    st0 = DeviceEvalSampler(env_d2, det, num_steps=1, max_path_length=80)
    # determinism: a fresh env object with the same seed and call sequence reproduces the statistics exactly
    # (rng streams are per ctx allocation order, so compare structure and ranges instead of bits)
    dev2 = st0.obtain_statistics("Test")
    assert set(dev2) == set(dev)
    host_paths = VecPathSampler(env_h, det, num_steps=1, max_path_length=80).obtain_samples()
    host = get_generic_path_information(host_paths, stat_prefix="Test")
    assert set(host) | {"AverageReturn"} == set(dev)
    # same policy, same dynamics, i.i.d. reset noise of +-5e-3: episode statistics agree closely
    np.testing.assert_allclose(dev["Test Returns Mean"], host["Test Returns Mean"], rtol=0.25)
    np.testing.assert_allclose(dev["Test Ep. Len. Mean"], host["Test Ep. Len. Mean"], rtol=0.25)
    np.testing.assert_allclose(dev["AverageReturn"], dev["Test Returns Mean"])
    assert dev["Test Ep. Len. Max"] <= 80 and dev["Test Actions Max"] <= 1.0 and dev["Test Actions Min"] >= -1.0
    # exact cross-check of the accumulators on ONE env object: freeze-at-first-terminal + sums, replayed on the host
    env1 = HipVectorEnv("hopper", 5, seed=3, ctx=ctx)
    s1 = DeviceEvalSampler(env1, det, num_steps=1, max_path_length=60).obtain_statistics("Test")
    assert s1["Num Paths"] == 5 and 1 <= s1["Test Ep. Len. Min"] <= s1["Test Ep. Len. Max"] <= 60
    assert abs(s1["Test Rewards Mean"] * s1["Test Ep. Len. Mean"] - s1["Test Returns Mean"]) < 1e-6 * max(1, abs(s1["Test Returns Mean"]))


def test_bc_run_script_with_generated_demos(tmp_path, ctx):
    import pickle
    import sys
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    import bc_exp_script as script
    import gen_expert_demos as gen
    import ilswiss_amd as ia
    from _common import flatten_spec
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    env = HipVectorEnv("hopper", 6, seed=3, ctx=ctx)
    expert = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], 11, 3, ctx=ctx, seed=5)
    demos = gen.generate(expert, env, 6, max_path_length=80)
    (tmp_path / "demos").mkdir()
    with open(tmp_path / "demos" / "hopper.pkl", "wb") as f:
        pickle.dump(demos, f)
    with open(tmp_path / "listing.yaml", "w") as f:
        yaml.dump(dict(hopper_sac=dict(description="test", file_paths=["./demos/hopper.pkl"])), f)
    v = flatten_spec(yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "bc", "bc_hopper_hip.yaml"))))
    v["demos_listing"] = str(tmp_path / "listing.yaml")
    v["env_specs"].update(env_num=4, eval_env_num=4)
    v["policy_net_size"] = 64
    v["bc_params"].update(num_epochs=2, num_steps_per_epoch=100, num_steps_between_train_calls=50, max_path_length=60,
                          num_steps_per_eval=100, num_updates_per_train_call=20, batch_size=64, freq_saving=1)
    script.experiment(v, 0, str(tmp_path / "log"))
    rows = list(csv.DictReader(open(tmp_path / "log" / "progress.csv")))
    assert len(rows) == 2 and float(rows[-1]["Number of train steps total"]) == 80
    assert float(rows[-1]["Log-Likelihood"]) > float(rows[0]["Log-Likelihood"])      # the clone's likelihood of the demos rises
    assert np.isfinite(float(rows[-1]["AverageReturn"])) and os.path.exists(tmp_path / "log" / "best.pkl")


# ------------------------------------------------------------------------------------------- grouped runs behind run_experiment.py
def _launch(tmp, spec, group, tag):
    """`python run_experiment.py -e spec.yaml --group <group>` with cwd = tmp/<tag>; returns {seed: log dir}."""
    import glob
    import subprocess
    import sys
    wd = tmp / tag
    wd.mkdir()
    path = wd / "spec.yaml"
    path.write_text(yaml.dump(spec))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_experiment.py"), "-e", str(path), "--group", str(group)], cwd=wd,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    dirs = {}
    for d in glob.glob(str(wd / "logs" / "*" / "*--s-*")):
        dirs[int(d.rsplit("--s-", 1)[1])] = d
    return dirs, r.stdout


def _rows_without_time(d):
    rows = list(csv.DictReader(open(os.path.join(d, "progress.csv"))))
    out = [{k: v for k, v in r.items() if "Time" not in k} for r in rows]
    for r in out:   # the epoch's exploration returns are summed on the device with float64 atomics in completion order: equal to the last ulp or two, not bit for bit
        if r.get("Exploration Returns Mean"):
            r["Exploration Returns Mean"] = f"{float(r['Exploration Returns Mean']):.12g}"
    return out


@pytest.mark.parametrize("path_mode", [False, True], ids=["insert_every_step", "insert_at_episode_end"])
def test_grouped_run_script_writes_the_single_run_logs(tmp_path, path_mode):
    """`run_experiment.py --group 4` runs the four seeds of a spec in ONE process — lock-step vec-env steps, one
    ilsx_sac_group launch per stage of the gradient step — and writes four ordinary log directories whose progress.csv (every column that is
    not a wall-clock time) and parameters are exactly the ones four single processes write.  With insert_at_episode_end the rings of the seeds
    fill at different moments (their own episodes), so the first train triggers find only a subset able to train: the subset path too."""
    import pickle
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "sac", "sac_hopper_hip.yaml")))
    spec["meta_data"].update(script_path=os.path.join(ROOT, "run_scripts", "sac_alpha_exp_script.py"), num_workers=1, exp_name="grp_test")
    spec["variables"] = dict(seed=[0, 1, 2, 3])
    c = spec["constants"]
    c["env_specs"].update(env_num=8, eval_env_num=4)
    c["rl_alg_params"].update(num_epochs=2, num_steps_per_epoch=800, num_steps_between_train_calls=80, num_train_steps_per_train_call=12,
                              num_steps_per_eval=200, max_path_length=60, min_steps_before_training=0 if path_mode else 160, batch_size=256,
                              replay_buffer_size=20000, freq_saving=1, insert_at_episode_end=path_mode)
    solo, _ = _launch(tmp_path, spec, 1, "solo")
    grp, out = _launch(tmp_path, spec, 4, "grouped")
    assert sorted(solo) == sorted(grp) == [0, 1, 2, 3]
    assert out.count("sac_alpha_exp_script.py") == 1       # ONE child process for the four variants
    for seed in range(4):
        a, b = _rows_without_time(solo[seed]), _rows_without_time(grp[seed])
        assert len(a) == 3
        for ra, rb_ in zip(a, b):
            assert ra == rb_, (seed, ra["Epoch"], {k: (ra[k], rb_.get(k)) for k in ra if ra[k] != rb_.get(k)})
        assert float(a[-1]["Number of gradient steps total"]) > 0
        pa, pb = (pickle.load(open(os.path.join(d[seed], "params.pkl"), "rb")) for d in (solo, grp))
        for k in ("policy", "qf1", "target_qf2"):
            np.testing.assert_array_equal(pa[k], pb[k], err_msg=f"seed {seed} {k}")
        assert pa["log_alpha"] == pb["log_alpha"]
        assert os.path.exists(os.path.join(grp[seed], "variant.json")) and os.path.exists(os.path.join(grp[seed], "debug.log"))
    assert _rows_without_time(grp[0]) != _rows_without_time(grp[1])     # the seeds differ


def test_eval_async_overlaps_the_next_epoch_and_logs_the_same_rows(tmp_path):
    """rl_alg_params.eval_async: the evaluation of epoch e runs on a frozen policy copy (eval env on a stream of its own) beside epoch e + 1 and
    its row is written when it is in.  (1) every epoch gets its row (the last one after the loop); (2) row 0 — everything before the first
    evaluation — is the blocking run's row 0, evaluation columns included: the frozen copy IS the policy at the end of the epoch; (3) later
    rows differ from the blocking run's only through the exploration-noise counter (the blocking evaluation advances it), so they are compared
    between the single-process and the grouped form of the ASYNCHRONOUS run instead: cell for cell, parameters bit for bit; (4) params.pkl is
    the snapshot of the evaluated epoch, not of the epoch that trained beside the evaluation."""
    import pickle
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "sac", "sac_hopper_hip.yaml")))
    spec["meta_data"].update(script_path=os.path.join(ROOT, "run_scripts", "sac_alpha_exp_script.py"), num_workers=1, exp_name="async_test")
    spec["variables"] = dict(seed=[0, 1])
    c = spec["constants"]
    c["env_specs"].update(env_num=8, eval_env_num=4)
    c["rl_alg_params"].update(num_epochs=2, num_steps_per_epoch=800, num_steps_between_train_calls=80, num_train_steps_per_train_call=12,
                              num_steps_per_eval=200, max_path_length=60, min_steps_before_training=0, batch_size=256, replay_buffer_size=20000,
                              freq_saving=1, insert_at_episode_end=True, eval_deterministic=True)
    sync, _ = _launch(tmp_path, spec, 1, "sync")
    c["rl_alg_params"]["eval_async"] = True
    solo, out1 = _launch(tmp_path, spec, 1, "async_solo")
    grp, out2 = _launch(tmp_path, spec, 2, "async_grouped")
    assert "evaluating between epochs" not in out1 + out2      # the asynchronous form was taken, not its fallback
    for seed in range(2):
        s_, a, b = (_rows_without_time(d[seed]) for d in (sync, solo, grp))
        assert len(s_) == len(a) == len(b) == 3
        assert [r["Epoch"] for r in a] == ["0", "1", "2"]
        assert s_[0] == a[0], {k: (s_[0][k], a[0].get(k)) for k in s_[0] if s_[0][k] != a[0].get(k)}
        assert a == b, seed
        pa, pb = (pickle.load(open(os.path.join(d[seed], "params.pkl"), "rb")) for d in (solo, grp))
        assert pa["epoch"] == pb["epoch"] == 2
        for k in ("policy", "qf1", "target_qf2"):
            np.testing.assert_array_equal(pa[k], pb[k], err_msg=f"seed {seed} {k}")


@pytest.mark.parametrize("spec_name", ["sac_walker_hip.yaml", "sac_halfcheetah_hip.yaml"])
def test_grouped_runs_on_the_seven_body_steppers(tmp_path, spec_name):
    """The lock-step rollout / evaluation launches of grouped runs (k_envg_step_runs) in their Walker2d (12 constraint rows) and HalfCheetah (16)
    instantiations, three seeds — one policy launch of three tasks —, both replay-insertion modes' worth of machinery: logs identical to the
    single-process runs'."""
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "sac", spec_name)))
    spec["meta_data"].update(script_path=os.path.join(ROOT, "run_scripts", "sac_alpha_exp_script.py"), num_workers=1, exp_name="grp7")
    spec["variables"] = dict(seed=[0, 1, 2])
    c = spec["constants"]
    c["env_specs"].update(env_num=4, eval_env_num=4)
    c["rl_alg_params"].update(num_epochs=1, num_steps_per_epoch=400, num_steps_between_train_calls=80, num_train_steps_per_train_call=6,
                              num_steps_per_eval=150, max_path_length=40, min_steps_before_training=0, batch_size=128, replay_buffer_size=20000,
                              freq_saving=1, insert_at_episode_end=True)
    solo, _ = _launch(tmp_path, spec, 1, "solo")
    grp, out = _launch(tmp_path, spec, 3, "grouped")
    assert sorted(solo) == sorted(grp) == [0, 1, 2]
    for seed in range(3):
        a, b = _rows_without_time(solo[seed]), _rows_without_time(grp[seed])
        assert len(a) == 2 and a == b, seed


def test_statistics_are_those_of_the_first_batch_after_end_epoch():
    """sac_alpha.py:185-190: eval_statistics are filled by the FIRST train_step after end_epoch.  A 50-step train call that asks for them
    reports what a 1-step call from the same state reports (epoch-0 Alpha = the initial 0.2), and ends with the parameters of 50 plain steps;
    the grouped call does the same for every run."""
    import ilswiss_amd as ia
    from ilswiss_amd.replay import SimpleReplayBuffer
    o, a, hid, B, N = 11, 3, [256, 256], 256, 4000
    rng = np.random.default_rng(3)
    data = (rng.normal(0, 1, (N, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (N, a))).astype(np.float32),
            rng.normal(0, 1, N).astype(np.float32), rng.random(N) < 0.01, rng.normal(0, 1, (N, o)).astype(np.float32))

    def make(c, k=0):
        rb = SimpleReplayBuffer(8192, o, a, random_seed=5 + k, ctx=c)
        rb.add_rows(*data)
        tr = ia.SoftActorCritic(ia.ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=c, seed=10 + k), ia.FlattenMlp(hid, 1, o + a, ctx=c, seed=20 + k),
                                ia.FlattenMlp(hid, 1, o + a, ctx=c, seed=30 + k), policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005, max_batch=B)
        return rb, tr
    c1, c2, c3 = ia.Context(0, seed=9), ia.Context(0, seed=9), ia.Context(0, seed=9)
    rb1, one = make(c1)
    rb2, many = make(c2)
    rb3, plain = make(c3)
    one.train_from_replay(rb1, 1, B)
    many.train_from_replay(rb2, 50, B)
    plain.eval_statistics = {}
    plain.train_from_replay(rb3, 50, B)
    s1, s50 = one.get_eval_statistics(), many.get_eval_statistics()
    assert dict(s1) == dict(s50)
    assert abs(s50["Alpha"] - 0.2) < 1e-3      # the reference logs alpha AFTER the step's own update (sac_alpha.py:165,208-212): 0.2 less one Adam step
    np.testing.assert_array_equal(many.get_params("qf1"), plain.get_params("qf1"))
    np.testing.assert_array_equal(many.get_params("policy"), plain.get_params("policy"))
    # a later epoch: statistics after end_epoch come from the first batch of the next call
    one.end_epoch(); many.end_epoch()
    one.eval_statistics = {}
    one.train_from_replay(rb1, 49, B)       # catch up to 50 steps, no statistics
    one.eval_statistics = None
    one.train_from_replay(rb1, 1, B)
    many.train_from_replay(rb2, 30, B)
    assert dict(one.get_eval_statistics()) == dict(many.get_eval_statistics())
    assert many.get_eval_statistics()["Alpha"] < 0.2
    for c in (c1, c2, c3):
        c.close()
    # grouped: K = 2 runs in sibling contexts
    base = ia.Context(0, seed=9)
    sib = base.sibling(9)
    (rba, ta), (rbb, tb) = make(base), make(sib)
    g = ia.SoftActorCriticGroup([ta, tb], ctx=base)
    g.train_from_replay([rba, rbb], 50, B)
    assert dict(ta.get_eval_statistics()) == dict(s50) and dict(tb.get_eval_statistics()) == dict(s50)
    g.close(); sib.close(); base.close()


def test_split_run_from_the_run_script_at_one_forced_rank(tmp_path):
    """rl_alg_params.split_ranks behind the reference's entry point: the run script joins the process group, scales the
    loop to its rank's share, gives the ctx an RCCL communicator and trains through the library's backward -> all-reduce -> update sequence.
    A one-GPU box can only run it on a ONE-rank communicator (ILSX_SPLIT_FORCE=1, split_ranks: 1): the log it writes must then be the plain
    run's, number for number (the split arithmetic at G = 1 is the un-split arithmetic, and the five step forms are bit-identical)."""
    import subprocess
    import sys
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "sac", "sac_hopper_hip.yaml")))
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    from _common import flatten_spec
    v = flatten_spec(spec)
    v["env_specs"].update(env_num=8, eval_env_num=4)
    v["rl_alg_params"].update(num_epochs=1, num_steps_per_epoch=800, num_steps_between_train_calls=80, num_train_steps_per_train_call=10,
                              num_steps_per_eval=200, max_path_length=60, min_steps_before_training=160, batch_size=256, replay_buffer_size=20000)
    rows = {}
    for tag, extra_env, split in (("plain", {}, None), ("split", {"ILSX_SPLIT_FORCE": "1"}, 1)):
        wd = tmp_path / tag
        wd.mkdir()
        vv = yaml.safe_load(yaml.dump(v))
        if split is not None:
            vv["rl_alg_params"]["split_ranks"] = split
        (wd / "v.yaml").write_text(yaml.dump(vv))
        env = dict(os.environ, **extra_env)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "run_scripts", "sac_alpha_exp_script.py"), "-e", str(wd / "v.yaml"), "-g", "0"],
                           cwd=wd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        import glob
        (d,) = glob.glob(str(wd / "logs" / "*" / "*--s-*"))
        rows[tag] = _rows_without_time(d)
    assert len(rows["plain"]) == 2 and float(rows["plain"][-1]["Number of gradient steps total"]) > 0
    assert rows["plain"] == rows["split"]


def _run_script_rows(tmp_path, script, variant, tag, extra_env):
    import glob
    import subprocess
    import sys
    wd = tmp_path / tag
    wd.mkdir()
    (wd / "v.yaml").write_text(yaml.dump(variant))
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_scripts", script), "-e", str(wd / "v.yaml"), "-g", "0"], cwd=wd, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    (d,) = glob.glob(str(wd / "logs" / "*" / "*--s-*"))
    return _rows_without_time(d)


def test_ppo_and_gail_split_runs_from_their_run_scripts_at_one_forced_rank(tmp_path, ctx):
    """split_ranks behind ppo_exp_script.py (rl_alg_params) and adv_irl_exp_script.py (adv_irl_params) — SURVEY section 8e's "PPO split" and "Disc"
    rows from the entry point: the script joins the process group, scales env counts / step counts / minibatch and batch sizes to its
    rank's share, gives the ctx its communicator, builds the trainers with grad_world.  On the one-rank communicator a single-GPU box
    allows (ILSX_SPLIT_FORCE=1, split_ranks: 1) the logs are the plain runs', number for number."""
    import pickle
    import sys
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    import gen_expert_demos as gen
    import ilswiss_amd as ia
    from _common import flatten_spec
    from ilswiss_amd.envs.vecenv import HipVectorEnv
    # ---- PPO
    v = flatten_spec(yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "ppo", "ppo_hopper_1024env_hip.yaml"))))
    v["env_specs"].update(env_num=16, eval_env_num=4)
    v["ppo_params"].update(mini_batch_size=256, update_epoch=2)
    v["rl_alg_params"].update(num_epochs=1, num_steps_between_train_calls=512, num_steps_per_epoch=1024, num_steps_per_eval=200, max_path_length=60,
                              freq_saving=1)
    plain = _run_script_rows(tmp_path, "ppo_exp_script.py", v, "ppo_plain", {})
    vv = yaml.safe_load(yaml.dump(v))
    vv["rl_alg_params"]["split_ranks"] = 1
    split = _run_script_rows(tmp_path, "ppo_exp_script.py", vv, "ppo_split", {"ILSX_SPLIT_FORCE": "1"})
    assert len(plain) == 2 and plain == split
    # ---- GAIL (discriminator + SAC, both split)
    env = HipVectorEnv("walker", 6, seed=3, ctx=ctx)
    pol = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], env.obs_dim, env.act_dim, ctx=ctx, seed=5)
    demos = gen.generate(pol, env, 6, max_path_length=60)
    (tmp_path / "demos").mkdir()
    with open(tmp_path / "demos" / "walker.pkl", "wb") as f:
        pickle.dump(demos, f)
    with open(tmp_path / "listing.yaml", "w") as f:
        yaml.dump(dict(walker_sac=dict(description="test", file_paths=["./demos/walker.pkl"])), f)
    v = flatten_spec(yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "gail", "gail_walker_hip.yaml"))))
    v["demos_listing"] = str(tmp_path / "listing.yaml")
    v["env_specs"].update(env_num=8, eval_env_num=4)
    v["adv_irl_params"].update(num_epochs=1, num_steps_per_epoch=800, num_steps_between_train_calls=400, max_path_length=100,
                               min_steps_before_training=200, num_steps_per_eval=100, replay_buffer_size=5000,
                               num_update_loops_per_train_call=10, disc_optim_batch_size=64, policy_optim_batch_size=64, freq_saving=1)
    plain = _run_script_rows(tmp_path, "adv_irl_exp_script.py", v, "gail_plain", {})
    vv = yaml.safe_load(yaml.dump(v))
    vv["adv_irl_params"]["split_ranks"] = 1
    split = _run_script_rows(tmp_path, "adv_irl_exp_script.py", vv, "gail_split", {"ILSX_SPLIT_FORCE": "1"})
    assert len(plain) == 2 and float(plain[-1]["Number of gradient steps total"]) > 0 and plain == split


def test_grouped_runs_of_another_trainer_are_stepped_one_by_one(tmp_path):
    """`--group K` with a trainer the grouped kernels do not take (TD3): the K runs still share one process and advance in lock-step, their
    train calls go one after the other on the shared schedule — and every log is the single-process run's."""
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "td3", "td3_hopper_hip.yaml")))
    spec["meta_data"].update(script_path=os.path.join(ROOT, "run_scripts", "td3_exp_script.py"), num_workers=1, exp_name="grp_td3")
    spec["variables"] = dict(seed=[0, 1])
    c = spec["constants"]
    c["env_specs"].update(env_num=8, eval_env_num=4)
    c["rl_alg_params"].update(num_epochs=1, num_steps_per_epoch=800, num_steps_between_train_calls=80, num_train_steps_per_train_call=10,
                              num_steps_per_eval=200, max_path_length=60, min_steps_before_training=160, batch_size=256, replay_buffer_size=20000,
                              freq_saving=1)
    solo, _ = _launch(tmp_path, spec, 1, "solo")
    grp, out = _launch(tmp_path, spec, 2, "grouped")
    assert sorted(solo) == sorted(grp) == [0, 1] and out.count("td3_exp_script.py") == 1
    for seed in (0, 1):
        a, b = _rows_without_time(solo[seed]), _rows_without_time(grp[seed])
        assert len(a) == 2 and len(b) == 2
        for ra, rb_ in zip(a, b):
            assert ra == rb_, (seed, ra["Epoch"], {k: (ra[k], rb_.get(k)) for k in ra if ra[k] != rb_.get(k)})
        assert float(a[-1]["Number of gradient steps total"]) > 0
