"""GPU suite: get_snapshot -> load_snapshot into a FRESH trainer, then both take the same further steps and must stay
bit-identical (parameters, targets, optimiser moments + step counts all restored) — the load_snapshot halves of
td3.py:198-206, sac.py:259-270, ppo.py, adv_irl.py and bc.py, and the `load_params` resume path of the run scripts
(rlkit/core/logger.py:31-49, run_scripts/sac_alpha_exp_script.py:106-113,142-146).  Pattern: test_hip_parity.py::test_snapshot_roundtrip."""
import csv
import os
import pickle
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(rng, B, o, a):
    return dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32), actions=np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                rewards=rng.normal(0, 1, (B, 1)).astype(np.float32), terminals=(rng.random((B, 1)) < 0.1).astype(np.float32),
                next_observations=rng.normal(0, 1, (B, o)).astype(np.float32))


def test_td3_snapshot_roundtrip(ctx):
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.td3 import TD3, MlpGaussianNoisePolicy
    rng = np.random.default_rng(3)
    o, a, hid, B = 11, 3, [64, 64], 32

    def make(seed):
        pol = MlpGaussianNoisePolicy(hid, o, a, policy_noise=0.2, policy_noise_clip=0.5, output_activation="tanh", ctx=ctx, seed=seed)
        return TD3(pol, FlattenMlp(hid, 1, o + a, ctx=ctx, seed=seed + 1), FlattenMlp(hid, 1, o + a, ctx=ctx, seed=seed + 2),
                   max_batch=B, policy_lr=3e-4, qf_lr=3e-4)
    t1, t2 = make(1), make(50)
    data = [(_batch(rng, B, o, a), rng.normal(0, 1, (B, a)).astype(np.float32)) for _ in range(7)]
    for b, e in data[:3]:      # an odd number of steps: the delayed policy update's parity must be restored too
        t1.train_step(b, eps_target=e)
    t2.load_snapshot(pickle.loads(pickle.dumps(t1.get_snapshot())))
    for b, e in data[3:]:
        t1.train_step(b, eps_target=e)
        t2.train_step(b, eps_target=e)
    for k in ("pi", "q1", "q2", "tpi", "tq1", "tq2"):
        np.testing.assert_array_equal(t1.get_flat_params(k), t2.get_flat_params(k), err_msg=k)
    s1, s2 = t1.get_snapshot(), t2.get_snapshot()
    assert s1["policy_optimizer"]["step"] == s2["policy_optimizer"]["step"] == 4      # steps 0, 2, 4, 6
    assert s1["qf1_optimizer"]["step"] == 7 and s1["qf1_optimizer"]["n_train_steps"] == 7
    np.testing.assert_array_equal(s1["qf2_optimizer"]["exp_avg_sq"], s2["qf2_optimizer"]["exp_avg_sq"])


def test_sac_v_snapshot_roundtrip(ctx):
    import ilswiss_amd as ia
    from ilswiss_amd.sac_v import SoftActorCriticV
    rng = np.random.default_rng(4)
    o, a, hid, B = 11, 3, [64, 64], 32

    def make(seed):
        pol = ia.ReparamTanhMultivariateGaussianPolicy(hid, o, a, ctx=ctx, seed=seed)
        nets = [ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=seed + 1), ia.FlattenMlp(hid, 1, o + a, ctx=ctx, seed=seed + 2),
                ia.FlattenMlp(hid, 1, o, ctx=ctx, seed=seed + 3)]
        return SoftActorCriticV(pol, *nets, max_batch=B, policy_lr=3e-4, qf_lr=3e-4, vf_lr=3e-4, alpha=0.2)
    t1, t2 = make(1), make(60)
    data = [(_batch(rng, B, o, a), rng.normal(0, 1, (B, a)).astype(np.float32)) for _ in range(5)]
    for b, e in data[:2]:
        t1.train_step(b, eps=e)
    t2.load_snapshot(pickle.loads(pickle.dumps(t1.get_snapshot())))
    for b, e in data[2:]:
        t1.train_step(b, eps=e)
        t2.train_step(b, eps=e)
    for k in ("qf1", "qf2", "policy", "vf", "target_vf"):
        np.testing.assert_array_equal(t1.get_flat_params(k), t2.get_flat_params(k), err_msg=k)
    assert t1.get_snapshot()["vf_optimizer"]["step"] == 5


def test_ppo_snapshot_roundtrip(ctx):
    from ilswiss_amd.networks import FlattenMlp
    from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
    rng = np.random.default_rng(5)
    o, a, hid = 11, 3, [64, 64]

    def make(seed):
        pol = ReparamMultivariateGaussianPolicy(hid, o, a, conditioned_std=False, hidden_activation="tanh", ctx=ctx, seed=seed)
        vf = FlattenMlp(hid, 1, o, hidden_activation="tanh", ctx=ctx, seed=seed + 1)
        return PPO(pol, vf, max_samples=2048, mini_batch_size=64, update_epoch=2, policy_lr=3e-4, value_lr=3e-4)
    t1, t2 = make(1), make(70)

    def trajs():
        return [dict(observations=rng.normal(0, 1, (L, o)).astype(np.float32), actions=rng.normal(0, 0.7, (L, a)).astype(np.float32),
                     rewards=rng.normal(0.5, 1.0, (L, 1)).astype(np.float32)) for L in (40, 100, 9)]
    t1.train_step(trajs())                      # library-drawn shuffles: the shuffle counter is part of the state
    t2.load_snapshot(pickle.loads(pickle.dumps(t1.get_snapshot())))
    for _ in range(2):
        tj = trajs()
        t1.train_step(tj)
        t2.train_step(tj)
    np.testing.assert_array_equal(t1.get_flat_params(0), t2.get_flat_params(0))
    np.testing.assert_array_equal(t1.get_flat_params(1), t2.get_flat_params(1))
    s = t1.get_snapshot()
    assert s["policy_optimizer"]["step"] == s["vf_optimizer"]["step"] == 3 * 2 * 3   # 3 calls x 2 epochs x ceil(149/64) minibatches
    assert s["policy_optimizer"]["exp_avg"].size == s["policy"].size                 # mean net | action_log_std


def test_disc_and_bc_snapshot_roundtrip(ctx):
    import ilswiss_amd as ia
    from ilswiss_amd.adv_irl import MLPDisc
    from ilswiss_amd.bc import BC
    from ilswiss_amd.snapshot import get_opt, set_opt
    rng = np.random.default_rng(6)
    o, a, B = 17, 6, 64
    d1 = MLPDisc(o + a, hid_dim=128, hid_act="tanh", use_bn=False, ctx=ctx, seed=1).bind(o, max_batch=B, disc_lr=3e-4, disc_momentum=0.9)
    d2 = MLPDisc(o + a, hid_dim=128, hid_act="tanh", use_bn=False, ctx=ctx, seed=2).bind(o, max_batch=B, disc_lr=3e-4, disc_momentum=0.9)

    def rows():
        return (rng.normal(0, 1, (B, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                rng.normal(0.3, 1, (B, o)).astype(np.float32), np.tanh(rng.normal(0, 1, (B, a))).astype(np.float32),
                rng.random((B, 1)).astype(np.float32))
    for _ in range(2):
        r = rows()
        d1.train_step(*r[:4], eps=r[4])
    d2.set_flat_params(d1.get_flat_params())
    set_opt(ctx.lib, "disc", d2.h, get_opt(ctx.lib, "disc", d1.h, d1.num_params))
    for _ in range(2):
        r = rows()
        d1.train_step(*r[:4], eps=r[4])
        d2.train_step(*r[:4], eps=r[4])
    np.testing.assert_array_equal(d1.get_flat_params(), d2.get_flat_params())
    assert get_opt(ctx.lib, "disc", d2.h, d2.num_params)["step"] == 4
    # behaviour cloning
    p1 = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], o, a, ctx=ctx, seed=1)
    p2 = ia.ReparamTanhMultivariateGaussianPolicy([64, 64], o, a, ctx=ctx, seed=9)
    b1, b2 = BC("MLE", p1, batch_size=B, lr=1e-3, momentum=0.5), BC("MLE", p2, batch_size=B, lr=1e-3, momentum=0.5)

    def bcb():
        return dict(observations=rng.normal(0, 1, (B, o)).astype(np.float32), actions=np.tanh(rng.normal(0, 0.5, (B, a))).astype(np.float32))
    b1.train_step(bcb())
    b2.load_snapshot(b1.get_snapshot())
    for _ in range(2):
        x = bcb()
        b1.train_step(x)
        b2.train_step(x)
    np.testing.assert_array_equal(p1.get_flat_params(), p2.get_flat_params())


def test_load_params_resumes_a_run(tmp_path, ctx):
    """Run the SAC script for 2 epochs, then resume from its log directory with `load_params` for 2 more: progress.csv keeps
    growing in the same directory, counters continue, the replay buffer comes back (save_replay_buffer), and the restored
    trainer state is the saved one."""
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "run_scripts"))
    import sac_alpha_exp_script as script
    from ilswiss_amd.snapshot import load_from_file
    spec = yaml.safe_load(open(os.path.join(ROOT, "exp_specs", "sac", "sac_hopper_hip.yaml")))
    v = script.flatten_spec(spec)
    v["env_specs"]["env_num"] = 4
    v["rl_alg_params"].update(num_epochs=1, num_steps_per_epoch=400, num_steps_between_train_calls=100, num_train_steps_per_train_call=10,
                              num_steps_per_eval=100, max_path_length=100, min_steps_before_training=100, batch_size=64,
                              replay_buffer_size=5000, freq_saving=1, save_replay_buffer=True)
    log = str(tmp_path / "run")
    alg = script.experiment(v, 0, log)
    saved = pickle.load(open(os.path.join(log, "params.pkl"), "rb"))
    extra = pickle.load(open(os.path.join(log, "extra_data.pkl"), "rb"))
    assert extra["epoch"] == 1 and extra["_n_env_steps_total"] == 800 and len(extra["replay_buffer"]["rewards"]) == 800
    np.testing.assert_array_equal(saved["qf1"], alg.trainer.get_params("qf1"))
    v2 = dict(v, load_params=dict(load_replay_buffer=True, load_model=True, load_path=log))
    v2["rl_alg_params"] = dict(v["rl_alg_params"], num_epochs=3)
    alg2 = script.experiment(v2, 0, log)
    rows = list(csv.DictReader(open(os.path.join(log, "progress.csv"))))
    assert [r["Epoch"] for r in rows] == ["0", "1", "2", "3"]
    assert float(rows[-1]["Number of env steps total"]) == 1600 and float(rows[-1]["Number of train calls total"]) == 16
    assert alg2.replay_buffer.num_steps_can_sample() == 1600
    # load_from_file on a fresh algorithm object restores exactly what was saved
    v3 = dict(v2)
    v3["rl_alg_params"] = dict(v["rl_alg_params"], num_epochs=-1)      # range(start, 0): constructs, restores, trains nothing
    saved3 = pickle.load(open(os.path.join(log, "params.pkl"), "rb"))
    alg3 = script.experiment(v3, 0, str(tmp_path / "other"))
    for k in ("policy", "qf1", "target_qf2"):
        np.testing.assert_array_equal(alg3.trainer.get_params(k), saved3[k])
    assert alg3.trainer.get_snapshot()["qf1_optimizer"]["step"] == saved3["qf1_optimizer"]["step"] == 160
