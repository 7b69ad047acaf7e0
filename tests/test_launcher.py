"""CPU suite: exp_spec -> variant grid (run_experiment.py:25-45 contract) and the spec files shipped in exp_specs/."""
import glob
import os

import pytest

import numpy as np
import yaml

from ilswiss_amd.launcher import variants

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nested_variables_expand_to_the_cartesian_grid():
    spec = dict(meta_data=dict(script_path="run_scripts/x.py", exp_name="x", num_workers=2),
                variables=dict(seed=[0, 1, 2], sac_params=dict(reward_scale=[2.0, 4.0]), adv_irl_params=dict(grad_pen_weight=[8.0])),
                constants=dict(net_size=256, sac_params=dict(discount=0.99), adv_irl_params=dict(mode="gail2")))
    vs = list(variants(spec))
    assert len(vs) == 6 and [v["exp_id"] for v in vs] == list(range(6))
    assert {(v["seed"], v["sac_params"]["reward_scale"]) for v in vs} == {(s, r) for s in (0, 1, 2) for r in (2.0, 4.0)}
    for v in vs:
        assert v["sac_params"]["discount"] == 0.99 and v["adv_irl_params"] == dict(mode="gail2", grad_pen_weight=8.0)
        assert v["script_path"] == "run_scripts/x.py" and v["net_size"] == 256
    assert spec["constants"]["sac_params"] == dict(discount=0.99)   # the spec itself is left alone


def test_shipped_specs_name_existing_scripts_and_expand():
    specs = glob.glob(os.path.join(ROOT, "exp_specs", "*", "*.yaml"))
    assert len(specs) >= 5
    for path in specs:
        spec = yaml.safe_load(open(path))
        assert os.path.exists(os.path.join(ROOT, spec["meta_data"]["script_path"])), path
        v = next(variants(spec))
        assert "env_specs" in v and "seed" in v, path


def test_path_statistics_match_reference_vectors():
    """get_generic_path_information / get_average_returns / create_stats_ordered_dict against the reference's own outputs
    (tests/golden/g15_eval_stats.npz, core/eval_util.py:15-142): same keys in the same order, same values."""
    import os

    import numpy as np
    from ilswiss_amd.samplers import create_stats_ordered_dict, get_average_returns, get_generic_path_information
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g15_eval_stats.npz"))
    for tag, with_success in (("plain", False), ("success", True)):
        paths = []
        for i, T in enumerate(g[tag + "_lens"]):
            infos = [dict(env_id=i, **({"is_success": float((i % 2 == 0) and t == T - 1)} if with_success else {})) for t in range(T)]
            paths.append(dict(rewards=g[f"{tag}_rew{i}"], actions=g[f"{tag}_act{i}"], terminals=np.zeros((T, 1)), env_infos=infos))
        st = get_generic_path_information(paths, stat_prefix="Test")
        assert list(st.keys()) == [str(k) for k in g[tag + "_keys"]]
        np.testing.assert_allclose([float(np.asarray(v).reshape(-1)[0]) for v in st.values()], g[tag + "_vals"], rtol=1e-12, atol=1e-12)
        assert abs(get_average_returns(paths) - float(g[tag + "_avg_return"])) < 1e-12
        np.testing.assert_allclose(get_average_returns(paths, std=True), g[tag + "_avg_return_std"], rtol=1e-12)
    keys, vals = [], []
    for d in (create_stats_ordered_dict("A", 3.5), create_stats_ordered_dict("B", np.array([2.0])),
              create_stats_ordered_dict("C", (np.array([1.0, 2.0, 4.0]), np.array([5.0]))),
              create_stats_ordered_dict("D", np.array([1.0, 3.0]), stat_prefix="P", exclude_max_min=True),
              create_stats_ordered_dict("E", [])):
        for k, v in d.items():
            keys.append(k); vals.append(float(np.asarray(v).reshape(-1)[0]))
    assert keys == [str(k) for k in g["corner_keys"]]
    np.testing.assert_allclose(vals, g["corner_vals"], rtol=1e-12)


def test_variant_grid_matches_reference_vectors():
    """The variant list (content AND order: exp_id is the position) against launcher_util.build_nested_variant_generator run on
    the same specs (tests/golden/g16_variants.npz): nested variables, a single-experiment spec, non-alphabetical key order."""
    import copy
    import json

    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g16_variants.npz"))
    for i in range(4):
        spec, ref = json.loads(str(g[f"spec{i}"])), json.loads(str(g[f"variants{i}"]))
        mine = list(variants(copy.deepcopy(spec)))
        assert [v.pop("exp_id") for v in mine] == list(range(len(ref)))
        assert mine == ref, i


def test_progress_csv_matches_the_reference_logger(tmp_path, capsys):
    """TabularLogger writes the progress.csv rlkit/core/logger.py writes (golden G17): header from the first dump in insertion
    order, str() of every value, CSV quoting of string cells; a column that first appears later extends the header (our
    extension, documented in DESIGN.md §6) without disturbing the earlier cells."""
    import numpy as np
    from ilswiss_amd.algorithm import TabularLogger
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g17_logger_csv.npz"))
    rows = [[("Epoch", 0), ("AverageReturn", np.float64(12.5)), ("QF1 Loss", np.float32(0.25)), ("Alpha", 0.2),
             ("Number of env steps total", 4096), ("Note", "a,b")],
            [("Epoch", 1), ("AverageReturn", np.float64(1234.56789012345)), ("QF1 Loss", np.float32(1e-7)), ("Alpha", 0.19999),
             ("Number of env steps total", 8192), ("Note", "x")]]
    lg = TabularLogger(str(tmp_path))
    for r in rows:
        for k, v in r:
            lg.record_tabular(k, v)
        lg.dump_tabular()
    capsys.readouterr()
    assert open(tmp_path / "progress.csv", newline="").read() == str(g["csv_text"])
    lg.record_tabular("Epoch", 2); lg.record_tabular("Late Column", 7)
    lg.dump_tabular()
    capsys.readouterr()
    lines = open(tmp_path / "progress.csv", newline="").read().split("\r\n")
    assert lines[0] == "Epoch,AverageReturn,QF1 Loss,Alpha,Number of env steps total,Note,Late Column"
    assert lines[1] == '0,12.5,0.25,0.2,4096,"a,b",' and lines[3] == "2,,,,,,7"


def test_log_dir_layout_matches_the_reference(tmp_path, monkeypatch):
    """logs/<exp-name>/<exp_name>_<Y_m_d_H_M_S>_<id:04d>--s-<seed>/variant.json: directory name at a frozen clock and the
    variant.json text against launcher_util.create_log_dir + logger.log_variant (golden G18)."""
    import time

    import numpy as np
    from ilswiss_amd import algorithm
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g18_logdir.npz"))
    real_strftime = time.strftime
    monkeypatch.setattr(algorithm.time, "strftime", lambda fmt, *a: real_strftime(fmt, (2024, 3, 9, 7, 5, 1, 5, 69, 0)))
    variant = dict(seed=17, exp_id=3, exp_name="sac_hopper_hip", net_size=256, sac_params=dict(reward_scale=1.0, alpha=0.2, policy_lr=3e-4),
                   env_specs=dict(env_name="hopper", env_kwargs={}, env_num=4096), flags=[True, None, 1e-7], script_path="run_scripts/x.py")
    d = algorithm.setup_log_dir("sac_hopper_hip", 3, 17, variant, base_dir=str(tmp_path))
    assert os.path.relpath(d, str(tmp_path)) == str(g["rel_dir"])
    assert open(os.path.join(d, "variant.json")).read() == str(g["variant_json"])


def test_normalize_exp_demos_statistics():
    """scripts/normalize_exp_demos.py: the reference's get_normalized rule (normalize_exp_demos.py:24-38) — train statistics applied
    to every split, a constant observation axis keeps std 1, actions untouched."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("norm_demos", os.path.join(ROOT, "scripts", "normalize_exp_demos.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(0)

    def path(L):
        o = rng.normal(3.0, 2.0, (L, 4))
        o[:, 2] = 7.0                       # constant axis
        return dict(observations=o, next_observations=o + 0.1, actions=rng.uniform(-1, 1, (L, 2)), rewards=np.zeros((L, 1)))
    train, test = [path(50), path(30)], [path(20)]
    tr, te, mean, std = m.normalize_paths(train, test)
    allo = np.vstack([p["observations"] for p in train])
    np.testing.assert_allclose(mean[0], allo.mean(0))
    assert std[0, 2] == 1.0 and np.allclose(std[0, [0, 1, 3]], allo.std(0)[[0, 1, 3]])
    z = np.vstack([p["observations"] for p in tr])
    assert np.allclose(z.mean(0), 0, atol=1e-12) and np.allclose(z.std(0)[[0, 1, 3]], 1)
    np.testing.assert_allclose(te[0]["next_observations"], (test[0]["next_observations"] - mean) / std)
    assert tr[0]["actions"] is train[0]["actions"]


def test_loop_driver_defaults_are_base_algorithms():
    """base_algorithm.py:21-54 (+ torch_rl_algorithm.py:8-10: batch_size and num_train_steps_per_train_call have no default), typed in."""
    import inspect
    from ilswiss_amd.algorithm import DeviceRLAlgorithm
    P = inspect.signature(DeviceRLAlgorithm.__init__).parameters
    want = dict(num_epochs=100, num_steps_per_epoch=10000, num_steps_between_train_calls=20, num_steps_per_eval=1000, max_path_length=1000,
                min_steps_before_training=5000, replay_buffer=None, replay_buffer_size=10000, freq_saving=1, save_replay_buffer=False,
                save_best=False, best_key="AverageReturn", no_terminal=False, eval_deterministic=False)
    assert {k: P[k].default for k in want} == want
    assert P["batch_size"].default is inspect.Parameter.empty and P["num_train_steps_per_train_call"].default is inspect.Parameter.empty
    # no **kwargs, as in the reference: a misspelt or unsupported key is an error, not a silently different run
    assert not any(p.kind == p.VAR_KEYWORD for p in P.values())
    for k, dflt in (("eval_policy", None), ("eval_sampler", None), ("save_epoch", False), ("save_best_starting_from_epoch", 0),
                    ("eval_no_terminal", False), ("wrap_absorbing", False), ("render", False), ("freq_log_visuals", 1), ("eval_preprocess_func", None)):
        assert P[k].default == dflt, k


def test_trainers_refuse_another_optimiser_or_criterion():
    """sac_alpha.py:35 / td3.py:36-37: `optimizer_class`, `qf_criterion` are read by the reference; values libilsx does not implement raise."""
    from ilswiss_amd.sac import check_swallowed_kwargs

    class SGD:   # stands for torch.optim.SGD
        pass

    class Adam:
        pass
    check_swallowed_kwargs(dict(optimizer_class=Adam, env=None, foo=1), "X")   # Adam and unknown keys: fine, as in the reference
    with pytest.raises(NotImplementedError, match="optimizer_class"):
        check_swallowed_kwargs(dict(optimizer_class=SGD), "X")
    with pytest.raises(NotImplementedError, match="qf_criterion"):
        check_swallowed_kwargs(dict(qf_criterion=SGD()), "X")



def test_run_experiment_group_hands_each_child_k_variant_files(tmp_path):
    """run_experiment.py --group K / meta_data.seeds_per_process: one child per K consecutive variants (`-e a.yaml b.yaml ...`), children dealt
    round-robin over --gpus; every variant keeps its own file (run_experiment.py:39-66 of the reference writes one per grid point)."""
    import json
    import subprocess
    import sys
    fake = tmp_path / "fake_script.py"
    fake.write_text("import json, sys\nopen(sys.argv[0] + '.calls', 'a').write(json.dumps(sys.argv[1:]) + '\\n')\n")
    spec = dict(meta_data=dict(script_path=str(fake), exp_name="grp", num_workers=1, seeds_per_process=4),
                variables=dict(seed=list(range(10))), constants=dict(env_specs=dict(env_name="hopper")))
    path = tmp_path / "spec.yaml"
    path.write_text(yaml.dump(spec))

    def calls(*extra):
        if os.path.exists(str(fake) + ".calls"):
            os.remove(str(fake) + ".calls")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "run_experiment.py"), "-e", str(path), "--log-root", str(tmp_path / "logs"), *extra],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        cs = [json.loads(line) for line in open(str(fake) + ".calls")]
        return sorted(cs, key=lambda c: yaml.safe_load(open(c[1]))["exp_id"])      # children run concurrently: order by their first variant
    cs = calls("--gpus", "2", "-g", "3")
    files = [[a for a in c[1:c.index("-g")]] for c in cs]
    assert [len(f) for f in files] == [4, 4, 2] and all(c[0] == "-e" for c in cs)
    assert [c[c.index("-g") + 1] for c in cs] == ["3", "4", "3"]                # children round-robin over the two GPUs
    seeds = [yaml.safe_load(open(p))["seed"] for f in files for p in f]
    assert seeds == list(range(10))                                              # consecutive variants, each its own file
    assert [len(c[1:c.index("-g")]) for c in calls("--group", "1")] == [1] * 10  # the flag overrides the spec


def test_group_loop_schedule_on_stub_runs():
    """DeviceRLAlgorithmGroup's lock-step schedule (host logic, no device): every run takes steps_per_epoch / env_num vec steps per epoch, a
    train call happens at the runs' train trigger (base_algorithm.py:293-299) for exactly the runs that are due AND can train — a run whose
    ring is still empty is skipped and its gate does not advance — and every run is evaluated and ends its epoch once per epoch."""
    from ilswiss_amd.algorithm import DeviceRLAlgorithm, DeviceRLAlgorithmGroup

    class Ctx:
        def sync(self):
            pass

    class Trainer:
        def __init__(self):
            self.ctx, self.calls, self.epochs_ended = Ctx(), [], 0

        def train_from_replay(self, rb, n, B):
            self.calls.append((rb.size, n, B))

        def end_epoch(self):
            self.epochs_ended += 1

    class Ring:
        size = 0

    class Env:
        def rollout_stats(self, reset=True):
            return 0, 0.0

    class Run:   # the attributes and pieces of DeviceRLAlgorithm the group drives
        _train_due = DeviceRLAlgorithm._train_due
        _count_train_call = DeviceRLAlgorithm._count_train_call

        def __init__(self, fills_after):
            self.trainer, self.replay_buffer, self.training_env = Trainer(), Ring(), Env()
            self.num_epochs, self.num_env_steps_per_epoch, self.env_num, self.on_policy = 1, 400, 4, False
            self.num_steps_between_train_calls, self.num_train_steps_per_train_call, self.batch_size = 100, 7, 32
            self.max_path_length, self.no_terminal, self.min_steps_before_training = 50, False, 0
            self._n_env_steps_total = self._n_prev_train_env_steps = self._n_train_steps_total = self._n_grad_steps_total = 0
            self.fills_after, self.vec_steps, self.evals = fills_after, 0, []

        def _vec_step(self, begin_only=False):
            self.vec_steps += 1
            self._n_env_steps_total += self.env_num
            if self._n_env_steps_total >= self.fills_after:    # insert_at_episode_end: the ring fills when ITS first episode ends
                self.replay_buffer.size = self._n_env_steps_total

        def _vec_step_end(self):
            pass

        def _can_train(self):
            return self.replay_buffer.size >= 1

        def _eval_collect(self):
            return dict(AverageReturn=1.0)

        def evaluate(self, epoch, epoch_time, total_time, collected=None):
            self.evals.append((epoch, collected))

    runs = [Run(fills_after=40), Run(fills_after=250), Run(fills_after=40)]
    grp = DeviceRLAlgorithmGroup(runs)
    assert grp._lockstep_args is None          # stubs: the per-step path
    grp.train()
    for r in runs:
        assert r.vec_steps == 2 * 100 and r._n_env_steps_total == 800            # epochs 0 and 1 (num_epochs + 1, base_algorithm.py:64)
        assert [e for e, _ in r.evals] == [0, 1] and r.trainer.epochs_ended == 2
    # runs 0 and 2 train at every 100-step trigger (8 calls); run 1 cannot at 100 and 200 (ring empty), its gate stays open, so it trains at
    # the very next vec step after its ring fills (252 env steps) and then every 100 from there: 252, 352, ..., 752 = 6 calls
    assert [len(r.trainer.calls) for r in runs] == [8, 6, 8]
    assert runs[1].trainer.calls[0][0] == 252 and runs[0].trainer.calls[0] == (100, 7, 32)
    assert runs[1]._n_train_steps_total == 6 and runs[1]._n_grad_steps_total == 42
    import pytest as _pt
    bad = Run(40)
    bad.batch_size = 64
    with _pt.raises(ValueError):
        DeviceRLAlgorithmGroup([runs[0], bad])
