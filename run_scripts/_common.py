"""Shared plumbing of the run scripts (the reference repeats it in every run_scripts/*_exp_script.py): variant loading
(what run_experiment.py writes per grid point, or a full exp_spec whose first grid point is taken), seeding
(launcher_util.py:330-344), env construction (rlkit/envs/__init__.py:72-132) and the log directory
(launcher_util.py:209-297)."""
import argparse
import os
import sys

import numpy as np
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.algorithm import setup_log_dir  # noqa: E402
from ilswiss_amd.envs.vecenv import get_envs  # noqa: E402
from ilswiss_amd.launcher import variants  # noqa: E402


def flatten_spec(spec):
    """A flat variant passes through; a full exp_spec yields its first grid point."""
    if "constants" not in spec:
        return spec
    return next(variants(spec))


def make_envs(variant, ctx, **vec_kwargs):
    """training env (env_num / training_env_num envs) + a small eval env (its paths are walked on the host)."""
    seed = int(variant.get("seed", 0))
    env_specs = dict(variant["env_specs"])
    n_train = int(env_specs.get("env_num", env_specs.get("training_env_num", 1)))
    n_eval = int(env_specs.get("eval_env_num", min(n_train, 16)))
    training_env = get_envs(dict(env_specs, env_num=n_train, training_env_seed=seed), ctx=ctx, **vec_kwargs)
    eval_kwargs = dict(vec_kwargs)
    if vec_kwargs.get("norm_obs"):   # ppo_exp_script.py:68-75: the eval env shares the statistics and does not update them
        eval_kwargs.update(obs_rms=training_env.obs_rms, update_obs_rms=False)
    eval_env = get_envs(dict(env_specs, env_num=n_eval, training_env_seed=seed + 10007), ctx=ctx, **eval_kwargs)
    return training_env, eval_env, training_env.single_env_view()


# ---- several variants in ONE process (run_experiment.py --group K / meta_data.seeds_per_process): `-e a.yaml b.yaml ...`.  Every variant
# goes through the run script's own experiment() unchanged; start() hands run k a sibling context of run 0's (same device and stream, its own
# Philox key and object counter — what it would have in a process of its own) and train() collects the built algorithms instead of training
# them; main() then advances them in lock-step (ilswiss_amd.algorithm.DeviceRLAlgorithmGroup).
_GROUP = None


def start(variant, gpu):
    seed = int(variant.get("seed", 0))
    np.random.seed(seed)
    if _GROUP is not None and _GROUP["ctx"] is not None:
        # ILSX_GROUP_SHARE_STREAM=1 (A/B): every run of the group on run 0's stream — the runs' small rollout launches then queue up behind
        # each other (measured on ten 4-env Hopper runs: 4.9 s of sampling per epoch against 1.7 s on a stream per run)
        ctx = _GROUP["ctx"].sibling(seed, share_stream=bool(os.environ.get("ILSX_GROUP_SHARE_STREAM")))
        ia.device.set_default_context(ctx)
        return ctx
    ctx = ia.set_gpu_mode(True, gpu, seed=seed)
    if _GROUP is not None:
        _GROUP["ctx"] = ctx
    return ctx


def train(algorithm, variant):
    """`load_params` (sac_alpha_exp_script.py:106-113): resume from <load_path>/{params,extra_data}.pkl, then train."""
    from ilswiss_amd.snapshot import load_from_file
    epoch = 0
    if variant.get("load_params"):
        algorithm, epoch = load_from_file(algorithm, **variant["load_params"])
    print("Start from epoch", epoch)
    if _GROUP is not None:
        _GROUP["runs"].append((algorithm, epoch))
        return algorithm
    algorithm.train(start_epoch=epoch)
    return algorithm


def _log_dir(variant, default_name):
    load_path = (variant.get("load_params") or {}).get("load_path")
    if load_path:   # a resumed run keeps logging into the directory it resumes from (sac_alpha_exp_script.py:142-146)
        return load_path
    return setup_log_dir(variant.get("exp_name", default_name), int(variant.get("exp_id", 0)), int(variant.get("seed", 0)), variant)


def main(experiment, default_name):
    global _GROUP
    ap = argparse.ArgumentParser()
    ap.add_argument("-e", "--experiment", required=True, nargs="+",
                    help="experiment specification file; several files = that many runs in this one process, stepped in lock-step")
    ap.add_argument("-g", "--gpu", type=int, default=0, help="gpu id")
    args = ap.parse_args()
    variants_ = []
    for path in args.experiment:
        with open(path) as f:
            variants_.append(flatten_spec(yaml.safe_load(f)))
    if len(variants_) == 1:
        return experiment(variants_[0], args.gpu, _log_dir(variants_[0], default_name))
    from ilswiss_amd.algorithm import DeviceRLAlgorithmGroup
    _GROUP = dict(ctx=None, runs=[])
    for v in variants_:
        experiment(v, args.gpu, _log_dir(v, default_name))
    runs, _GROUP = _GROUP["runs"], None
    epochs = {e for _, e in runs}
    if len(epochs) != 1:
        raise SystemExit(f"grouped runs must resume from the same epoch (found {sorted(epochs)})")
    group = DeviceRLAlgorithmGroup([a for a, _ in runs])
    group.train(start_epoch=epochs.pop())
    return group
