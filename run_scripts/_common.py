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


def start(variant, gpu):
    seed = int(variant.get("seed", 0))
    np.random.seed(seed)
    return ia.set_gpu_mode(True, gpu, seed=seed)


def train(algorithm, variant):
    """`load_params` (sac_alpha_exp_script.py:106-113): resume from <load_path>/{params,extra_data}.pkl, then train."""
    from ilswiss_amd.snapshot import load_from_file
    epoch = 0
    if variant.get("load_params"):
        algorithm, epoch = load_from_file(algorithm, **variant["load_params"])
    print("Start from epoch", epoch)
    algorithm.train(start_epoch=epoch)
    return algorithm


def main(experiment, default_name):
    ap = argparse.ArgumentParser()
    ap.add_argument("-e", "--experiment", required=True, help="experiment specification file")
    ap.add_argument("-g", "--gpu", type=int, default=0, help="gpu id")
    args = ap.parse_args()
    with open(args.experiment) as f:
        variant = flatten_spec(yaml.safe_load(f))
    load_path = (variant.get("load_params") or {}).get("load_path")
    if load_path:   # a resumed run keeps logging into the directory it resumes from (sac_alpha_exp_script.py:142-146)
        log_dir = load_path
    else:
        log_dir = setup_log_dir(variant.get("exp_name", default_name), int(variant.get("exp_id", 0)), int(variant.get("seed", 0)), variant)
    return experiment(variant, args.gpu, log_dir)
