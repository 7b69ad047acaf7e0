#!/usr/bin/env python
"""SAC (twin Q, auto alpha) experiment script on the MI355X engine — same contract as the reference's
run_scripts/sac_alpha_exp_script.py:25-156: `python run_scripts/sac_alpha_exp_script.py -e <variant.yaml> -g <gpu>`,
variant keys env_specs / net_size / num_hidden_layers / sac_params / rl_alg_params / seed / exp_name / exp_id.
Accepts either a flat variant (what run_experiment.py:39-45 writes per grid point) or a full exp_spec with
meta_data / variables / constants (the first grid point is taken)."""
import argparse
import os
import sys

import numpy as np
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.algorithm import DeviceRLAlgorithm, setup_log_dir  # noqa: E402
from ilswiss_amd.envs.vecenv import get_envs  # noqa: E402


def flatten_spec(spec):
    if "constants" not in spec:
        return spec
    v = dict(spec["constants"])
    for k, vals in (spec.get("variables") or {}).items():
        v[k] = vals[0] if isinstance(vals, list) else vals
    v.update(spec.get("meta_data") or {})
    v.setdefault("exp_id", 0)
    return v


def experiment(variant, gpu=0, log_dir=None):
    seed = int(variant.get("seed", 0))
    np.random.seed(seed)                                   # set_seed (launcher_util.py:330-344)
    ctx = ia.set_gpu_mode(True, gpu, seed=seed)
    env_specs = dict(variant["env_specs"])
    env_specs["eval_env_seed"] = env_specs["training_env_seed"] = seed  # sac_alpha_exp_script.py:128-132
    training_env = get_envs(env_specs, ctx=ctx)
    # the eval sampler walks its paths on the host (one episode per env): keep it small
    n_eval = int(env_specs.get("eval_env_num", min(int(env_specs.get("env_num", 1)), 16)))
    eval_env = get_envs(dict(env_specs, env_num=n_eval, training_env_seed=seed + 10007), ctx=ctx)
    env = training_env.single_env_view()
    obs_dim, action_dim = training_env.obs_dim, training_env.act_dim
    net_size, num_hidden = variant["net_size"], variant["num_hidden_layers"]
    qf1 = ia.FlattenMlp(hidden_sizes=num_hidden * [net_size], input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    qf2 = ia.FlattenMlp(hidden_sizes=num_hidden * [net_size], input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    policy = ia.ReparamTanhMultivariateGaussianPolicy(hidden_sizes=num_hidden * [net_size], obs_dim=obs_dim,
                                                      action_dim=action_dim, ctx=ctx)
    alg = dict(variant["rl_alg_params"])
    trainer = ia.SoftActorCritic(policy=policy, qf1=qf1, qf2=qf2, env=env, max_batch=alg.get("batch_size", 256),
                                 **variant["sac_params"])
    algorithm = DeviceRLAlgorithm(trainer=trainer, env=env, training_env=training_env, eval_env=eval_env,
                                  exploration_policy=policy, log_dir=log_dir, **alg)
    algorithm.train()
    return algorithm


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-e", "--experiment", required=True, help="experiment specification file")
    ap.add_argument("-g", "--gpu", type=int, default=0, help="gpu id")
    args = ap.parse_args()
    with open(args.experiment) as f:
        variant = flatten_spec(yaml.safe_load(f))
    log_dir = setup_log_dir(variant.get("exp_name", "sac"), int(variant.get("exp_id", 0)), int(variant.get("seed", 0)), variant)
    experiment(variant, args.gpu, log_dir)
