#!/usr/bin/env python
"""SAC (twin Q, auto alpha) experiment script on the MI355X engine — same contract as the reference's
run_scripts/sac_alpha_exp_script.py:25-156: `python run_scripts/sac_alpha_exp_script.py -e <variant.yaml> -g <gpu>`,
variant keys env_specs / net_size / num_hidden_layers / sac_params / rl_alg_params / seed / exp_name / exp_id.
Accepts either a flat variant (what run_experiment.py writes per grid point) or a full exp_spec with
meta_data / variables / constants (the first grid point is taken)."""
from _common import flatten_spec, ia, main, make_envs, split_info, start, train  # noqa: F401

from ilswiss_amd.algorithm import DeviceRLAlgorithm


def experiment(variant, gpu=0, log_dir=None):
    ctx = start(variant, gpu)
    training_env, eval_env, env = make_envs(variant, ctx)
    obs_dim, action_dim = training_env.obs_dim, training_env.act_dim
    net_size, num_hidden = variant["net_size"], variant["num_hidden_layers"]
    qf1 = ia.FlattenMlp(hidden_sizes=num_hidden * [net_size], input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    qf2 = ia.FlattenMlp(hidden_sizes=num_hidden * [net_size], input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    policy = ia.ReparamTanhMultivariateGaussianPolicy(hidden_sizes=num_hidden * [net_size], obs_dim=obs_dim,
                                                      action_dim=action_dim, ctx=ctx)
    alg = dict(variant["rl_alg_params"])
    split = split_info()          # rl_alg_params.split_ranks: G — this process is one rank of ONE run split over G GPUs (_common.py)
    if split is not None:
        alg = split.scale(alg)    # B / G rows, env_num / G envs (make_envs), a ring of replay_buffer_size / G rows per rank
    else:
        alg.pop("split_ranks", None)
    trainer = ia.SoftActorCritic(policy=policy, qf1=qf1, qf2=qf2, env=env, max_batch=alg.get("batch_size", 256),
                                 grad_world=split.world if split else 1, **variant["sac_params"])
    if split is not None:
        from ilswiss_amd.parallel import SplitRunStep
        SplitRunStep(trainer)     # gives the ctx its RCCL communicator: train_from_replay all-reduces the gradient arena on the ctx stream
    algorithm = DeviceRLAlgorithm(trainer=trainer, env=env, training_env=training_env, eval_env=eval_env,
                                  exploration_policy=policy, log_dir=log_dir, **alg)
    train(algorithm, variant)
    return algorithm


if __name__ == "__main__":
    main(experiment, "sac")
