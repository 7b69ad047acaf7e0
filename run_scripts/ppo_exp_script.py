#!/usr/bin/env python
"""PPO experiment script on the MI355X engine — same contract as the reference's run_scripts/ppo_exp_script.py:60-117:
variant keys env_specs / net_size / num_hidden_layers / ppo_params / rl_alg_params / seed.  Observations are
normalised by the training env's running statistics, which the eval env shares (:60-75); policy = Gaussian with a
state-independent log-std, value net and policy use tanh units (:82-96)."""
from _common import ia, main, make_envs, split_info, start, train  # noqa: F401

from ilswiss_amd.algorithm import DeviceRLAlgorithm
from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy


def experiment(variant, gpu=0, log_dir=None):
    ctx = start(variant, gpu)
    training_env, eval_env, env = make_envs(variant, ctx, norm_obs=True)
    obs_dim, action_dim = training_env.obs_dim, training_env.act_dim
    hid = variant["num_hidden_layers"] * [variant["net_size"]]
    vf = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim, output_size=1, hidden_activation="tanh", ctx=ctx)
    policy = ReparamMultivariateGaussianPolicy(hidden_sizes=hid, obs_dim=obs_dim, action_dim=action_dim, conditioned_std=False,
                                               hidden_activation="tanh", ctx=ctx)
    alg = dict(variant["rl_alg_params"])
    ppo_params = dict(variant["ppo_params"])
    split = split_info()          # rl_alg_params.split_ranks: G — this process is one rank of ONE run split over G GPUs (_common.py): env_num / G envs
    if split is not None:         # (make_envs), the step counts of the schedule and every minibatch divided by G, both gradient arenas all-reduced
        alg = split.scale(alg)
        ppo_params = split.scale_rows(ppo_params, ("mini_batch_size",))
        from ilswiss_amd.parallel import ensure_comm
        ensure_comm(ctx)
    else:
        alg.pop("split_ranks", None)
    horizon = max(1, alg["num_steps_between_train_calls"] // len(training_env))
    trainer = PPO(policy=policy, vf=vf, max_samples=horizon * len(training_env), grad_world=split.world if split else 1, **ppo_params)
    algorithm = DeviceRLAlgorithm(trainer=trainer, env=env, training_env=training_env, eval_env=eval_env,
                                  exploration_policy=policy, log_dir=log_dir, **alg)
    train(algorithm, variant)
    return algorithm


if __name__ == "__main__":
    main(experiment, "ppo")
