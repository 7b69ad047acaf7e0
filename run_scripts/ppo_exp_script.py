#!/usr/bin/env python
"""PPO experiment script on the MI355X engine — same contract as the reference's run_scripts/ppo_exp_script.py:60-117:
variant keys env_specs / net_size / num_hidden_layers / ppo_params / rl_alg_params / seed.  Observations are
normalised by the training env's running statistics, which the eval env shares (:60-75); policy = Gaussian with a
state-independent log-std, value net and policy use tanh units (:82-96)."""
from _common import ia, main, make_envs, start, train  # noqa: F401

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
    horizon = max(1, alg["num_steps_between_train_calls"] // len(training_env))
    trainer = PPO(policy=policy, vf=vf, max_samples=horizon * len(training_env), **variant["ppo_params"])
    algorithm = DeviceRLAlgorithm(trainer=trainer, env=env, training_env=training_env, eval_env=eval_env,
                                  exploration_policy=policy, log_dir=log_dir, **alg)
    train(algorithm, variant)
    return algorithm


if __name__ == "__main__":
    main(experiment, "ppo")
