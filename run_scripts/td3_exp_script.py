#!/usr/bin/env python
"""TD3 experiment script on the MI355X engine — same contract as the reference's run_scripts/td3_exp_script.py:
`python run_scripts/td3_exp_script.py -e <variant.yaml> -g <gpu>`, variant keys env_specs / net_size /
num_hidden_layers / policy_noise / policy_noise_clip / td3_params / rl_alg_params / seed."""
from _common import ia, main, make_envs, start, train  # noqa: F401

from ilswiss_amd.algorithm import DeviceRLAlgorithm
from ilswiss_amd.td3 import TD3, MlpGaussianNoisePolicy


def experiment(variant, gpu=0, log_dir=None):
    ctx = start(variant, gpu)
    training_env, eval_env, env = make_envs(variant, ctx)
    obs_dim, action_dim = training_env.obs_dim, training_env.act_dim
    hid = variant["num_hidden_layers"] * [variant["net_size"]]
    qf1 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    qf2 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    policy = MlpGaussianNoisePolicy(hidden_sizes=hid, obs_dim=obs_dim, action_dim=action_dim, output_activation="tanh",
                                    policy_noise=variant["policy_noise"], policy_noise_clip=variant["policy_noise_clip"], ctx=ctx)
    alg = dict(variant["rl_alg_params"])
    trainer = TD3(policy=policy, qf1=qf1, qf2=qf2, max_batch=alg.get("batch_size", 256), **variant["td3_params"])
    algorithm = DeviceRLAlgorithm(trainer=trainer, env=env, training_env=training_env, eval_env=eval_env,
                                  exploration_policy=policy, log_dir=log_dir, **alg)
    train(algorithm, variant)
    return algorithm


if __name__ == "__main__":
    main(experiment, "td3")
