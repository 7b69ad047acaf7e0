#!/usr/bin/env python
"""SAC (state-value variant, fixed temperature) experiment script on the MI355X engine — same contract as the
reference's run_scripts/sac_exp_script.py: variant keys env_specs / net_size / num_hidden_layers / sac_params /
rl_alg_params / seed."""
from _common import ia, main, make_envs, start, train  # noqa: F401

from ilswiss_amd.algorithm import DeviceRLAlgorithm
from ilswiss_amd.sac_v import SoftActorCriticV


def experiment(variant, gpu=0, log_dir=None):
    ctx = start(variant, gpu)
    training_env, eval_env, env = make_envs(variant, ctx)
    obs_dim, action_dim = training_env.obs_dim, training_env.act_dim
    hid = variant["num_hidden_layers"] * [variant["net_size"]]
    qf1 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    qf2 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    vf = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim, output_size=1, ctx=ctx)
    policy = ia.ReparamTanhMultivariateGaussianPolicy(hidden_sizes=hid, obs_dim=obs_dim, action_dim=action_dim, ctx=ctx)
    alg = dict(variant["rl_alg_params"])
    trainer = SoftActorCriticV(policy=policy, qf1=qf1, qf2=qf2, vf=vf, max_batch=alg.get("batch_size", 256), **variant["sac_params"])
    algorithm = DeviceRLAlgorithm(trainer=trainer, env=env, training_env=training_env, eval_env=eval_env,
                                  exploration_policy=policy, log_dir=log_dir, **alg)
    train(algorithm, variant)
    return algorithm


if __name__ == "__main__":
    main(experiment, "sac")
