#!/usr/bin/env python
"""HER experiment script (TD3 or SAC trainer) — contract of the reference's run_scripts/her_td3_exp_script.py / her_sac_exp_script.py:
`-e <variant.yaml> -g <gpu>`, variant keys env_specs / net_size / num_hidden_layers / td3_params | sac_params / rl_alg_params (incl.
relabel_type, her_ratio) / seed.  The reference's goal envs are gym's Fetch robots (MuJoCo); here `env_name: point-reach` selects the
stand-in of ilswiss_amd/her.py.  Writes progress.csv with the per-epoch success rate."""
from _common import ia, main, start  # noqa: F401

from ilswiss_amd import her
from ilswiss_amd.algorithm import TabularLogger


def experiment(variant, gpu=0, log_dir=None):
    ctx = start(variant, gpu)
    name = variant["env_specs"]["env_name"]
    if name != "point-reach":
        raise NotImplementedError(f"goal env {name!r}: the reference's Fetch envs need MuJoCo; only the stand-in 'point-reach' exists here")
    env = her.PointReachEnv(seed=int(variant.get("seed", 0)), **variant["env_specs"].get("env_kwargs", {}))
    sp = env.observation_space.spaces
    obs_dim, goal_dim, action_dim = sp["observation"].shape[0], sp["desired_goal"].shape[0], env.action_space.shape[0]
    hid = variant["num_hidden_layers"] * [variant["net_size"]]
    qf1 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + goal_dim + action_dim, output_size=1, ctx=ctx)
    qf2 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + goal_dim + action_dim, output_size=1, ctx=ctx)
    alg = dict(variant["rl_alg_params"])
    if "sac_params" in variant:        # her_sac_exp_script.py
        policy = ia.ReparamTanhMultivariateGaussianPolicy(hidden_sizes=hid, obs_dim=obs_dim + goal_dim, action_dim=action_dim, ctx=ctx)
        explore = her.ConditionedPolicy(policy)
        trainer = her.SAC(policy, qf1, qf2, max_batch=alg.get("batch_size", 128), **variant["sac_params"])
    else:
        policy = explore = her.MlpGaussianAndEpsilonPolicy(hidden_sizes=hid, obs_dim=obs_dim, action_dim=action_dim, condition_dim=goal_dim,
                                                           action_space=env.action_space, output_activation="tanh", ctx=ctx)
        trainer = her.TD3(policy, qf1, qf2, max_batch=alg.get("batch_size", 128), **variant["td3_params"])
    algorithm = her.HER(trainer, env, explore, **alg)
    logger = TabularLogger(log_dir)
    for epoch, success in enumerate(algorithm.train()):
        logger.record_tabular("Epoch", epoch)
        logger.record_tabular("Success Rate", success)
        logger.record_tabular("Number of env steps total", (epoch + 1) * algorithm.num_steps_per_epoch)
        logger.dump_tabular()
    return algorithm


if __name__ == "__main__":
    main(experiment, "her")
