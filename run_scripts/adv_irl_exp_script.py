#!/usr/bin/env python
"""Adversarial IRL (GAIL / AIRL / FAIRL / gail2) experiment script on the MI355X engine — same contract as the reference's
run_scripts/adv_irl_exp_script.py:30-201: expert demonstrations are looked up in `demos_listing.yaml`
(expert_name / expert_idx), a pickled `list[dict]` of trajectories with keys observations / actions / rewards /
next_observations / terminals (what run_scripts/gen_expert_demos.py writes); `traj_num` of them are drawn with
random.sample and loaded with `add_path` into the expert replay buffer; variant keys disc_* / policy_net_size /
policy_num_hidden_layers / adv_irl_params / sac_params / env_specs."""
import os
import pickle
import random

import numpy as np
import yaml
from _common import ia, main, make_envs, split_info, start, train  # noqa: F401

from ilswiss_amd.adv_irl import AdvIRLTrainer, MLPDisc
from ilswiss_amd.algorithm import DeviceRLAlgorithm
from ilswiss_amd.replay import EnvReplayBuffer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_demos(variant):
    listing = variant.get("demos_listing", os.path.join(ROOT, "demos_listing.yaml"))
    with open(listing) as f:
        listings = yaml.safe_load(f)
    path = listings[variant["expert_name"]]["file_paths"][variant["expert_idx"]]
    if not os.path.isabs(path):
        path = os.path.join(os.path.dirname(os.path.abspath(listing)), path)
    with open(path, "rb") as f:
        traj_list = pickle.load(f)
    return random.sample(traj_list, variant["traj_num"])          # adv_irl_exp_script.py:51-53


def demo_stat_wrapper(variant, traj_list):
    """adv_irl_exp_script.py:55-115: observation statistics of the sampled demonstrations -> ScaledEnv / MinmaxEnv for the
    envs, and the same map applied to the demonstrations' observations / next_observations in place."""
    from ilswiss_amd.envs.vecenv import EPS, MinmaxEnv, ProxyEnv, ScaledEnv
    obs = np.vstack([tj["observations"] for tj in traj_list])
    if variant.get("scale_env_with_demo_stats"):
        mean, std = np.mean(obs, axis=0), np.std(obs, axis=0)
        for tj in traj_list:
            for k in ("observations", "next_observations"):
                tj[k] = (tj[k] - mean) / (std + EPS)
        return ScaledEnv, dict(obs_mean=mean, obs_std=std, acts_mean=None, acts_std=None)
    if variant.get("minmax_env_with_demo_stats"):
        lo, hi = np.min(obs, axis=0), np.max(obs, axis=0)
        for tj in traj_list:
            for k in ("observations", "next_observations"):
                tj[k] = (tj[k] - lo) / (hi - lo + EPS)
        return MinmaxEnv, dict(obs_min=lo, obs_max=hi)
    return ProxyEnv, {}


def experiment(variant, gpu=0, log_dir=None):
    ctx = start(variant, gpu)
    random.seed(int(variant.get("seed", 0)))
    traj_list = load_demos(variant)
    wrapper, wrapper_kwargs = demo_stat_wrapper(variant, traj_list)
    training_env, eval_env, env = make_envs(variant, ctx, env_wrapper=wrapper, wrapper_kwargs=wrapper_kwargs)
    obs_dim, action_dim = training_env.obs_dim, training_env.act_dim
    p = dict(variant["adv_irl_params"])
    split = split_info()          # adv_irl_params.split_ranks: G — this process is one rank of ONE run split over G GPUs (_common.py): env_num / G envs
    G = split.world if split else 1   # (make_envs), a policy ring of replay_buffer_size / G rows, every batch and step count of the schedule / G
    if split is not None:
        p = split.scale_rows(split.scale(p), ("disc_optim_batch_size", "policy_optim_batch_size", "policy_optim_batch_size_from_expert"))
        p.pop("batch_size", None)
        from ilswiss_amd.parallel import ensure_comm
        ensure_comm(ctx)
    p.pop("split_ranks", None)
    if p.get("wrap_absorbing"):
        raise NotImplementedError("wrap_absorbing is off in the hot-path config (gail_walker.yaml:49)")
    # every rank of a split run holds ALL demonstrations and draws its share of each expert batch: its draws are keyed apart from the other ranks'
    expert_rb = EnvReplayBuffer(variant["adv_irl_params"]["replay_buffer_size"], env,
                                random_seed=int(np.random.randint(10000)) + (7919 * split.rank if split else 0), ctx=ctx)
    for tj in traj_list:                                           # adv_irl_exp_script.py:135-138
        expert_rb.add_path(tj, absorbing=False, env=env)
    hid = variant["policy_num_hidden_layers"] * [variant["policy_net_size"]]
    qf1 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    qf2 = ia.FlattenMlp(hidden_sizes=hid, input_size=obs_dim + action_dim, output_size=1, ctx=ctx)
    policy = ia.ReparamTanhMultivariateGaussianPolicy(hidden_sizes=hid, obs_dim=obs_dim, action_dim=action_dim, ctx=ctx)
    Bp = p.get("policy_optim_batch_size", 1024)
    input_dim = obs_dim + (obs_dim if p.get("state_only") else action_dim)   # adv_irl_exp_script.py:164-166
    disc = MLPDisc(input_dim, num_layer_blocks=variant["disc_num_blocks"], hid_dim=variant["disc_hid_dim"],
                   hid_act=variant["disc_hid_act"], use_bn=variant["disc_use_bn"], clamp_magnitude=variant["disc_clamp_magnitude"], ctx=ctx)
    sac = ia.SoftActorCritic(policy=policy, qf1=qf1, qf2=qf2, env=env, max_batch=Bp, grad_world=G, **variant["sac_params"])
    irl_keys = ("state_only", "disc_optim_batch_size", "policy_optim_batch_size", "policy_optim_batch_size_from_expert",
                "num_update_loops_per_train_call", "num_disc_updates_per_loop_iter", "num_policy_updates_per_loop_iter", "disc_lr",
                "disc_momentum", "use_grad_pen", "grad_pen_weight", "rew_clip_min", "rew_clip_max")
    trainer = AdvIRLTrainer(p["mode"], disc, sac, expert_rb, grad_world=G, **{k: p[k] for k in irl_keys if k in p})   # adv_irl.py:34-54 defaults otherwise
    loop_keys = ("num_epochs", "num_steps_per_epoch", "num_steps_between_train_calls", "max_path_length", "min_steps_before_training",
                 "eval_deterministic", "num_steps_per_eval", "replay_buffer_size", "no_terminal", "save_best", "freq_saving",
                 "save_epoch", "save_best_starting_from_epoch", "save_replay_buffer", "best_key", "eval_no_terminal", "wrap_absorbing",
                 "render", "freq_log_visuals")   # every BaseAlgorithm key of adv_irl_params (base_algorithm.py:21-54): honoured or refused, never dropped
    alg = {k: p[k] for k in loop_keys if k in p}
    if split is not None:
        alg.update(split_world=p["split_world"], split_agree=p["split_agree"])
    algorithm = DeviceRLAlgorithm(trainer=trainer, env=env, training_env=training_env, eval_env=eval_env, exploration_policy=policy,
                                  log_dir=log_dir, num_train_steps_per_train_call=p.get("num_update_loops_per_train_call", 1),
                                  batch_size=Bp, **alg)
    train(algorithm, variant)
    return algorithm


if __name__ == "__main__":
    main(experiment, "adv_irl")
