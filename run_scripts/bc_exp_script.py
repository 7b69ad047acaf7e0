#!/usr/bin/env python
"""Behaviour-cloning experiment script on the MI355X engine — same contract as the reference's run_scripts/bc_exp_script.py:
expert demonstrations through `demos_listing.yaml` (expert_name / expert_idx / traj_num) into an expert replay buffer, a
tanh-Gaussian policy of policy_net_size x policy_num_hidden_layers, `bc_params` = the kwargs of BC (bc/bc.py:14-41) plus the
loop keys.  The loop is BC.start_training (bc.py:56-75): no environment sampling, `num_updates_per_train_call` updates every
`num_steps_between_train_calls` counted steps, evaluation every epoch.  scale_env_with_demo_stats /
minmax_env_with_demo_stats wrap the envs in ScaledEnv / MinmaxEnv with the demonstrations' statistics (:56-97)."""
import time
from collections import OrderedDict

import numpy as np
from _common import ia, main, make_envs, start
from adv_irl_exp_script import demo_stat_wrapper, load_demos

from ilswiss_amd.algorithm import TabularLogger
from ilswiss_amd.bc import BC
from ilswiss_amd.replay import EnvReplayBuffer
from ilswiss_amd.samplers import DeviceEvalSampler


def experiment(variant, gpu=0, log_dir=None):
    ctx = start(variant, gpu)
    import random
    random.seed(int(variant.get("seed", 0)))
    traj_list = load_demos(variant)
    wrapper, wrapper_kwargs = demo_stat_wrapper(variant, traj_list)
    training_env, eval_env, env = make_envs(variant, ctx, env_wrapper=wrapper, wrapper_kwargs=wrapper_kwargs)
    p = dict(variant["bc_params"])
    expert_rb = EnvReplayBuffer(p["replay_buffer_size"], env, random_seed=int(np.random.randint(10000)), ctx=ctx)
    for tj in traj_list:
        expert_rb.add_path(tj, absorbing=False, env=env)
    hid = variant["policy_num_hidden_layers"] * [variant["policy_net_size"]]
    policy = ia.ReparamTanhMultivariateGaussianPolicy(hidden_sizes=hid, obs_dim=training_env.obs_dim, action_dim=training_env.act_dim, ctx=ctx)
    trainer = BC(p["mode"], policy, expert_replay_buffer=expert_rb, num_updates_per_train_call=p.get("num_updates_per_train_call", 1),
                 batch_size=p.get("batch_size", 1024), lr=p.get("lr", 1e-3), momentum=p.get("momentum", 0.0),
                 wrap_absorbing=p.get("wrap_absorbing", False))
    eval_policy = ia.MakeDeterministic(policy) if p.get("eval_deterministic", True) else policy
    sampler = DeviceEvalSampler(eval_env, eval_policy, p.get("num_steps_per_eval", 1000), p.get("max_path_length", 1000))
    lg, n_steps, n_updates, best, t_start = TabularLogger(log_dir), 0, 0, -np.inf, time.perf_counter()
    for epoch in range(p["num_epochs"]):                                  # bc.py:59-75
        t0 = time.perf_counter()
        for _ in range(p["num_steps_per_epoch"]):
            n_steps += 1
            if n_steps % p["num_steps_between_train_calls"] == 0:
                trainer.train_from_replay()
                n_updates += trainer.num_updates_per_train_call
        ctx.sync()
        st = OrderedDict(trainer.get_eval_statistics() or {})
        ev = sampler.obtain_statistics("Test")
        st["AverageReturn"] = ev.pop("AverageReturn")
        st.update(ev)
        for k, v in st.items():
            lg.record_tabular(k, float(v))
        lg.record_tabular("Number of train steps total", n_updates)
        lg.record_tabular("Epoch Time (s)", time.perf_counter() - t0)
        lg.record_tabular("Total Train Time (s)", time.perf_counter() - t_start)
        lg.record_tabular("Epoch", epoch)
        lg.dump_tabular()
        if p.get("freq_saving") and epoch % p["freq_saving"] == 0:
            lg.save("params.pkl", dict(epoch=epoch, **trainer.get_snapshot()))
        if p.get("save_best", True) and st["AverageReturn"] > best:
            best = st["AverageReturn"]
            lg.save("best.pkl", dict(epoch=epoch, **trainer.get_snapshot()))
        trainer.end_epoch()
    return trainer


if __name__ == "__main__":
    main(experiment, "bc")
