"""The outer RL loop with the reference's knobs and log schema:
`BaseAlgorithm.start_training / _try_to_train / _try_to_eval / evaluate` (rlkit/core/base_algorithm.py:
166-348,599-656) + `TorchRLAlgorithm._do_training` (rlkit/torch/algorithms/torch_rl_algorithm.py:16-34),
driven entirely on the device: one `rollout_step` per vec step (policy -> physics -> replay record ->
auto-reset) and `trainer.train_from_replay` per train call.  `rl_alg_params` YAML keys are the kwargs.

Log layout follows rlkit/launchers/launcher_util.py:209-297 + rlkit/core/logger.py:226-367:
`<log_dir>/{variant.json, progress.csv, params.pkl, best.pkl, extra_data.pkl}`; column names of progress.csv
are the reference's ("Test Returns Mean", "AverageReturn", "QF1 Loss", "Train Time (s)", ...).
"""
import csv
import json
import os
import pickle
import sys
import time
from collections import OrderedDict

import numpy as np

from .networks import MakeDeterministic
from .replay import EnvReplayBuffer
from .samplers import DeviceEvalSampler, VecPathSampler, get_average_returns, get_generic_path_information


class TabularLogger:
    """Minimal rllab-style tabular logger (rlkit/core/logger.py:226-227,300-336): stdout + progress.csv."""

    def __init__(self, log_dir=None):
        self.log_dir, self.row, self._header = log_dir, OrderedDict(), None
        if log_dir:
            os.makedirs(log_dir, exist_ok=True)
            path = os.path.join(log_dir, "progress.csv")
            if os.path.exists(path):   # resumed run (load_params.load_path == log dir): keep the earlier epochs' rows
                with open(path, newline="") as f:
                    rd = csv.DictReader(f)
                    self._header, self._rows = list(rd.fieldnames or []), [dict(r) for r in rd]

    def record_tabular(self, k, v):
        self.row[k] = v

    def log(self, line, with_timestamp=True):
        """logger.log (rlkit/core/logger.py:209-223): stdout + the text output `debug.log` (launcher_util.py:215,267-269)."""
        if with_timestamp:
            line = f"{time.strftime('%Y-%m-%d %H:%M:%S')} | {line}"
        print(line)
        if self.log_dir:
            with open(os.path.join(self.log_dir, "debug.log"), "a") as f:
                f.write(line + "\n")

    def dump_tabular(self):
        width = max(len(k) for k in self.row)
        self.log("-" * (width + 18), with_timestamp=False)   # the reference prints the table through logger.log too (:306-307)
        for k, v in self.row.items():
            self.log(f"{k:<{width}}  {v:>14.6g}" if isinstance(v, (int, float, np.floating, np.integer)) else f"{k:<{width}}  {v}",
                     with_timestamp=False)
        self.log("-" * (width + 18), with_timestamp=False)
        sys.stdout.flush()
        if self.log_dir:
            path = os.path.join(self.log_dir, "progress.csv")
            if self._header is None:
                self._header, self._rows = [], []
            fresh = [k for k in self.row if k not in self._header]
            self._rows.append(dict(self.row))
            if fresh:   # a statistic that first appears in a later epoch gets its column (earlier rows stay empty there)
                self._header += fresh
                with open(path, "w", newline="") as f:
                    w = csv.DictWriter(f, fieldnames=self._header, extrasaction="ignore")
                    w.writeheader()
                    w.writerows(self._rows)
            else:
                with open(path, "a", newline="") as f:
                    csv.DictWriter(f, fieldnames=self._header, extrasaction="ignore").writerow(self.row)
        self.row = OrderedDict()

    def save(self, name, obj):
        if self.log_dir:
            with open(os.path.join(self.log_dir, name), "wb") as f:
                pickle.dump(obj, f)


class DeviceRLAlgorithm:
    """TorchRLAlgorithm(trainer, env, training_env, eval_env, exploration_policy, **rl_alg_params)."""

    def __init__(self, trainer, env, training_env, eval_env, exploration_policy, *, batch_size, num_train_steps_per_train_call,
                 num_epochs=100, num_steps_per_epoch=10000, num_steps_between_train_calls=20, num_steps_per_eval=1000,
                 max_path_length=1000, min_steps_before_training=5000, replay_buffer_size=10000, no_terminal=False,
                 eval_deterministic=False, freq_saving=1, save_best=False, save_replay_buffer=False, replay_buffer=None, log_dir=None,
                 best_key="AverageReturn", bootstrap_open_segments=True, eval_on_device=True, insert_at_episode_end=False,
                 eval_policy=None, eval_sampler=None, save_epoch=False, save_best_starting_from_epoch=0, eval_no_terminal=False,
                 wrap_absorbing=False, render=False, render_kwargs=None, freq_log_visuals=1, eval_preprocess_func=None,
                 split_world=1, split_agree=None, eval_async=False):
        # keyword names and DEFAULTS are BaseAlgorithm's (base_algorithm.py:21-54); batch_size and num_train_steps_per_train_call have
        # none there either (torch_rl_algorithm.py:8-10).  Every shipped spec states all of them; log_dir / bootstrap_open_segments /
        # eval_on_device / insert_at_episode_end are ilswiss_amd keys.
        # insert_at_episode_end (an ilswiss_amd key, default off): the fused rollout keeps the reference's replay order — samples enter
        # the ring when their episode ends, contiguous and registered in _traj_endpoints (base_algorithm.py:509-519) — instead of
        # inserting every transition as it happens
        # BaseAlgorithm takes no **kwargs (an unknown key is a TypeError there, and here); of its own keys the ones below select machinery
        # this loop does not have: asking for them fails loudly instead of training something else
        for name, val in (("eval_no_terminal", eval_no_terminal), ("wrap_absorbing", wrap_absorbing), ("render", render),
                          ("eval_preprocess_func", eval_preprocess_func)):
            if val:
                raise NotImplementedError(f"DeviceRLAlgorithm({name}={val!r}) is not implemented (base_algorithm.py:46-53)")
        # split_world / split_agree (ilswiss_amd keys, set by the run scripts for rl_alg_params.split_ranks): this loop is one rank of a run
        # split over split_world GPUs — its counts are this rank's share, the logged env-step total is the run's, and a train call happens
        # only when EVERY rank can train (split_agree: an all-reduce of the local answer)
        self.split_world, self.split_agree = int(split_world), split_agree
        self.save_epoch, self.save_best_starting_from_epoch = bool(save_epoch), int(save_best_starting_from_epoch)
        if insert_at_episode_end and hasattr(training_env, "set_path_mode"):
            training_env.set_path_mode(True)
        self.no_terminal = bool(no_terminal)   # base_algorithm.py:195-196,208-210: stored terminal flags forced to False
        self.on_policy = bool(getattr(trainer, "on_policy", False))   # torch_rl_algorithm.py:30-32
        self.bootstrap_open_segments = bootstrap_open_segments
        self.trainer, self.env, self.training_env, self.eval_env = trainer, env, training_env, eval_env
        self.exploration_policy = exploration_policy
        self.num_epochs, self.num_env_steps_per_epoch = num_epochs, num_steps_per_epoch
        self.num_steps_between_train_calls = num_steps_between_train_calls
        self.num_train_steps_per_train_call = num_train_steps_per_train_call
        self.num_steps_per_eval, self.max_path_length = num_steps_per_eval, max_path_length
        self.min_steps_before_training, self.batch_size = min_steps_before_training, batch_size
        self.freq_saving, self.save_best, self.best_key = freq_saving, save_best, best_key
        self.save_replay_buffer = bool(save_replay_buffer)
        self.env_num = len(training_env)
        if replay_buffer is None and not self.on_policy:
            seed = int(np.random.randint(10000))  # base_algorithm.py:118-120
            replay_buffer = EnvReplayBuffer(replay_buffer_size, env, random_seed=seed, ctx=trainer.ctx)
        self.replay_buffer = replay_buffer
        if eval_policy is None:   # base_algorithm.py:87-92
            eval_policy = MakeDeterministic(exploration_policy) if eval_deterministic else exploration_policy
        # evaluation runs on the device unless asked otherwise (the host-walked VecPathSampler is the reference's own loop)
        sampler_cls = DeviceEvalSampler if (eval_on_device and hasattr(eval_env, "h")) else VecPathSampler
        self.eval_sampler = eval_sampler if eval_sampler is not None else sampler_cls(eval_env, eval_policy, num_steps_per_eval, max_path_length)
        self.logger = TabularLogger(log_dir)
        # eval_async (an ilswiss_amd key, default off): the evaluation of epoch e runs on a FROZEN copy of the policy, on the eval env's own
        # context (stream), while epoch e + 1 samples and trains; its row is written when it is in (before the next evaluation starts).
        # Same statistics, same snapshots, the same evaluation-env sequence as the blocking form; what differs is (1) the time columns and
        # (2) the exploration-noise stream from epoch 1 on: a blocking evaluation advances the policy context's act-call counter
        # (ilsx_policy_act keys its Philox draw with it), the frozen copy advances its own context's — same distribution, other draws.
        self.eval_async = self._eval_async_setup() if eval_async else None
        self._eval_pending = None
        self._n_env_steps_total = self._n_train_steps_total = self._n_prev_train_env_steps = self._n_grad_steps_total = 0
        self._n_rollouts_total, self.best_statistic_so_far = 0, -np.inf
        self._t_sample = self._t_train = self._t_eval = 0.0

    # base_algorithm.py:364-367
    def _can_train(self):
        ok = self.replay_buffer.num_steps_can_sample() >= max(self.min_steps_before_training, 1)
        return self.split_agree(ok) if self.split_agree is not None else ok

    def _train_on_policy(self, start_epoch):
        """The on-policy branch (torch_rl_algorithm.py:30-32): every `num_steps_between_train_calls` env steps the trainer
        consumes ALL samples collected since the last call and the buffer is cleared.  Here the samples never leave the
        device: trainer.train_from_rollout runs the vec steps, GAE and the minibatch epochs."""
        ctx = self.trainer.ctx
        horizon = max(1, self.num_steps_between_train_calls // self.env_num)
        t_start = time.perf_counter()
        for epoch in range(start_epoch, self.num_epochs + 1):
            t_epoch = time.perf_counter()
            self.training_env.rollout_stats(reset=True)
            steps = 0
            while steps < self.num_env_steps_per_epoch:
                steps += self.trainer.train_from_rollout(self.training_env, horizon, self.max_path_length,
                                                         bootstrap=self.bootstrap_open_segments)
                self._n_train_steps_total += 1
                self._n_grad_steps_total += self.num_train_steps_per_train_call
            ctx.sync()
            self._n_env_steps_total += steps
            self._t_train = time.perf_counter() - t_epoch   # rollout + update are one device pipeline here
            self._t_sample = 0.0
            if hasattr(self.eval_env, "sync_obs_rms"):
                self.eval_env.sync_obs_rms()
            t0 = time.perf_counter()
            self.evaluate(epoch, time.perf_counter() - t_epoch, time.perf_counter() - t_start)
            self._t_eval = time.perf_counter() - t0
            self.trainer.end_epoch()

    # ---- the pieces of one off-policy epoch (base_algorithm.py:183-286), shared by train() and DeviceRLAlgorithmGroup
    def _vec_step(self, begin_only=False):
        """One sampling iteration of all envs: policy -> physics -> replay record -> auto-reset.  begin_only: enqueue it and return; the
        caller ends it with _vec_step_end() (which, with insert_at_episode_end, waits for the step and inserts the finished episodes)."""
        random_actions = self.replay_buffer.num_steps_can_sample() < self.min_steps_before_training
        label = getattr(self.trainer, "expert_policy", None)
        self._vec_step_open = bool(begin_only) and label is None and hasattr(self.training_env, "rollout_step_end")
        kw = dict(begin_only=True) if self._vec_step_open else {}
        self.training_env.rollout_step(self.exploration_policy, self.replay_buffer, self.max_path_length,
                                       random_actions=random_actions, no_terminal=self.no_terminal, label_policy=label, **kw)
        self._n_env_steps_total += self.env_num

    def _vec_step_end(self):
        if getattr(self, "_vec_step_open", False):
            self.training_env.rollout_step_end()
            self._vec_step_open = False

    def _train_due(self):
        return self._n_env_steps_total - self._n_prev_train_env_steps >= self.num_steps_between_train_calls

    def _count_train_call(self):
        self._n_prev_train_env_steps = self._n_env_steps_total   # _try_to_train (base_algorithm.py:293-299): the gate only advances when training ran
        self._n_train_steps_total += 1                                   # the reference counts CALLS here (:298)
        self._n_grad_steps_total += self.num_train_steps_per_train_call

    def train(self, start_epoch=0):
        if self.on_policy:
            return self._train_on_policy(start_epoch)
        ctx = self.trainer.ctx
        pool = None
        t_start = time.perf_counter()
        for epoch in range(start_epoch, self.num_epochs + 1):  # num_epochs + 1 (base_algorithm.py:64)
            t_epoch = time.perf_counter()
            self._t_sample = self._t_train = 0.0
            self.training_env.rollout_stats(reset=True)
            for _ in range(self.num_env_steps_per_epoch // self.env_num):
                t0 = time.perf_counter()
                self._vec_step()
                if self._train_due():
                    ctx.sync()
                    t1 = time.perf_counter()
                    self._t_sample += t1 - t0
                    if self._can_train():
                        self._count_train_call()
                        self.trainer.train_from_replay(self.replay_buffer, self.num_train_steps_per_train_call, self.batch_size)
                        ctx.sync()
                    self._t_train += time.perf_counter() - t1
                else:
                    self._t_sample += time.perf_counter() - t0
            ctx.sync()
            t0 = time.perf_counter()
            if self.eval_async is not None:
                if pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    pool = ThreadPoolExecutor(max_workers=1)
                self._eval_async_start(pool, epoch, t0 - t_epoch, t0 - t_start)
            else:
                self.evaluate(epoch, time.perf_counter() - t_epoch, time.perf_counter() - t_start)
                self._t_eval = time.perf_counter() - t0
            self.trainer.end_epoch()
        if pool is not None:
            self._eval_async_join()
            pool.shutdown()

    def _eval_collect(self):
        """The evaluation rollouts (device work + host waits, no logging): what evaluate() reduces and logs."""
        if isinstance(self.eval_sampler, DeviceEvalSampler):
            return self.eval_sampler.obtain_statistics(stat_prefix="Test")
        return self.eval_sampler.obtain_samples()

    # ---- eval_async: the evaluation of an epoch overlapped with the next epoch's sampling and training
    def _eval_async_setup(self):
        """(frozen policy copy, sampler on it) when this run can evaluate beside its training: a deterministic device evaluation
        (MakeDeterministic over a network of this library) whose env lives on a context — a stream — of its own (the run scripts build it
        there when rl_alg_params.eval_async is set).  None, with the reason printed, otherwise: the evaluation then blocks as always."""
        s = self.eval_sampler
        why = None
        if type(s) is not DeviceEvalSampler:
            why = "the evaluation is not a DeviceEvalSampler"
        else:
            pi, ppo, det = s._handles()
            if pi is None or not det:
                why = "the evaluation policy is not MakeDeterministic over a device network (eval_deterministic: true)"
            elif s.env.ctx is self.trainer.ctx or s.env.ctx.stream == self.trainer.ctx.stream:
                why = "the eval env shares the training stream"
        if why is not None:
            print(f"DeviceRLAlgorithm(eval_async=True): {why}; evaluating between epochs", flush=True)
            return None
        frozen = s.policy.stochastic_policy.copy(ctx=s.env.ctx)
        return dict(live=s.policy.stochastic_policy, frozen=frozen,
                    sampler=DeviceEvalSampler(s.env, MakeDeterministic(frozen), s.num_steps, s.max_path_length))

    def _eval_freeze(self):
        """the policy as it is now -> the frozen copy (the caller has waited for the training stream)"""
        a = self.eval_async
        a["frozen"].set_flat_params(a["live"].get_flat_params())

    def _eval_begin(self, epoch, epoch_time, total_time):
        """Everything evaluate() logs that is NOT the evaluation rollouts, read at the end of the epoch: trainer statistics, the epoch's
        exploration episodes, counters, times, and the snapshot an asynchronous evaluation will save under its own result."""
        pre = dict(epoch=epoch, epoch_time=epoch_time, total_time=total_time, ts=self.trainer.get_eval_statistics())
        pre["exploration"] = self.training_env.rollout_stats(reset=True)
        self._n_rollouts_total += int(pre["exploration"][0])
        pre["counters"] = (self._n_train_steps_total, self._n_grad_steps_total, self._n_env_steps_total * self.split_world, self._n_rollouts_total)
        pre["times"] = (self._t_train, self._t_eval, self._t_sample)
        return pre

    def _eval_async_start(self, pool, epoch, epoch_time, total_time):
        """End of an epoch in the asynchronous form: log the evaluation that ran beside this epoch, freeze the policy, take what the row and
        the snapshots need, start this epoch's evaluation.  Returns after the hand-over; the training loop goes on."""
        self._eval_async_join()
        self._eval_freeze()
        pre = self._eval_begin(epoch, epoch_time, total_time)
        pre["snapshot"] = self.get_epoch_snapshot()
        pre["extra"] = self.get_extra_data_to_save(epoch)
        pre["t0"] = time.perf_counter()
        sampler = self.eval_async["sampler"]

        def run():
            out = sampler.obtain_statistics(stat_prefix="Test")
            return out, time.perf_counter() - pre["t0"]
        self._eval_pending = (pre, pool.submit(run))

    def _eval_async_join(self):
        if self._eval_pending is not None:
            pre, fut = self._eval_pending
            self._eval_pending = None
            collected, self._t_eval = fut.result()
            self._eval_finish(pre, collected)

    def evaluate(self, epoch, epoch_time, total_time, collected=None):
        pre = dict(epoch=epoch, epoch_time=epoch_time, total_time=total_time, ts=self.trainer.get_eval_statistics())
        if collected is None:
            collected = self._eval_collect()
        pre["exploration"] = self.training_env.rollout_stats(reset=True)
        self._n_rollouts_total += int(pre["exploration"][0])
        pre["counters"] = (self._n_train_steps_total, self._n_grad_steps_total, self._n_env_steps_total * self.split_world, self._n_rollouts_total)
        pre["times"] = (self._t_train, self._t_eval, self._t_sample)
        return self._eval_finish(pre, collected)

    def _eval_finish(self, pre, collected):
        """Reduce and log one epoch's row, save its snapshots (base_algorithm.py:302-348, 599-656).  pre: _eval_begin's dict (+ "snapshot" /
        "extra" taken at the end of the epoch when the evaluation ran beside the next one)."""
        epoch, epoch_time, total_time = pre["epoch"], pre["epoch_time"], pre["total_time"]
        st = OrderedDict()
        ts = pre["ts"]
        if ts:
            st.update(ts)
        if isinstance(self.eval_sampler, DeviceEvalSampler):
            dev_stats = collected
            average_return = dev_stats.pop("AverageReturn")
            st.update(dev_stats)
        else:
            test_paths = collected
            st.update(get_generic_path_information(test_paths, stat_prefix="Test"))
            average_return = get_average_returns(test_paths)
        episodes, ret_sum = pre["exploration"]
        if episodes > 0:
            st["Exploration Returns Mean"] = ret_sum / episodes
            st["Exploration Num Paths"] = episodes
        st["AverageReturn"] = average_return
        lg = self.logger
        for k, v in st.items():
            lg.record_tabular(k, float(np.mean(v)))
        # base_algorithm.py:322-343
        n_calls, n_grad, n_env, n_roll = pre["counters"]
        t_train, t_eval, t_sample = pre["times"]
        lg.record_tabular("Number of train calls total", n_calls)   # the reference's column and counter
        lg.record_tabular("Number of gradient steps total", n_grad)  # ours: calls x steps per call
        lg.record_tabular("Number of env steps total", n_env)
        lg.record_tabular("Number of rollouts total", n_roll)
        lg.record_tabular("Train Time (s)", t_train)
        lg.record_tabular("(Previous) Eval Time (s)", t_eval)
        lg.record_tabular("Sample Time (s)", t_sample)
        lg.record_tabular("Epoch Time (s)", epoch_time)
        lg.record_tabular("Total Train Time (s)", total_time)
        lg.record_tabular("Epoch", epoch)
        lg.dump_tabular()
        snap = dict(epoch=epoch, statistics=dict(st))
        snapshot = (lambda: pre["snapshot"]) if "snapshot" in pre else self.get_epoch_snapshot   # the state the evaluated policy belongs to
        if self.freq_saving and epoch % self.freq_saving == 0:
            lg.save("params.pkl", dict(snap, **snapshot()))
        if self.save_epoch:   # base_algorithm.py:647-649
            lg.save(f"epoch{epoch}.pkl", dict(snap, **snapshot()))
        if st[self.best_key] > self.best_statistic_so_far:
            self.best_statistic_so_far = st[self.best_key]
            if self.save_best and epoch >= self.save_best_starting_from_epoch:   # :650-656
                lg.save("best.pkl", dict(snap, **snapshot()))
        extra = pre["extra"] if "extra" in pre else self.get_extra_data_to_save(epoch)
        if "extra" in pre:
            extra["best_statistic_so_far"] = self.best_statistic_so_far
        lg.save("extra_data.pkl", extra)
        return st

    # ---- snapshots / resume (base_algorithm.py:560-597; logger.load_from_file -> ilswiss_amd/snapshot.py)
    def get_epoch_snapshot(self):
        """trainer.get_snapshot() + the training env's running observation statistics when it normalises (a PPO policy only
        makes sense on the observations it was trained on; the reference pickles the env's obs_rms with the algorithm)."""
        snap = dict(self.trainer.get_snapshot())
        rms = getattr(self.training_env, "obs_rms", None)
        if rms is not None:
            m, v, c = rms._get()
            snap["obs_rms"] = dict(mean=m, var=v, count=c)
        return snap

    def load_snapshot(self, snap):
        self.trainer.load_snapshot(snap)
        if "obs_rms" in snap:
            for env in (self.training_env, self.eval_env):
                if getattr(env, "obs_rms", None) is not None:
                    env.obs_rms.set(snap["obs_rms"]["mean"], snap["obs_rms"]["var"], snap["obs_rms"]["count"])

    def get_extra_data_to_save(self, epoch):
        d = dict(epoch=epoch, _n_env_steps_total=self._n_env_steps_total, _n_train_steps_total=self._n_train_steps_total,
                 _n_grad_steps_total=self._n_grad_steps_total, _n_rollouts_total=self._n_rollouts_total,
                 _n_prev_train_env_steps=self._n_prev_train_env_steps, best_statistic_so_far=self.best_statistic_so_far)
        if self.save_replay_buffer and self.replay_buffer is not None:   # base_algorithm.py:574-576
            from .snapshot import dump_replay
            d["replay_buffer"] = dump_replay(self.replay_buffer)
        return d

    def set_steps(self, extra):   # base_algorithm.py:591-597
        for k in ("_n_env_steps_total", "_n_train_steps_total", "_n_grad_steps_total", "_n_rollouts_total",
                  "_n_prev_train_env_steps", "best_statistic_so_far"):
            if k in extra:
                setattr(self, k, extra[k])


class DeviceRLAlgorithmGroup:
    """K independent off-policy runs (seeds) of one experiment advanced in lock-step by ONE process on ONE GPU — what the reference does with K
    worker processes per GPU (run_experiment.py:57-78), here with every stage of the gradient step being one launch for all K runs
    (SoftActorCriticGroup / ilsx_sac_group, SURVEY section 8e "co-resident seeds").  Every run stays an ordinary DeviceRLAlgorithm with its own
    envs, replay ring, trainer, sibling context (its own Philox key and stream ids) and log directory: progress.csv / variant.json / params.pkl
    per seed, as K separate processes would write them (launcher_util.py:209-297).  The arithmetic of a run does not depend on its company
    (tests/test_hip_parity.py::test_sac_group_lockstep_is_bitwise_the_independent_runs), so the non-time columns of every progress.csv are the
    ones its single-process run writes (tests/test_loop_hip.py::test_grouped_run_script_writes_the_single_run_logs).

    The K runs share the schedule (epochs, steps per epoch, env_num, train trigger); whether a run CAN train at a trigger is its own matter
    (`insert_at_episode_end`: its ring fills when ITS episodes end) — runs that cannot yet are skipped, exactly as their own process would.
    Trainers that are not SoftActorCritic (or differ in shape) are stepped one after the other on the shared stream.  The time columns are the
    group's wall time for the phase (all K runs advance in it)."""

    SCHEDULE = ("num_epochs", "num_env_steps_per_epoch", "env_num", "num_steps_between_train_calls", "num_train_steps_per_train_call",
                "batch_size", "on_policy")

    def __init__(self, algorithms):
        self.algs = list(algorithms)
        if not self.algs:
            raise ValueError("DeviceRLAlgorithmGroup: no runs")
        a0 = self.algs[0]
        for a in self.algs[1:]:
            for k in self.SCHEDULE:
                if getattr(a, k) != getattr(a0, k):
                    raise ValueError(f"DeviceRLAlgorithmGroup: the runs of a group share one schedule; {k} differs ({getattr(a, k)} vs {getattr(a0, k)})")
        if a0.on_policy:
            raise NotImplementedError("DeviceRLAlgorithmGroup steps off-policy runs (the on-policy branch is one device pipeline per run)")
        self.ctx = a0.trainer.ctx
        self._lockstep_args = self._plain_rollouts()
        self._groups = {}   # subset of run indices -> SoftActorCriticGroup (the whole set in steady state; subsets only while rings fill)

    def _plain_rollouts(self):
        """ctypes arrays for ilsx_rollout_steps_lockstep when every run samples through the plain fused rollout (a HIP vec env, a device
        policy, a device ring, no expert relabelling, one max_path_length / no_terminal for all); None otherwise."""
        import ctypes as C
        algs, a0 = self.algs, self.algs[0]
        ok = all(hasattr(a.training_env, "h") and hasattr(a.exploration_policy, "h") and hasattr(a.replay_buffer, "h") and
                 getattr(a.trainer, "expert_policy", None) is None and a.max_path_length == a0.max_path_length and
                 a.no_terminal == a0.no_terminal and type(a.training_env).rollout_step is type(a0.training_env).rollout_step and
                 not hasattr(a.exploration_policy, "_ppo") for a in algs)
        if not ok or not hasattr(a0.trainer.ctx.lib, "ilsx_rollout_steps_lockstep"):
            return None
        from .envs.vecenv import HipVectorEnv
        if not all(type(a.training_env) is HipVectorEnv for a in algs):
            return None
        K = len(algs)
        arr = lambda xs: (C.c_void_p * K)(*[x.h for x in xs])  # noqa: E731
        return (arr([a.training_env for a in algs]), arr([a.exploration_policy for a in algs]), arr([a.replay_buffer for a in algs]),
                (C.c_int64 * K)(*[int(a.min_steps_before_training) for a in algs]), a0.trainer.ctx.lib)

    def _eval_lockstep(self, samplers=None):
        """All runs' evaluation rollouts in one library call (ilsx_eval_rollouts_lockstep: one launch per stage for the runs' small eval envs);
        None when the runs do not evaluate through the same device sampler or cannot share launches."""
        import ctypes as C

        from . import _lib
        algs, a0 = self.algs, self.algs[0]
        ss = samplers if samplers is not None else [getattr(a, "eval_sampler", None) for a in algs]
        if len(algs) < 2 or not all(type(s_) is DeviceEvalSampler for s_ in ss) or not hasattr(a0.trainer.ctx.lib, "ilsx_eval_rollouts_lockstep"):
            return None
        hs = [s_._handles() for s_ in ss]
        if any(h[0] is None for h in hs) or len({(s_.num_steps, s_.max_path_length, h[2]) for s_, h in zip(ss, hs)}) != 1:
            return None
        K = len(algs)
        envs = (C.c_void_p * K)(*[s_.env.h for s_ in ss])
        pis = (C.c_void_p * K)(*[h[0] for h in hs])
        st = (C.c_double * (18 * K))()
        lib = a0.trainer.ctx.lib
        rc = lib.ilsx_eval_rollouts_lockstep(envs, pis, K, int(ss[0].max_path_length), int(hs[0][2]), int(ss[0].num_steps), st)
        if rc == _lib.ILSX_ERR_UNSUPPORTED:
            return None
        _lib.check(rc)
        return [s_.stats_dict(st[18 * k:18 * (k + 1)], stat_prefix="Test") for k, s_ in enumerate(ss)]

    def _groupable(self, idx):
        from .sac import SoftActorCritic
        trs = [self.algs[i].trainer for i in idx]
        if len(trs) < 2 or not all(type(t) is SoftActorCritic for t in trs):
            return False
        t0 = trs[0]
        sig = lambda t: (t.policy.obs_dim, t.policy.action_dim, tuple(t.policy.hidden_sizes), tuple(t.qf1.hidden_sizes), t.max_batch,  # noqa: E731
                         t.reward_scale, t.discount)
        return all(sig(t) == sig(t0) for t in trs) and len(t0.policy.hidden_sizes) == 2 and getattr(t0, "grad_world", 1) == 1

    def _train(self, idx):
        a0 = self.algs[0]
        n, B = a0.num_train_steps_per_train_call, a0.batch_size
        key = tuple(idx)
        if key not in self._groups:
            grp = None
            if self._groupable(idx):
                from .sac import SoftActorCriticGroup
                try:
                    grp = SoftActorCriticGroup([self.algs[i].trainer for i in idx], ctx=self.ctx)
                except Exception as e:   # noqa: BLE001 — shapes the grouped kernels do not take: one run after the other, and say so
                    print(f"DeviceRLAlgorithmGroup: runs {list(idx)} are stepped one by one ({e})", flush=True)
            self._groups[key] = grp
        grp = self._groups[key]
        for i in idx:
            self.algs[i]._count_train_call()
        if grp is not None:
            grp.train_from_replay([self.algs[i].replay_buffer for i in idx], n, B)
        else:
            for i in idx:
                self.algs[i].trainer.train_from_replay(self.algs[i].replay_buffer, n, B)

    def sync(self):
        for a in self.algs:
            a.trainer.ctx.sync()
        self.ctx.sync()

    def train(self, start_epoch=0):
        from concurrent.futures import ThreadPoolExecutor
        algs, a0 = self.algs, self.algs[0]
        # the runs' evaluation rollouts are host-driven loops of small launches (one C call per rollout round, a wait every 32 vec steps): one
        # thread per run, each on its run's stream — ctypes drops the GIL inside the calls, the K evaluations overlap on the GPU
        pool = ThreadPoolExecutor(max_workers=len(algs)) if len(algs) > 1 else None
        # eval_async on every run: the K evaluations of an epoch run (in lock-step, on the frozen copies and the eval envs' own streams) beside
        # the next epoch's sampling and training; their K rows are written when they are in
        use_async = all(getattr(a, "eval_async", None) is not None for a in algs)
        apool = ThreadPoolExecutor(max_workers=1) if use_async else None
        pending = None

        def join(pending):
            if pending is not None:
                pres, fut = pending
                collected, t_ev = fut.result()
                for a, pre, c in zip(algs, pres, collected):
                    a._t_eval = t_ev
                    a._eval_finish(pre, c)
            return None
        t_start = time.perf_counter()
        for epoch in range(start_epoch, a0.num_epochs + 1):
            t_epoch = time.perf_counter()
            t_sample = t_train = 0.0
            for a in algs:
                a.training_env.rollout_stats(reset=True)
            left = a0.num_env_steps_per_epoch // a0.env_num
            while left > 0:
                t0 = time.perf_counter()
                # vec steps until the first run's train trigger (at least one): one library call for the whole stretch when every run
                # samples through the plain fused rollout, else step by step from here
                n = max(1, min(left, min(-(-(a.num_steps_between_train_calls - (a._n_env_steps_total - a._n_prev_train_env_steps)) // a.env_num)
                                         for a in algs)))
                if self._lockstep_args is not None:
                    envs, pis, rbs, mins, lib = self._lockstep_args
                    from . import _lib
                    _lib.check(lib.ilsx_rollout_steps_lockstep(envs, pis, rbs, len(algs), n, int(a0.max_path_length), mins, 0, int(a0.no_terminal)))
                    for a in algs:
                        a._n_env_steps_total += n * a.env_num
                else:
                    n = 1
                    for a in algs:          # enqueue every run's vec step on its own stream ...
                        a._vec_step(begin_only=True)
                    for a in algs:          # ... then do the host part of each (insert_at_episode_end: wait + insert the finished episodes)
                        a._vec_step_end()
                left -= n
                due = [i for i, a in enumerate(algs) if a._train_due()]
                if due:
                    self.sync()
                    t1 = time.perf_counter()
                    t_sample += t1 - t0
                    can = [i for i in due if algs[i]._can_train()]
                    if can:
                        self._train(can)
                        self.sync()
                    t_train += time.perf_counter() - t1
                else:
                    t_sample += time.perf_counter() - t0
            self.sync()
            t_eval0 = time.perf_counter()
            if use_async:
                pending = join(pending)
                pres = []
                for a in algs:
                    a._t_sample, a._t_train = t_sample, t_train
                    a._eval_freeze()
                    pre = a._eval_begin(epoch, t_eval0 - t_epoch, t_eval0 - t_start)
                    pre["snapshot"], pre["extra"] = a.get_epoch_snapshot(), a.get_extra_data_to_save(epoch)
                    pres.append(pre)
                    a.trainer.end_epoch()
                samplers = [a.eval_async["sampler"] for a in algs]

                def run(samplers=samplers, t0=t_eval0):
                    collected = self._eval_lockstep(samplers)
                    if collected is None:
                        collected = list(pool.map(lambda s_: s_.obtain_statistics(stat_prefix="Test"), samplers)) if pool else \
                            [samplers[0].obtain_statistics(stat_prefix="Test")]
                    return collected, time.perf_counter() - t0
                pending = (pres, apool.submit(run))
                continue
            collected = self._eval_lockstep()
            if collected is None:
                collected = list(pool.map(lambda a: a._eval_collect(), algs)) if pool else [algs[0]._eval_collect()]
            t_eval = time.perf_counter() - t_eval0
            for a, c in zip(algs, collected):
                a._t_sample, a._t_train = t_sample, t_train
                a.evaluate(epoch, t_eval0 - t_epoch, t_eval0 - t_start, collected=c)
                a._t_eval = t_eval
                a.trainer.end_epoch()
        join(pending)
        if apool:
            apool.shutdown()
        if pool:
            pool.shutdown()

    def close(self):
        for g in self._groups.values():
            if g is not None:
                g.close()
        self._groups = {}


def setup_log_dir(exp_name, exp_id, seed, variant, base_dir="logs"):
    """logs/<exp-name>/<exp_name>_<timestamp>_<id>--s-<seed>/ + variant.json (launcher_util.py:209-297)."""
    ts = time.strftime("%Y_%m_%d_%H_%M_%S")
    d = os.path.join(base_dir, exp_name.replace("_", "-"), f"{exp_name}_{ts}_{exp_id:04d}--s-{seed}")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "variant.json"), "w") as f:
        json.dump(variant, f, indent=2, sort_keys=True, default=str)
    return d
