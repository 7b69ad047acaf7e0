#!/usr/bin/env python
"""The CPU restatement run END TO END on the reference's own loop: who owns the return gap — the simulator or
the engine?

Everything here is the checker side of the repo, none of it the product: oracle/sac_alpha_torch.py (the SAC-alpha step in the
reference's idiom: autograd + torch.optim.Adam; pinned by the reference-generated g4 fixtures), oracle/replay.py (the ring buffer with
its numpy RandomState index stream, pinned by g10/g19/g23), oracle/mlp.py's init rule (pinned by g2) and oracle/planar_env.c (the
fp64 scalar C statement of this repo's planar Hopper, the function the HIP stepper is bit-checked against to 1e-8).  The loop is
BaseAlgorithm.start_training (/root/reference/rlkit/core/base_algorithm.py:166-291) stated again for exactly the reference's
sac_hopper.yaml:17-47:

  * 4 training envs stepped together, one PathBuilder per env (:171-176); every env step counts env_num steps (:199)
  * samples enter the replay ring only when their episode ends: _handle_vec_step(add_buf=False) fills the path builders (:204-218,
    :423-466), _handle_vec_rollout_ending -> _handle_path -> add_sample + terminate_episode (:509-519, :399-421)
  * a terminal ends its env's path that step; otherwise (elif, :262-276) every path of length >= max_path_length ends
  * no warm-up: min_steps_before_training 0, so actions come from the stochastic policy from step 0 (:371-380)
  * a train call of 1000 gradient steps (batch 512, RandomState indices) once 1000 env steps have passed (:280-286, :293-299)
  * evaluation after every epoch: MakeDeterministic(policy) on the 4-env eval vec env, whole rollouts of all 4 envs until
    >= 10000 steps have been seen (rlkit/samplers/vec_sampler.py:5-97,126-146); "Test Returns Mean" = mean path return
    (rlkit/core/eval_util.py:get_average_returns)

Only the simulator is this repo's own (MuJoCo is not in the image), and it is THE SAME simulator the HIP engine runs: if the
two engines' return distributions agree, the gap to the README figure is a property of that simulator; if they do not, the HIP engine
has a bug the step-level fixtures cannot see.

    python tools/returns_cpu.py --seed 3 --out gpurun_out/returns_cpu/seed3.csv            # 1 thread, ~3-4 h on a 2.1 GHz Xeon core
    python tools/returns_cpu.py --seeds 0 1 2 3 4 5 --jobs 6 --outdir profiles/r05_returns_cpu   # side by side, one thread each
"""
import argparse
import csv
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_stepper():
    so = os.path.join(ROOT, "oracle", "_build", "liborc_planar.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.orc_planar_step.restype = C.c_int
    return lib


class CpuVecEnv:
    """n planar envs on oracle/planar_env.c; reset rule of oracle/planar_env.py:186-192 (gym HopperEnv.reset_model)."""

    def __init__(self, lib, model, n, rng):
        from ilswiss_amd.envs.vecenv import model_struct
        self.lib, self.m, self.ms, self.n, self.rng = lib, model, model_struct(model), n, rng
        self.nq = model["n_body"] + 2
        self.o, self.a = 2 * self.nq - 1, len(model["act_bodies"])
        self.q, self.v = np.zeros((n, self.nq)), np.zeros((n, self.nq))
        self._ob, self._r, self._d = np.empty(self.o), C.c_double(), C.c_int()

    def _obs(self, i):
        clip = self.m.get("qvel_clip", 0.0)
        v = np.clip(self.v[i], -clip, clip) if clip and clip > 0 else self.v[i]
        return np.concatenate([self.q[i, 1:], v])

    def reset(self, ids):
        nz, out = self.m["reset_noise"], []
        for i in ids:
            self.q[i] = np.asarray(self.m["init_qpos"], np.float64) + self.rng.uniform(-nz, nz, self.nq)
            self.v[i] = self.rng.uniform(-nz, nz, self.nq)
            out.append(self._obs(i))
        return np.array(out)

    def step(self, actions, ids):
        p = lambda x: x.ctypes.data_as(C.c_void_p)   # noqa: E731
        obs, rew, done = [], [], []
        for a, i in zip(actions, ids):
            a = np.ascontiguousarray(a, np.float64)
            self.lib.orc_planar_step(C.byref(self.ms), p(self.q[i]), p(self.v[i]), p(a), p(self._ob), C.byref(self._r), C.byref(self._d))
            obs.append(self._ob.copy()); rew.append(self._r.value); done.append(bool(self._d.value))
        return np.array(obs), np.array(rew), np.array(done)


def run(seed, out_csv, epochs=102, steps_per_epoch=10000, between=1000, per_call=1000, batch=512, env_num=4, eval_steps=10000,
        max_path_length=1000, replay_size=1000000, net=256, layers=2, quiet=False):
    import torch
    torch.set_num_threads(1)
    from ilswiss_amd.envs.models import MODELS
    from oracle import mlp as omlp
    from oracle.replay import ReplayOracle
    from oracle.sac_alpha_torch import SacAlphaTorch, _mlp

    # set_seed(seed) of the reference (run_scripts/sac_alpha_exp_script.py:150-151) seeds numpy / torch / random; here: one
    # numpy Generator per consumer, all derived from the seed
    ss = np.random.SeedSequence(seed)
    r_init, r_env, r_eval, r_act, r_train, r_buf = [np.random.default_rng(s) for s in ss.spawn(6)]
    lib, model = load_stepper(), MODELS["hopper"]()
    env, eval_env = CpuVecEnv(lib, model, env_num, r_env), CpuVecEnv(lib, model, env_num, r_eval)
    o, a, hidden = env.o, env.a, layers * [net]
    pi = omlp.init_mlp(r_init, o, hidden, a, init_w=1e-3, n_heads=2)          # policies.py:207-237
    q1 = omlp.init_mlp(r_init, o + a, hidden, 1, init_w=3e-3)                 # networks.py:57-83
    q2 = omlp.init_mlp(r_init, o + a, hidden, 1, init_w=3e-3)
    sac = SacAlphaTorch(o, a, hidden, pi, q1, q2, reward_scale=1.0, discount=0.99, policy_lr=3e-4, qf_lr=3e-4, soft_target_tau=0.005,
                        alpha=0.2, policy_mean_reg_weight=1e-3, policy_std_reg_weight=1e-3)    # sac_hopper.yaml:37-47
    buf = ReplayOracle(replay_size, o, a, random_seed=int(r_buf.integers(10000)))               # base_algorithm.py:118-120

    def act(obs, deterministic):   # policies.py:241-246,248-307
        with torch.no_grad():
            mu, ls = _mlp(sac.pi_p, torch.as_tensor(np.asarray(obs, np.float32)), sac.nh, 2)
            if deterministic:
                return torch.tanh(mu).numpy().astype(np.float64)
            ls = torch.clamp(ls, -20.0, 2.0)
            eps = torch.as_tensor(r_act.standard_normal(mu.shape).astype(np.float32))
            return torch.tanh(mu + torch.exp(ls) * eps).numpy().astype(np.float64)

    def evaluate():   # vec_sampler.py:5-97,126-146
        rets, lens, total = [], [], 0
        while total < eval_steps:
            ready = np.arange(env_num)
            obs = eval_env.reset(ready)
            ret, ln = np.zeros(env_num), np.zeros(env_num, int)
            for _ in range(max_path_length):
                nobs, rew, term = eval_env.step(act(obs, True), ready)
                ret[ready] += rew; ln[ready] += 1
                obs, ready = nobs[~term], ready[~term]
                if len(ready) == 0:
                    break
            rets += list(ret); lens += list(ln); total += int(ln.sum())
        return np.array(rets), np.array(lens)

    os.makedirs(os.path.dirname(os.path.abspath(out_csv)), exist_ok=True)
    f = open(out_csv, "w", newline="")
    w = csv.writer(f)
    w.writerow(["Epoch", "Number of env steps total", "Number of gradient steps total", "Test Returns Mean", "Test Returns Std",
                "Test Num Paths", "Test Path Length Mean", "Exploration Returns Mean", "Exploration Num Paths", "Replay size", "Alpha",
                "QF1 Loss", "Policy Loss", "Total Time (s)"])
    ids = np.arange(env_num)
    obs = env.reset(ids)
    paths = [[] for _ in range(env_num)]
    n_env_steps = n_prev_train = n_grad = 0
    t_start, last = time.time(), None   # `last`: the epoch's eval_statistics — filled by the FIRST train_step after end_epoch (sac_alpha.py:185-190)

    def end_paths(which):   # _handle_vec_rollout_ending (:509-519)
        done_rets = []
        for i in which:
            for row in paths[i]:
                buf.add_sample(*row)
            buf.terminate_episode()
            done_rets.append(sum(r[2] for r in paths[i]))
            paths[i] = []
        return done_rets

    for epoch in range(epochs + 1):   # num_epochs + 1 (base_algorithm.py:64)
        expl = []
        for _ in range(steps_per_epoch // env_num):
            actions = act(obs, False)
            nobs, rew, term = env.step(actions, ids)
            n_env_steps += env_num
            for i in range(env_num):
                paths[i].append((obs[i].astype(np.float32), actions[i].astype(np.float32), rew[i], int(term[i]), nobs[i].astype(np.float32)))
            if term.any():
                which = np.where(term)[0]
                expl += end_paths(which)
                nobs[which] = env.reset(which)
            elif any(len(p) >= max_path_length for p in paths):
                which = [i for i in range(env_num) if len(paths[i]) >= max_path_length]
                expl += end_paths(which)
                nobs[which] = env.reset(which)
            obs = nobs
            if n_env_steps - n_prev_train >= between:
                n_prev_train = n_env_steps
                for _ in range(per_call):
                    b = buf.gather(buf.draw_indices(batch))
                    res = sac.train_step(b, r_train.standard_normal((batch, a)).astype(np.float32),
                                         r_train.standard_normal((batch, a)).astype(np.float32))
                    if last is None:   # "Alpha" is the temperature after that step's own update (sac_alpha.py:160-166,208-212)
                        last = dict(res, alpha=float(sac.alpha))
                n_grad += per_call
        rets, lens = evaluate()
        st = last or {}
        w.writerow([epoch, n_env_steps, n_grad, rets.mean(), rets.std(), len(rets), lens.mean(), np.mean(expl) if expl else "", len(expl),
                    buf.size, st.get("alpha", float(sac.alpha)), st.get("qf1_loss", ""), st.get("policy_loss", ""), time.time() - t_start])
        f.flush()
        last = None   # end_epoch
        if not quiet:
            print(f"seed {seed} epoch {epoch} env {n_env_steps} grad {n_grad} test {rets.mean():.1f} ({len(rets)} paths) "
                  f"expl {np.mean(expl) if expl else float('nan'):.1f} alpha {float(sac.alpha):.4f} t {time.time() - t_start:.0f}s", flush=True)
    f.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--seeds", type=int, nargs="*", default=None)
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--outdir", default=os.path.join(ROOT, "gpurun_out", "returns_cpu"))
    ap.add_argument("--epochs", type=int, default=102)
    ap.add_argument("--steps-per-epoch", type=int, default=10000)
    ap.add_argument("--eval-steps", type=int, default=10000)
    args = ap.parse_args()
    if args.seeds:
        procs = []
        for s in args.seeds:
            cmd = [sys.executable, os.path.abspath(__file__), "--seed", str(s), "--out", os.path.join(args.outdir, f"seed{s}.csv"),
                   "--epochs", str(args.epochs), "--steps-per-epoch", str(args.steps_per_epoch), "--eval-steps", str(args.eval_steps)]
            os.makedirs(args.outdir, exist_ok=True)
            procs.append(subprocess.Popen(cmd, stdout=open(os.path.join(args.outdir, f"seed{s}.log"), "w"), stderr=subprocess.STDOUT,
                                          env=dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")))
            while args.jobs and sum(p.poll() is None for p in procs) >= args.jobs:
                time.sleep(5)
        sys.exit(max(p.wait() for p in procs))
    run(args.seed, args.out or os.path.join(args.outdir, f"seed{args.seed}.csv"), epochs=args.epochs,
        steps_per_epoch=args.steps_per_epoch, eval_steps=args.eval_steps)


if __name__ == "__main__":
    main()
