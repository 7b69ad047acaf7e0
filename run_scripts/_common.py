"""Shared plumbing of the run scripts (the reference repeats it in every run_scripts/*_exp_script.py): variant loading
(what run_experiment.py writes per grid point, or a full exp_spec whose first grid point is taken), seeding
(launcher_util.py:330-344), env construction (rlkit/envs/__init__.py:72-132) and the log directory
(launcher_util.py:209-297)."""
import argparse
import os
import sys

import numpy as np
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.algorithm import setup_log_dir  # noqa: E402
from ilswiss_amd.envs.vecenv import get_envs  # noqa: E402
from ilswiss_amd.launcher import variants  # noqa: E402


def flatten_spec(spec):
    """A flat variant passes through; a full exp_spec yields its first grid point."""
    if "constants" not in spec:
        return spec
    return next(variants(spec))


def make_envs(variant, ctx, **vec_kwargs):
    """training env (env_num / training_env_num envs) + a small eval env (its paths are walked on the host)."""
    seed = int(variant.get("seed", 0))
    env_specs = dict(variant["env_specs"])
    n_train = int(env_specs.get("env_num", env_specs.get("training_env_num", 1)))
    n_eval = int(env_specs.get("eval_env_num", min(n_train, 16)))
    if _SPLIT is not None:       # a rank of a split run steps its share of the envs, seeded apart from the other ranks' shares
        if n_train % _SPLIT.world:
            raise ValueError(f"env_num={n_train} does not split over split_ranks={_SPLIT.world}")
        n_train //= _SPLIT.world
        seed += 1000003 * _SPLIT.rank
    training_env = get_envs(dict(env_specs, env_num=n_train, training_env_seed=seed), ctx=ctx, **vec_kwargs)
    eval_kwargs = dict(vec_kwargs)
    if vec_kwargs.get("norm_obs"):   # ppo_exp_script.py:68-75: the eval env shares the statistics and does not update them
        eval_kwargs.update(obs_rms=training_env.obs_rms, update_obs_rms=False)
    # rl_alg_params.eval_async: the eval env gets a context (a HIP stream) of its own, so that DeviceRLAlgorithm can evaluate an epoch on a
    # frozen policy copy beside the next epoch's sampling and training (the env's own seed drives its resets: the context only carries the stream)
    use_async = bool((variant.get("rl_alg_params") or {}).get("eval_async")) and hasattr(ctx, "rng_stream_cursor")
    ectx = ctx.sibling(seed + 10007) if use_async else ctx
    if use_async:
        ectx.rng_stream_cursor(set_to=ctx.rng_stream_cursor())    # the eval env takes the Philox stream id it would have had on the run's ctx ...
    eval_env = get_envs(dict(env_specs, env_num=n_eval, training_env_seed=seed + 10007), ctx=ectx, **eval_kwargs)
    if use_async:
        ctx.rng_stream_cursor(set_to=ectx.rng_stream_cursor())    # ... and the run's ctx skips it: ring and trainer keep their streams
    return training_env, eval_env, training_env.single_env_view()


# ---- several variants in ONE process (run_experiment.py --group K / meta_data.seeds_per_process): `-e a.yaml b.yaml ...`.  Every variant
# goes through the run script's own experiment() unchanged; start() hands run k a sibling context of run 0's (same device and stream, its own
# Philox key and object counter — what it would have in a process of its own) and train() collects the built algorithms instead of training
# them; main() then advances them in lock-step (ilswiss_amd.algorithm.DeviceRLAlgorithmGroup).
_GROUP = None


def start(variant, gpu):
    seed = int(variant.get("seed", 0))
    np.random.seed(seed)
    if _GROUP is not None and _GROUP["ctx"] is not None:
        # ILSX_GROUP_SHARE_STREAM=1 (A/B): every run of the group on run 0's stream — the runs' small rollout launches then queue up behind
        # each other (measured on ten 4-env Hopper runs: 4.9 s of sampling per epoch against 1.7 s on a stream per run)
        ctx = _GROUP["ctx"].sibling(seed, share_stream=bool(os.environ.get("ILSX_GROUP_SHARE_STREAM")))
        ia.device.set_default_context(ctx)
        return ctx
    # a split run: the ctx key drives the policy noise of this rank's rows — the rank is mixed in, or every shard would draw the same eps rows
    # (G-fold correlated noise in the batch); the networks' init seeds come from np.random and stay identical on all ranks
    ctx = ia.set_gpu_mode(True, gpu, seed=seed + (7919 * _SPLIT.rank if _SPLIT else 0))
    if _GROUP is not None:
        _GROUP["ctx"] = ctx
    return ctx


def train(algorithm, variant):
    """`load_params` (sac_alpha_exp_script.py:106-113): resume from <load_path>/{params,extra_data}.pkl, then train."""
    from ilswiss_amd.snapshot import load_from_file
    epoch = 0
    if variant.get("load_params"):
        algorithm, epoch = load_from_file(algorithm, **variant["load_params"])
    print("Start from epoch", epoch)
    if _GROUP is not None:
        _GROUP["runs"].append((algorithm, epoch))
        return algorithm
    algorithm.train(start_epoch=epoch)
    return algorithm


# ---- one run split over G GPUs (rl_alg_params.split_ranks: G; SURVEY section 8e, BASELINE config 5 "RCCL grad all-reduce per run").  Every rank is a
# process with a full replica of the networks, env_num / G envs, a ring of replay_buffer_size / G rows and B / G rows of every batch; the
# library all-reduces the gradient arena between backward and update (ilswiss_amd.parallel.SplitRunStep).  The ranks come from a launcher
# (`torchrun --nproc-per-node G run_scripts/sac_alpha_exp_script.py -e v.yaml`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env) or, when
# there is no RANK in the env, from main() itself: it re-executes the script G times, rank r on GPU g + r, rendezvous on 127.0.0.1.
def split_ranks_of(variant):
    g = int((variant.get("rl_alg_params") or variant.get("adv_irl_params") or {}).get("split_ranks", 1) or 1)   # (adversarial IRL keeps its loop keys in adv_irl_params)
    return g if (g > 1 or os.environ.get("ILSX_SPLIT_FORCE")) else 0     # ILSX_SPLIT_FORCE: the split path on a one-rank communicator (tests)


from ilswiss_amd.parallel import SplitInfo  # noqa: E402


_SPLIT = None


def split_info():
    return _SPLIT


def _split_spawn(world, gpu):
    """No launcher: `world` copies of this command, rank r on GPU gpu + r.  A failing rank stops the others."""
    import socket
    import subprocess
    import time
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ILSX_SPLIT_GPU0=str(gpu))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env))
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad:
            rc = bad[0].returncode
            for p in procs:
                if p.poll() is None:
                    p.kill()          # exactly the PIDs started above
            break
        time.sleep(0.2)
    return rc or max(p.wait() for p in procs)


def _split_join(world, gpu):
    """This process is rank RANK of `world`: process group (RCCL), GPU, and the SplitInfo the run script reads."""
    global _SPLIT
    import torch
    import torch.distributed as dist
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541")):
        os.environ.setdefault(k, v)
    rank, have = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if have != max(world, 1):
        raise SystemExit(f"rl_alg_params.split_ranks={world} but WORLD_SIZE={have}")
    local = int(os.environ.get("ILSX_SPLIT_GPU0", gpu)) + int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    _SPLIT = SplitInfo(have, rank, dist)
    return local


def _log_dir(variant, default_name):
    load_path = (variant.get("load_params") or {}).get("load_path")
    if load_path:   # a resumed run keeps logging into the directory it resumes from (sac_alpha_exp_script.py:142-146)
        return load_path
    return setup_log_dir(variant.get("exp_name", default_name), int(variant.get("exp_id", 0)), int(variant.get("seed", 0)), variant)


def main(experiment, default_name):
    global _GROUP
    ap = argparse.ArgumentParser()
    ap.add_argument("-e", "--experiment", required=True, nargs="+",
                    help="experiment specification file; several files = that many runs in this one process, stepped in lock-step")
    ap.add_argument("-g", "--gpu", type=int, default=0, help="gpu id")
    args = ap.parse_args()
    variants_ = []
    for path in args.experiment:
        with open(path) as f:
            variants_.append(flatten_spec(yaml.safe_load(f)))
    if len(variants_) == 1:
        v, gpu = variants_[0], args.gpu
        world = split_ranks_of(v)
        if world:
            if "RANK" not in os.environ and world > 1:
                raise SystemExit(_split_spawn(world, gpu))
            gpu = _split_join(world, gpu)
        try:
            # only rank 0 of a split run logs (the replicas are identical by construction)
            return experiment(v, gpu, _log_dir(v, default_name) if not _SPLIT or _SPLIT.rank == 0 else None)
        finally:
            if _SPLIT is not None and _SPLIT.dist is not None:
                _SPLIT.dist.barrier()
                _SPLIT.dist.destroy_process_group()
    if any(split_ranks_of(v) for v in variants_):
        raise SystemExit("split_ranks and several runs per process (--group) do not combine: a split run owns its GPUs")
    from ilswiss_amd.algorithm import DeviceRLAlgorithmGroup
    _GROUP = dict(ctx=None, runs=[])
    for v in variants_:
        experiment(v, args.gpu, _log_dir(v, default_name))
    runs, _GROUP = _GROUP["runs"], None
    epochs = {e for _, e in runs}
    if len(epochs) != 1:
        raise SystemExit(f"grouped runs must resume from the same epoch (found {sorted(epochs)})")
    group = DeviceRLAlgorithmGroup([a for a, _ in runs])
    group.train(start_epoch=epochs.pop())
    return group
