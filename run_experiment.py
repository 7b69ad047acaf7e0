#!/usr/bin/env python
"""Launcher with the reference's contract (run_experiment.py:12-78): `python run_experiment.py -e <exp_spec.yaml> -g <gpu>`
expands `variables` (nested dicts of value lists) into the Cartesian grid of variants, writes each as
logs/variants-for-<exp_name>/variants-<timestamp>/<i>.yaml (constants + grid point + meta_data + exp_id) and runs
`python <script_path> -e <variant.yaml> -g <gpu>` with at most `num_workers` processes at a time.

MI355X additions:
  `--gpus N` spreads the children round-robin over GPUs g, g+1, ... g+N-1 — the "independent seeds shard across the 8 GPUs of a node, no
  collective" path of SURVEY §8e (one process and one libilsx context per GPU);
  `--group K` (or `meta_data.seeds_per_process: K` in the spec) hands every child K consecutive variant files: the K runs live in ONE process
  and advance in lock-step, every stage of their gradient steps being one launch for all of them (DeviceRLAlgorithmGroup / ilsx_sac_group —
  BASELINE config 5's "4 seeds per GPU" shape).  Each run still gets its own variant file and its own log directory (variant.json,
  progress.csv, params.pkl), as the reference's K worker processes would write them (run_experiment.py:39-66, launcher_util.py:209-297);
  e.g. `python run_experiment.py -e exp_specs/sac/sac_humanoid_hip.yaml --group 4 --gpus 8` = 32 seeds x 1024 envs over 8 GPUs."""
import argparse
import datetime
import os
import subprocess
import sys
import time

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ilswiss_amd.launcher import variants  # noqa: E402


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-e", "--experiment", required=True, help="experiment specification file")
    ap.add_argument("-g", "--gpu", type=int, default=0, help="(first) gpu id")
    ap.add_argument("--gpus", type=int, default=1, help="spread the variants over this many GPUs")
    ap.add_argument("--group", type=int, default=None,
                    help="runs per child process, stepped in lock-step on one GPU (default: meta_data.seeds_per_process, else 1)")
    ap.add_argument("--log-root", default="logs")
    args = ap.parse_args()
    with open(args.experiment) as f:
        spec = yaml.safe_load(f)
    meta = spec["meta_data"]
    stamp = datetime.datetime.now().strftime("%Y_%m_%d_%H_%M_%S")
    vdir = os.path.join(args.log_root, "variants-for-" + meta["exp_name"], "variants-" + stamp)
    os.makedirs(vdir)
    with open(os.path.join(vdir, "exp_spec_definition.yaml"), "w") as f:
        yaml.dump(spec, f, default_flow_style=False)
    paths = []
    for v in variants(spec):
        paths.append(os.path.join(vdir, "%d.yaml" % v["exp_id"]))
        with open(paths[-1], "w") as f:
            yaml.dump(v, f, default_flow_style=False)
    group = max(1, int(args.group if args.group is not None else meta.get("seeds_per_process", 1)))
    jobs = [paths[i:i + group] for i in range(0, len(paths), group)]   # one child per job; a job's runs share a process, a GPU and a stream
    workers = max(1, min(int(meta.get("num_workers", 1)) * args.gpus, len(jobs)))
    # Like the reference (run_experiment.py:57-78) all `num_workers` children of a GPU run on it at once.  A single run's train windows use
    # merged "phase" kernels that want the GPU to themselves (include/ilsx.h ilsx_sac_phase_state); with company the library would notice,
    # roll the first window back and leave them by itself — telling the children up front saves them that one bounded wait.
    child_env = dict(os.environ)
    if workers > args.gpus:
        child_env.setdefault("ILSX_NO_PHASE", "1")
    running, nxt, failed = [], 0, 0
    while nxt < len(jobs) or running:
        while nxt < len(jobs) and len(running) < workers:
            cmd = [sys.executable, meta["script_path"], "-e", *jobs[nxt], "-g", str(args.gpu + nxt % args.gpus)]
            print(cmd, flush=True)
            running.append(subprocess.Popen(cmd, env=child_env))
            nxt += 1
        time.sleep(0.5)
        still = []
        for p in running:
            rc = p.poll()
            if rc is None:
                still.append(p)
            elif rc != 0:
                failed += 1
        running = still
    sys.exit(1 if failed else 0)
