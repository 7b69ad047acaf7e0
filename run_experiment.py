#!/usr/bin/env python
"""Launcher with the reference's contract (run_experiment.py:12-78): `python run_experiment.py -e <exp_spec.yaml> -g <gpu>`
expands `variables` (nested dicts of value lists) into the Cartesian grid of variants, writes each as
logs/variants-for-<exp_name>/variants-<timestamp>/<i>.yaml (constants + grid point + meta_data + exp_id) and runs
`python <script_path> -e <variant.yaml> -g <gpu>` with at most `num_workers` processes at a time.

MI355X addition: `--gpus N` spreads the variants round-robin over GPUs g, g+1, ... g+N-1 — the "independent seeds shard
across the 8 GPUs of a node, no collective" path of SURVEY §8e (one process and one libilsx context per GPU)."""
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
    workers = max(1, min(int(meta.get("num_workers", 1)) * args.gpus, len(paths)))
    # Like the reference (run_experiment.py:57-78) all `num_workers` children of a GPU run on it at once.  A single run's train windows use
    # merged "phase" kernels that want the GPU to themselves (include/ilsx.h ilsx_sac_phase_state); with company the library would notice,
    # roll the first window back and leave them by itself — telling the children up front saves them that one bounded wait.
    child_env = dict(os.environ)
    if workers > args.gpus:
        child_env.setdefault("ILSX_NO_PHASE", "1")
    running, nxt, failed = [], 0, 0
    while nxt < len(paths) or running:
        while nxt < len(paths) and len(running) < workers:
            cmd = [sys.executable, meta["script_path"], "-e", paths[nxt], "-g", str(args.gpu + nxt % args.gpus)]
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
