#!/usr/bin/env python
"""Acceptance runs of the grouped entry point: `run_experiment.py -e <spec> --group K` from a scratch directory, wall time
and the aggregate gradient-step rate read back from the K progress.csv files it writes.

  python tools/grouped_entry_rate.py exp_specs/sac/sac_humanoid_hip.yaml --group 4 --epochs 3 [--set rl_alg_params.num_steps_per_eval=2000]

Prints one JSON line: wall seconds of the launcher, per-epoch aggregate grad-steps/s in the train phase and over the whole epoch (from the
"Train Time (s)" / "Epoch Time (s)" columns of seed 0 — the group's wall time for the phase — and the K runs' gradient-step counts), final
returns per seed."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("spec")
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--seeds", type=int, default=None, help="first N seeds of the spec")
    ap.add_argument("--set", action="append", default=[], help="constants override, dotted key=value (yaml value)")
    ap.add_argument("--keep", default=None, help="copy the progress.csv files here (one per seed)")
    args = ap.parse_args()
    spec = yaml.safe_load(open(os.path.join(ROOT, args.spec)))
    spec["meta_data"]["script_path"] = os.path.join(ROOT, spec["meta_data"]["script_path"])
    if args.epochs is not None:
        spec["constants"]["rl_alg_params"]["num_epochs"] = args.epochs
    if args.seeds is not None:
        spec["variables"]["seed"] = spec["variables"]["seed"][:args.seeds]
    for kv in args.set:
        k, v = kv.split("=", 1)
        d = spec["constants"]
        path = k.split(".")
        for p in path[:-1]:
            d = d[p]
        d[path[-1]] = yaml.safe_load(v)
    wd = tempfile.mkdtemp(prefix="grp_entry_")
    sp = os.path.join(wd, "spec.yaml")
    with open(sp, "w") as f:
        yaml.dump(spec, f)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_experiment.py"), "-e", sp, "--group", str(args.group)], cwd=wd,
                       capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-4000:], file=sys.stderr)
        raise SystemExit(r.returncode)
    runs = {}
    for d in sorted(glob.glob(os.path.join(wd, "logs", "*", "*--s-*"))):
        seed = int(d.rsplit("--s-", 1)[1])
        runs[seed] = list(csv.DictReader(open(os.path.join(d, "progress.csv"))))
        if args.keep:
            os.makedirs(args.keep, exist_ok=True)
            with open(os.path.join(d, "progress.csv")) as src, open(os.path.join(args.keep, f"seed{seed}.csv"), "w") as dst:
                dst.write(src.read())
    K = len(runs)
    s0 = runs[min(runs)]
    per_epoch = []
    prev = 0.0
    for i, row in enumerate(s0):
        g = sum(float(runs[s][i]["Number of gradient steps total"]) for s in runs)
        dg = g - prev
        prev = g
        tt, te = float(row["Train Time (s)"]), float(row["Epoch Time (s)"])
        per_epoch.append(dict(epoch=int(float(row["Epoch"])), grad_steps=dg, train_phase_rate=dg / tt if tt > 0 else None, epoch_rate=dg / te if te > 0 else None,
                              sample_s=float(row["Sample Time (s)"]), train_s=tt, epoch_s=te))
    steady = [e for e in per_epoch[1:] if e["grad_steps"] > 0] or per_epoch
    out = dict(spec=args.spec, seeds=K, group=args.group, epochs=len(s0), launcher_wall_s=wall,
               aggregate_grad_steps_per_s_train_phase=sum(e["grad_steps"] for e in steady) / max(1e-9, sum(e["train_s"] for e in steady)),
               aggregate_grad_steps_per_s_whole_epoch=sum(e["grad_steps"] for e in steady) / max(1e-9, sum(e["epoch_s"] for e in steady)),
               env_steps_total_per_seed=float(s0[-1]["Number of env steps total"]),
               final_returns={s: float(runs[s][-1]["AverageReturn"]) for s in runs}, per_epoch=per_epoch[:6])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
