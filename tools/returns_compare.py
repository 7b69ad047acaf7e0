#!/usr/bin/env python
"""profiles/r05_returns.md: the two engines on the reference's loop, seed for seed.

    python tools/returns_compare.py [--cpu profiles/r05_returns_cpu] [--hip profiles/r05_returns_hip] > profiles/r05_returns_table.md

CPU = tools/returns_cpu.py (oracle/sac_alpha_torch.py + oracle/replay.py + oracle/planar_env.c); HIP = exp_specs/sac/sac_hopper_refloop_hip.yaml
through run_experiment.py (tools/returns_fill.sh).  Same simulator (the HIP stepper is bit-checked against the C stepper's numpy twin to
1e-8), same schedule, same replay order, same evaluation protocol; the random streams differ (numpy generators / Philox), so the comparison
is between DISTRIBUTIONS over seeds: last-10-epoch mean of "Test Returns Mean" per seed, mean / s.e. / median over seeds, Welch's t and the
Mann-Whitney U test between the engines, and how often a seed ends low (< 2500) or never learns (< 1000)."""
import argparse
import csv
import glob
import os

import numpy as np


def load(d):
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "seed*.csv")), key=lambda p: int(os.path.basename(p)[4:-4])):
        rows = list(csv.DictReader(open(f)))
        if not rows:
            continue
        r = np.array([float(x["Test Returns Mean"]) for x in rows])
        steps = int(float(rows[-1]["Number of env steps total"]))
        out[int(os.path.basename(f)[4:-4])] = dict(epochs=len(rows), steps=steps, r=r)
    return out


def summary(name, runs, n_epochs):
    fin = np.array([v["r"][max(0, n_epochs - 10):n_epochs].mean() for v in runs.values()])
    best = np.array([v["r"][:n_epochs].max() for v in runs.values()])
    return dict(name=name, n=len(fin), fin=fin, best=best, mean=fin.mean(), sd=fin.std(ddof=1) if len(fin) > 1 else 0.0,
                se=fin.std(ddof=1) / np.sqrt(len(fin)) if len(fin) > 1 else 0.0, median=np.median(fin), low=int((fin < 2500).sum()),
                dead=int((fin < 1000).sum()))


ROLLOUT_COLUMNS = ("Test ", "Num Paths", "AverageReturn", "Exploration ")   # what the rollouts produce (not trainer statistics, not wall-clock times)


def same(dir_a, dir_b):
    """`--same A B`: are the rollout columns of the runs both directories hold identical, cell for cell?  (The acceptance check of the grouped
    entry point: profiles/r06_returns_grouped vs profiles/r05_returns_hip.)  Returns (seeds compared, differing cells)."""
    seeds, bad = [], 0
    for fa in sorted(glob.glob(os.path.join(dir_a, "seed*.csv")), key=lambda p: int(os.path.basename(p)[4:-4])):
        fb = os.path.join(dir_b, os.path.basename(fa))
        if not os.path.exists(fb):
            continue
        a, b = list(csv.DictReader(open(fa))), list(csv.DictReader(open(fb)))
        cols = [k for k in a[0] if k.startswith(ROLLOUT_COLUMNS) or k in ROLLOUT_COLUMNS]
        d = abs(len(a) - len(b)) * len(cols) + sum(1 for ra, rb in zip(a, b) for k in cols if ra[k] != rb.get(k))
        print(f"seed {os.path.basename(fa)[4:-4]}: {len(a)} / {len(b)} epochs, {len(cols)} rollout columns, {d} differing cells")
        seeds.append(int(os.path.basename(fa)[4:-4]))
        bad += d
    print(f"{len(seeds)} seeds compared, {bad} differing cells")
    return seeds, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--same", nargs=2, metavar=("A", "B"), help="compare the rollout columns of two run directories cell for cell and exit")
    ap.add_argument("--cpu", default="profiles/r05_returns_cpu")
    ap.add_argument("--hip", default="profiles/r05_returns_hip")
    ap.add_argument("--hip-old", default="profiles/r04_returns", help="round 4's HIP runs (per-step insert, 16 eval envs), for the record")
    args = ap.parse_args()
    if args.same:
        seeds, bad = same(*args.same)
        raise SystemExit(0 if seeds and not bad else 1)
    cpu, hip = load(args.cpu), load(args.hip)
    if not cpu or not hip:
        raise SystemExit("no runs found")
    # compare at the epoch every run of both engines has reached (the CPU runs take hours: a table made mid-way says so)
    n_ep = min(min(v["epochs"] for v in cpu.values()), min(v["epochs"] for v in hip.values()))
    full = max(v["epochs"] for v in hip.values())
    print(f"## SAC Hopper on the reference's own loop: CPU restatement vs HIP engine, epoch {n_ep} of {full}"
          + ("" if n_ep == full else "  (PARTIAL: the CPU runs are still going)"))
    print()
    print("| seed | CPU restatement: epochs | last-10 mean | best epoch | HIP engine: epochs | last-10 mean | best epoch |")
    print("|---|---|---|---|---|---|---|")
    for sd in sorted(set(cpu) | set(hip)):
        def cell(runs):
            if sd not in runs:
                return "- | - | -"
            v = runs[sd]
            return f"{v['epochs']} | {v['r'][max(0, n_ep - 10):n_ep].mean():.0f} | {v['r'][:n_ep].max():.0f}"
        print(f"| {sd} | {cell(cpu)} | {cell(hip)} |")
    sc, sh = summary("CPU restatement", cpu, n_ep), summary("HIP engine", hip, n_ep)
    print()
    print("| engine | seeds | last-10 mean over seeds: mean +- sd (s.e.) | median | seeds ending < 2500 | seeds ending < 1000 | best epoch: mean +- sd |")
    print("|---|---|---|---|---|---|---|")
    for s in (sc, sh):
        print(f"| {s['name']} | {s['n']} | {s['mean']:.0f} +- {s['sd']:.0f} ({s['se']:.0f}) | {s['median']:.0f} | {s['low']} | {s['dead']} | "
              f"{s['best'].mean():.0f} +- {s['best'].std(ddof=1):.0f} |")
    print("| reference README.md:146 (MuJoCo Hopper-v2, SAC) | | 3403 +- 446 | | | | |")
    from scipy import stats
    t, p = stats.ttest_ind(sc["fin"], sh["fin"], equal_var=False)
    u, pu = stats.mannwhitneyu(sc["fin"], sh["fin"], alternative="two-sided")
    d = sh["mean"] - sc["mean"]
    se = np.sqrt(sc["se"] ** 2 + sh["se"] ** 2)
    print()
    print(f"HIP - CPU = {d:+.0f} (s.e. of the difference {se:.0f}); Welch t = {t:.2f}, p = {p:.2f}; Mann-Whitney U = {u:.0f}, p = {pu:.2f}; "
          f"seeds ending below 2500: CPU {sc['low']}/{sc['n']}, HIP {sh['low']}/{sh['n']} "
          f"(Fisher exact p = {stats.fisher_exact([[sc['low'], sc['n'] - sc['low']], [sh['low'], sh['n'] - sh['low']]])[1]:.2f}).")
    # learning curves side by side: mean over seeds every 10 epochs
    print()
    print("| epoch | " + " | ".join(str(e) for e in range(9, n_ep, 10)) + " |")
    print("|---|" + "---|" * len(range(9, n_ep, 10)))
    for name, runs in (("CPU mean over seeds", cpu), ("HIP mean over seeds", hip), ("CPU median", cpu), ("HIP median", hip)):
        f = np.median if "median" in name else np.mean
        print(f"| {name} | " + " | ".join(f"{f([v['r'][e] for v in runs.values()]):.0f}" for e in range(9, n_ep, 10)) + " |")
    old = glob.glob(os.path.join(args.hip_old, "sac_hopper_refschedule*seed*.csv")) + glob.glob(os.path.join(args.hip_old, "*refschedule*.csv"))
    if old:
        print()
        print(f"(round 4's HIP runs on the same schedule with per-step insert and 16 eval envs: {args.hip_old}/, profiles/r04_returns.md)")


if __name__ == "__main__":
    main()
