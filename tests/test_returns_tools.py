"""The tools behind profiles/r05_returns.md: tools/returns_cpu.py (the CPU restatement end to end on the reference's loop) takes a tiny
schedule through every branch of the loop — episode ends, the train trigger, evaluation — and tools/returns_compare.py turns two directories
of progress files into the comparison table."""
import csv
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_restatement_runs_the_reference_loop(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import returns_cpu
    out = str(tmp_path / "seed0.csv")
    # 2 epochs x 1200 env steps, a train call (200 gradient steps here) per 400 env steps, evaluation of >= 300 steps
    returns_cpu.run(0, out, epochs=1, steps_per_epoch=1200, between=400, per_call=200, batch=64, eval_steps=300, quiet=True)
    rows = list(csv.DictReader(open(out)))
    assert [int(r["Epoch"]) for r in rows] == [0, 1]
    assert [int(r["Number of env steps total"]) for r in rows] == [1200, 2400]
    assert [int(r["Number of gradient steps total"]) for r in rows] == [600, 1200]      # one call per 400 env steps
    last = rows[-1]
    assert int(last["Test Num Paths"]) % 4 == 0 and int(last["Test Num Paths"]) >= 4      # whole 4-env rollouts (vec_sampler.py:126-146)
    assert float(last["Test Path Length Mean"]) * int(last["Test Num Paths"]) >= 300
    # samples enter the ring only when their episode ends (base_algorithm.py:509-519): fewer rows than env steps while paths are open
    assert 0 < int(last["Replay size"]) <= 2400
    assert int(last["Exploration Num Paths"]) > 0 and np.isfinite(float(last["QF1 Loss"])) and float(last["Alpha"]) > 0


def test_compare_table(tmp_path):
    rng = np.random.default_rng(0)
    for eng, mu in (("cpu", 3000.0), ("hip", 2950.0)):
        d = tmp_path / eng
        d.mkdir()
        for s in range(4):
            with open(d / f"seed{s}.csv", "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Epoch", "Number of env steps total", "Test Returns Mean"])
                for e in range(25):
                    w.writerow([e, 10000 * (e + 1), min(mu, 150.0 * e) + rng.normal(0, 30)])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "returns_compare.py"), "--cpu", str(tmp_path / "cpu"), "--hip", str(tmp_path / "hip"),
                        "--hip-old", str(tmp_path / "none")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "| CPU restatement | 4 |" in r.stdout and "| HIP engine | 4 |" in r.stdout and "Welch t" in r.stdout and "epoch 25 of 25" in r.stdout


def test_grouped_acceptance_runs_reproduce_round_5_cell_for_cell():
    """profiles/r06_returns_grouped (ten / fourteen seeds of the reference schedule in ONE process, lock-step, through run_experiment.py --group)
    against profiles/r05_returns_hip (the same seeds as single-run processes, round 5): every rollout column of every epoch is identical."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import returns_compare
    seeds, bad = returns_compare.same(os.path.join(ROOT, "profiles", "r06_returns_grouped"), os.path.join(ROOT, "profiles", "r05_returns_hip"))
    assert len(seeds) >= 10 and bad == 0, (seeds, bad)

def test_last_session_acceptance_run_repeats_the_earlier_one_cell_for_cell():
    """profiles/r06b_returns_async (ten grouped seeds of the reference schedule with eval_async, run AFTER the steppers' Gauss-Seidel loops, mass
    matrix, lane constants and the grouped weight-gradient tiles were rewritten — DESIGN §3i) against profiles/r06_returns_async (the same
    command on the tree before): every rollout column of every epoch both hold is identical — the rewrites did not move a bit of a Hopper run."""
    import csv
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import returns_compare as rc
    n_seeds = 0
    for run in ("r06b_returns_async", "r06c_returns_async", "r06f_returns_async"):   # mid-session trees ; the round's final tree (tools/final_evidence.sh)
        for s in range(10):
            a = list(csv.DictReader(open(os.path.join(ROOT, "profiles", run, f"seed{s}.csv"))))
            b = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_returns_async", f"seed{s}.csv"))))
            cols = [k for k in a[0] if k.startswith(rc.ROLLOUT_COLUMNS) or k in rc.ROLLOUT_COLUMNS]
            n = min(len(a), len(b))
            assert n >= 103 and len(cols) >= 10
            assert all(ra[k] == rb[k] for ra, rb in zip(a[:n], b[:n]) for k in cols), (run, s)
            n_seeds += 1
    assert n_seeds == 30
