"""CPU suite: `python bench.py --gpus 2` must start two ranks by itself (no launcher), rendezvous on 127.0.0.1, barrier,
take the max over ranks and print ONE JSON line from rank 0 — the plumbing the driver's N = 1, 2, 4, 8 scaling runs go through.
`--dry-run` swaps the GPU work for a sleep and RCCL for gloo; spawn / process group / barrier / max-reduce are the same code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_spawns_two_ranks():
    d = _run(["--gpus", "2", "--steps", "20", "--warmup", "1", "--dry-run"])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["dry_run"] is True
    assert d["ms_per_step"] >= 2.0   # rank 1 sleeps 2 ms per step: the reported time is the slowest rank's


def test_bench_under_a_launcher_uses_its_ranks():
    """the driver's form: torch.distributed.run provides RANK / WORLD_SIZE; bench.py must not spawn again"""
    e = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_single_rank_dry_run():
    d = _run(["--steps", "3", "--dry-run"])
    assert d["n_gpus"] == 1


def test_a_leg_that_fails_on_one_rank_does_not_hang_the_others():
    """Ranks.leg: rank 1 raises before the leg's barrier; rank 0 must come through both of the leg's collectives and print the line"""
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], env={"ILSX_BENCH_DRY_FAIL_RANK": "1"})
    assert d["leg_errors"] == [0.0, 1.0] and abs(d["leg"]["dt_max"] - 0.001) < 1e-9
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], env={"ILSX_BENCH_DRY_FAIL_RANK": "0"})
    assert d["leg_errors"] == [1.0, 0.0] and "error" in d["leg"]
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert d["leg_errors"] == [0.0, 0.0] and abs(d["leg"]["dt_max"] - 0.002) < 1e-9
