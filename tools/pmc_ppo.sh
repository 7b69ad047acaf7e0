#!/bin/bash
# Memory-side traffic per launch of the PPO update's kernels at 32768-row minibatches (FETCH_SIZE / WRITE_SIZE, one counter per pass;
# the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE x 2 for wide coalesced reads).  Launches are told apart by grid size: the
# 1,048,576-row forwards of calc_adv are left out.   bash tools/pmc_ppo.sh <tag>  ->  gpurun_out/ppopmc_<tag>/summary.txt
set -u
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/ppopmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  ILSX_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$pass" -- python "$ROOT/tools/pmc_ppo_workload.py" > "$OUT/$pass.log" 2>&1
  echo "pass $pass rc=$?" >> "$OUT/passes.txt"
done
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "").replace("void ", "").split("(")[0].strip()
        grid = int(row.get("Grid_Size", 0) or 0)
        if any(s in k for s in ("k_mlp_fwd", "k_mlp_bwd_dx", "k_dw_big", "k_dw_reduce", "k_mlp_bwd_dw")):
            acc[(k, grid)][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# PPO update, 32768-row minibatches: memory-side bytes per launch (FETCH_SIZE KiB x 1024 x 2 [gfx950 correction], WRITE_SIZE KiB x 1024); launches by grid size")
summary = collections.defaultdict(list)
for (k, grid), cs in sorted(acc.items()):
    f = cs.get("FETCH_SIZE", []); w = cs.get("WRITE_SIZE", [])
    fm = 2048.0 * sum(f) / max(1, len(f)); wm = 1024.0 * sum(w) / max(1, len(w))
    print(f"{k[:60]:60s} grid {grid:9d} launches {max(len(f), len(w)):4d}  read {fm / 1e6:8.1f} MB  written {wm / 1e6:8.1f} MB")
    summary[k.split("<")[0]].append(dict(grid=grid, launches=max(len(f), len(w)), read_bytes=fm, written_bytes=wm))
# summary.json: what bench_aux.py reads for ppo_8192x128.roofline.traffic (copy to profiles/rNN_ppopmc_summary.json); _meta names the sources it describes
import hashlib, json
repo = os.path.dirname(os.path.dirname(root))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(repo, "ilswiss_amd", "csrc", "*.hip")) + glob.glob(os.path.join(repo, "ilswiss_amd", "csrc", "*.h"))
                + glob.glob(os.path.join(repo, "ilswiss_amd", "csrc", "*.inc")) + [os.path.join(repo, "include", "ilsx.h")]):
    h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
summary["_meta"] = dict(csrc_sha256=h.hexdigest())
json.dump(summary, open(os.path.join(root, "summary.json"), "w"), indent=1, sort_keys=True)
PY
cat "$OUT/passes.txt"
