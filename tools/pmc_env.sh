#!/bin/bash
# Dynamic instruction mix and wait fractions of the planar stepper (k_env_step) — rocprofv3 counter passes over tools/rollout_overhead.py.
#   bash tools/pmc_env.sh <tag>    (on the GPU box, from the repo root)
set -u
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/envpmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  name=$(echo "$pass" | tr ' ' '+' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$name" -- python "$ROOT/tools/rollout_overhead.py" 4096 > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items()):
    if "k_env" not in k:
        continue
    w = v.get("SQ_WAVES", 0) or 1
    print(k, "waves/launch", round(w, 1), {c: round(x / w, 1) for c, x in v.items() if c.startswith("SQ_")})
PY
