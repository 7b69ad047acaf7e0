#!/bin/bash
# Dynamic instruction mix and wait fractions of the 3-D stepper (k_env3dw_step<23>: one wavefront per Humanoid env) — rocprofv3 counter
# passes over tools/env3d_rate.py ("fp64 issue-bound" was asserted with counters for the planar kernel only).
#   bash tools/pmc_env3d.sh <tag>    (on the GPU box, from the repo root)  ->  gpurun_out/env3dpmc_<tag>/summary.json
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/env3dpmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  name=$(echo "$pass" | tr ' ' '+' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$name" -- python "$ROOT/tools/env3d_rate.py" humanoid 1024 12 > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items()):
    if "k_env3d" not in k:
        continue
    w = v.get("SQ_WAVES", 0) or 1
    print(k, "waves/launch", round(w, 1), {c: round(x / w, 1) for c, x in v.items() if c.startswith("SQ_")})
PY
