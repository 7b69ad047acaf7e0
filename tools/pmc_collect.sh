#!/bin/bash
# rocprofv3 counter passes for profiles/ (run on the GPU box from the repo root):  bash tools/pmc_collect.sh <tag>
# One counter group per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit together);
# --pmc is never combined with trace domains other than --kernel-trace.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  name=$(echo "$pass" | tr ' ' '+' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$name" -- python "$ROOT/tools/pmc_workload.py" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
cat "$OUT/passes.txt"
