#!/bin/bash
# Dynamic instruction mix per wave of the grouped (K co-resident seeds) SAC lock-step kernels — rocprofv3 counter passes over
# tools/grp_sweep.py.  The fp32 MFMA of this part does not overlap with the VALU / LDS issue of the same SIMD (profiles/
# r04_mfma_overlap.txt), so instructions per wave, not occupancy, are what the launch time is made of.
#   bash tools/pmc_grp.sh <tag> [task] [K]   (on the GPU box, from the repo root)  ->  gpurun_out/grppmc_<tag>/summary.json
set -u
TAG=${1:-r05}
TASK=${2:-hopper}
K=${3:-8}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/grppmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  name=$(echo "$pass" | tr ' ' '+' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$name" -- python "$ROOT/tools/grp_sweep.py" $TASK $K 60 > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items()):
    if not any(s in k for s in ("k_mlp2", "k_mlp_bwd_dw", "k_sac_tail")):
        continue
    w = v.get("SQ_WAVES", 0) or 1
    print(k[:70], "waves/launch", round(w, 1), {c: round(x / w, 1) for c, x in v.items() if c.startswith("SQ_") and c != "SQ_WAVES"})
PY
