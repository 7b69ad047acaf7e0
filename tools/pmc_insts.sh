#!/bin/bash
# Dynamic instruction mix of the SAC step's kernels (one rocprofv3 counter pass over tools/tail_trace.py, ILSX_NO_GRAPH=1):
# instructions per wave by class.  With ONE wave per SIMD every VALU instruction costs 4 cycles whether or not it depends on
# the previous one (tools/ubench/icache.hip), so instructions per wave x 1.7 ns is a floor of the workgroup lifetime.
#   bash tools/pmc_insts.sh <tag>    (on the GPU box, from the repo root)
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/insts_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_FLAT"; do
  name=$(echo "$pass" | tr ' ' '+' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$name" -- python "$ROOT/tools/tail_trace.py" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items()):
    if "k_mlp" not in k and "k_sac" not in k:
        continue
    w = v.get("SQ_WAVES", 0) or 1
    print(k, "waves/launch", round(w, 1), {c: round(x / w, 1) for c, x in v.items() if c.startswith("SQ_INSTS")})
PY
