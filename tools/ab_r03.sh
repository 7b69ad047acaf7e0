#!/bin/bash
# Same-box A/B of the single-run SAC step: the round-3 tree (a copy under _r03/, built there; not tracked) against HEAD and HEAD's
# variant builds, interleaved (box-to-box differences are larger than what is being measured: 12.2k vs 14.8k grad-steps/s on two boxes).
mkdir -p gpurun_out
F="--no-aux --no-split-run --no-cpu-baseline --no-seeds --steps 10 --warmup 3"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], "value %.0f ms/step %.2f" % (d["value"], d["ms_per_step"]), [(k["kernel"][:22], round(k["avg_launch_us"],2)) for k in d["roofline"]["kernels"]])
PY
}
for rep in a b; do
  if [ -d _r03 ]; then (cd _r03 && timeout 200 python bench.py $F > ../gpurun_out/ab_r03_$rep.json 2>/dev/null); show gpurun_out/ab_r03_$rep.json; fi
  timeout 200 python bench.py $F > gpurun_out/ab_head_$rep.json 2>/dev/null; show gpurun_out/ab_head_$rep.json
done
