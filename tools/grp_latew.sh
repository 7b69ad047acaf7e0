#!/bin/bash
# Same-box A/B of the grouped lock-step's "late weights" launch shape (kernels.h GRP == 3: the layer-1 weight slice fetched right before
# its MFMA phase, 4 waves per SIMD): ILSX_GRP_LATE = 0 (off) / 1 (on) / unset (each launch decides by its size) -> gpurun_out/grp_latew.jsonl
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/grp_latew.jsonl
: > $out
for rep in 1 2; do
  for cfg in "hopper 4" "hopper 6" "hopper 8" "hopper 12" "hopper 16" "walker 8" "humanoid 4" "humanoid 8"; do
    for late in 0 1 auto; do
      if [ $late = auto ]; then unset ILSX_GRP_LATE; else export ILSX_GRP_LATE=$late; fi
      timeout 120 python tools/grp_sweep.py $cfg 1500 | sed "s/^{/{\"late\": \"$late\", /" >> $out 2>> gpurun_out/grp_latew.err || echo "{\"failed\": \"$late $cfg\"}" >> $out
    done
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/grp_latew.jsonl"):
    d = json.loads(l)
    if "failed" in d: print(d); continue
    ks = {k.split("<")[0][6:] + ("<" + k.split("<")[1][-12:] if "<" in k else ""): round(v["avg_us"], 1) for k, v in d["kernels"].items()}
    print("late", d["late"], d["task"], "K", d["K"], "us/lockstep %.1f" % d["us_per_lockstep"], "agg %.0f" % d["aggregate_grad_steps_per_s"], d["finite"], ks)
PY
