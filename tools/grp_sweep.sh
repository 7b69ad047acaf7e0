#!/bin/bash
# The macro-tile knobs of the grouped SAC lock-step, one process per setting -> gpurun_out/grp_sweep.jsonl
# (ILSX_GRP_MT = row tiles per workgroup; libilsx_w1.so = the same library with the macro-tile kernels compiled for one wave per SIMD:
#  make -C ilswiss_amd/csrc VAR=w1 VARFLAGS=-DILSX_MT_WAVES=1 — skipped when it has not been built)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/grp_sweep.jsonl
: > $out
for lib in ilswiss_amd/libilsx.so ilswiss_amd/libilsx_w1.so; do
  [ -f $lib ] || continue
  for mt in 1 2 4; do
    if [ $mt = 1 ] && [ $lib != ilswiss_amd/libilsx.so ]; then continue; fi
    for cfg in "hopper 8" "hopper 4" "humanoid 4" "walker 8"; do
      ILSX_LIB=$lib ILSX_GRP_MT=$mt timeout 120 python tools/grp_sweep.py $cfg 1500 >> $out 2>> gpurun_out/grp_sweep.err || echo "{\"failed\": \"$lib $mt $cfg\"}" >> $out
    done
  done
done
# the weight gradients as 8-wave tiles instead of one-wavefront strips (ILSX_DW_GRP_STRIP=0), at the default macro tile
for cfg in "hopper 8" "humanoid 4"; do
  ILSX_DW_GRP_STRIP=0 timeout 120 python tools/grp_sweep.py $cfg 1500 | sed 's/"mt": "default"/"mt": "default, dW tiles"/' >> $out 2>> gpurun_out/grp_sweep.err
done
python - <<'PY'
import json
for l in open("gpurun_out/grp_sweep.jsonl"):
    d = json.loads(l)
    if "failed" in d: print(d); continue
    ks = {k.split("<")[0] + ("<" + k.split("<")[1][:22] if "<" in k else ""): round(v["avg_us"], 1) for k, v in d["kernels"].items()}
    print(d["lib"], "mt", d["mt"], d["task"], "K", d["K"], "us/lockstep %.1f" % d["us_per_lockstep"], "agg %.0f" % d["aggregate_grad_steps_per_s"], d["finite"], ks)
PY
