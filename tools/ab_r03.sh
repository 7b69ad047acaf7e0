mkdir -p gpurun_out
F="--no-aux --no-split-run --no-cpu-baseline --no-seeds --steps 10 --warmup 3"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], "value %.0f ms/step %.2f" % (d["value"], d["ms_per_step"]), [(k["kernel"][:22], round(k["avg_launch_us"],2)) for k in d["roofline"]["kernels"]])
PY
}
(cd _r03 && timeout 200 python bench.py $F > ../gpurun_out/ab_r03_a.json 2>/dev/null); show gpurun_out/ab_r03_a.json
timeout 200 python bench.py $F > gpurun_out/ab_head_a.json 2>/dev/null; show gpurun_out/ab_head_a.json
(cd _r03 && timeout 200 python bench.py $F > ../gpurun_out/ab_r03_b.json 2>/dev/null); show gpurun_out/ab_r03_b.json
timeout 200 python bench.py $F > gpurun_out/ab_head_b.json 2>/dev/null; show gpurun_out/ab_head_b.json
ILSX_NO_PHASE=1 timeout 200 python bench.py $F > gpurun_out/ab_head_nophase.json 2>/dev/null; show gpurun_out/ab_head_nophase.json
(cd _r03 && ILSX_NO_PHASE=1 timeout 200 python bench.py $F > ../gpurun_out/ab_r03_nophase.json 2>/dev/null); show gpurun_out/ab_r03_nophase.json
ILSX_GANTT_NO_BUILD=1 timeout 200 python tools/phase_gantt.py > gpurun_out/phase_gantt_head.txt 2>&1; tail -40 gpurun_out/phase_gantt_head.txt
