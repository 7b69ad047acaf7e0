#!/bin/bash
# Round-5 mid-round check on the GPU box (from the repo root): GPU suite, BatchNorm discriminator rate, the split-run step at one rank in its
# three forms, a short headline bench.   bash tools/r05_check.sh  ->  gpurun_out/r05_check/
set -u
OUT=gpurun_out/r05_check
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -5 $OUT/gpu_tests.log
timeout 120 python tools/discbn_rate.py > $OUT/discbn_rate.txt 2>&1; cat $OUT/discbn_rate.txt
split() {  # $1 = label, rest = env
  label=$1; shift
  env "$@" ILSX_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-aux --no-cpu-baseline --no-seeds --steps 5 --warmup 2 2> $OUT/split_$label.err | grep '^{' | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(form='$label', headline=d.get('value'), **(d.get('split_run') or {}))))"
}
: > $OUT/split_run_1rank.jsonl
split phase_kernels ILSX_SPLIT_SEGMENTS=0 >> $OUT/split_run_1rank.jsonl
split one_launch_per_stage ILSX_SPLIT_NO_PHASE=1 >> $OUT/split_run_1rank.jsonl
split whole_step_graph ILSX_SPLIT_GRAPH=1 >> $OUT/split_run_1rank.jsonl
cut -c1-420 $OUT/split_run_1rank.jsonl
timeout 300 python bench.py --no-aux --no-cpu-baseline --no-seeds > $OUT/bench_short.json 2> $OUT/bench_short.err; cut -c1-600 $OUT/bench_short.json
