#!/bin/bash
# BatchNorm discriminator on the GPU box: its tests, its rate without a profiler, rocprofv3 kernel statistics of the same script.
#   bash tools/r05_bnprof.sh [full]   (full: the whole GPU suite instead of the discriminator tests)  ->  gpurun_out/r05_bnprof/
set -u
OUT=gpurun_out/r05_bnprof; mkdir -p $OUT
if [ "${1:-}" = full ]; then python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; else python -m pytest tests/test_disc.py -m gpu -q > $OUT/gpu_tests.log 2>&1; fi
tail -4 $OUT/gpu_tests.log
timeout 120 python tools/discbn_rate.py > $OUT/discbn_rate.txt 2>&1; cat $OUT/discbn_rate.txt
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/rocprof -o bn -- python $ROOT/tools/discbn_rate.py > $ROOT/$OUT/rocprof.log 2>&1
cd $ROOT
f=$(find $OUT/rocprof -name "*.db" | head -1)
python tools/rocpd_summary.py "$f" > $OUT/bn_kernel_stats.csv 2> $OUT/summary.err
rm -rf $OUT/rocprof
head -40 $OUT/bn_kernel_stats.csv | cut -c1-200
