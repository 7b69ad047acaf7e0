set -u
OUT=gpurun_out/r05_bnprof; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -6 $OUT/gpu_tests.log
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/rocprof -o bn -- python $ROOT/tools/discbn_rate.py > $ROOT/$OUT/rocprof.log 2>&1
cd $ROOT
f=$(find $OUT/rocprof -name "*.db" | head -1)
python tools/rocpd_summary.py "$f" > $OUT/bn_kernel_stats.csv 2> $OUT/summary.err
rm -rf $OUT/rocprof
head -60 $OUT/bn_kernel_stats.csv | cut -c1-260
