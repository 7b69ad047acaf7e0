# End-of-round evidence on the GPU box (from the repo root): bench line, the other configs, rocprofv3 kernel statistics of the
# same bench command (ILSX_NO_GRAPH=1: rocprofv3 does not see graph launches here) -> gpurun_out/prof_e/; copy into profiles/.
set -u
ROOT=$(pwd)
mkdir -p gpurun_out/prof_e
timeout 300 python bench.py > gpurun_out/prof_e/bench.json 2> gpurun_out/prof_e/bench.err
timeout 600 python bench_aux.py ppo gail td3 seeds > gpurun_out/prof_e/bench_aux.log 2>&1
cd /tmp && export TMPDIR=/tmp
ILSX_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_e/rocprof -o r01e -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_e/rocprof.log 2>&1
cd $ROOT
f=$(find gpurun_out/prof_e/rocprof -name "*.db" | head -1)
python tools/rocpd_summary.py "$f" > gpurun_out/prof_e/kernel_stats.csv 2> gpurun_out/prof_e/summary.err
head -c 600 gpurun_out/prof_e/bench.json; echo; head -20 gpurun_out/prof_e/kernel_stats.csv
