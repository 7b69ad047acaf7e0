# End-of-round evidence on the GPU box (from the repo root):  bash tools/profile_round.sh [tag]   (default r02)
# bench line, the other configs, rocprofv3 kernel statistics of the same bench command (ILSX_NO_GRAPH=1: rocprofv3 does not see
# graph launches here), the PMC passes (regenerated every round: bench.py reads the newest summary for roofline.traffic), the
# per-launch timeline of the SAC step and the stage timers of the 3-D stepper -> gpurun_out/prof_<tag>/; copy into profiles/.
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python bench_aux.py td3 seeds > $OUT/bench_aux.log 2>&1
ILSX_NO_PHASE=1 timeout 200 python tools/step_gantt.py > $OUT/step_gantt_8launch.txt 2>&1
timeout 200 python tools/phase_gantt.py > $OUT/phase_gantt.txt 2>&1
timeout 200 python tools/rollout_overhead.py 4096 > $OUT/rollout_overhead.txt 2>&1
timeout 100 python tools/rollout_overhead.py 8192 >> $OUT/rollout_overhead.txt 2>&1
(cd tools/ubench && timeout 60 ./tilesync) > $OUT/tilesync.txt 2>&1
# the split-run leg on a one-rank communicator in its four forms: merged phase kernels (default since round 5), one launch per stage, graph
# segments between the two all-reduces, the whole step with the collective in one capture
: > $OUT/split_run_1rank.jsonl
for form in "phase_kernels:ILSX_SPLIT_SEGMENTS=0" "one_launch_per_stage:ILSX_SPLIT_NO_PHASE=1" "graph_segments:ILSX_SPLIT_SEGMENTS=1" "whole_step_graph:ILSX_SPLIT_GRAPH=1"; do
  label=${form%%:*}; kv=${form#*:}
  env $kv ILSX_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-aux --no-cpu-baseline --no-seeds --steps 5 --warmup 2 2> /dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(form='$label', **(d.get('split_run') or {}))))" >> $OUT/split_run_1rank.jsonl 2>&1
done
timeout 120 python tools/discbn_rate.py > $OUT/discbn_rate.txt 2>&1
timeout 200 bash tools/ppo_ab.sh ILSX_DW_BIG=1 ILSX_DW_BIG=0 > $OUT/ppo_ab.txt 2>&1
for u in mfma_peak mfma_valu_overlap; do [ -x tools/ubench/$u ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/$u tools/ubench/$u.hip 2> /dev/null; done
(timeout 60 tools/ubench/mfma_peak; timeout 60 tools/ubench/mfma_valu_overlap) > $OUT/mfma_ubench.txt 2>&1
[ -x tools/ubench/mfma_32x32 ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/ubench/mfma_32x32 tools/ubench/mfma_32x32.hip 2> /dev/null
timeout 60 tools/ubench/mfma_32x32 > $OUT/mfma_32x32.txt 2>&1
timeout 120 python tools/replay_rate.py > $OUT/replay_rate.txt 2>&1
timeout 200 python tools/grp_streams_ab.py abcd > $OUT/grp_streams_ab.txt 2>&1
timeout 300 python tools/grouped_entry_rate.py exp_specs/sac/sac_humanoid_hip.yaml --group 4 --epochs 4 > $OUT/grouped_entry_humanoid.json 2>&1
timeout 300 python tools/grouped_entry_rate.py exp_specs/sac/sac_hopper_refloop_hip.yaml --group 10 --epochs 3 > $OUT/grouped_entry_refloop_3ep.json 2>&1
timeout 60 python tools/fwd_rate.py 32768 > $OUT/fwd_rate.txt 2>&1
timeout 120 python tools/step_gantt.py 8 > $OUT/step_gantt_K8.txt 2>&1
(timeout 100 python tools/env3d_rate.py humanoid 1024 40; timeout 100 python tools/env3d_rate.py ant 1024 40) > $OUT/env3d_rate.txt 2>&1
timeout 200 python tools/env_rate.py > $OUT/env2d_rate.txt 2>&1
# stage clock of the planar stepper (measurement build: make -C ilswiss_amd/csrc VAR=egprof VARFLAGS=-DILSX_EG_PROFILE; skipped when it has not been built)
[ -f ilswiss_amd/libilsx_egprof.so ] && (for t in hopper walker halfcheetah; do ILSX_LIB=ilswiss_amd/libilsx_egprof.so timeout 100 python tools/env2d_phases.py $t 4096; done) > $OUT/env2d_phases.txt 2>&1
timeout 300 bash tools/ab_dw_low.sh > $OUT/dw_low_ab.txt 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC tools/ubench/env3d_phases.hip -o /tmp/libe3p.so 2> /dev/null
(python tools/ubench/env3d_phases.py humanoid 1024 8; python tools/ubench/env3d_phases.py ant 1024 8) > $OUT/env3d_phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
ILSX_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/rocprof -o $TAG -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split-run --no-aux > $ROOT/$OUT/rocprof.log 2>&1
cd $ROOT
f=$(find $OUT/rocprof -name "*.db" | head -1)
python tools/rocpd_summary.py "$f" > $OUT/kernel_stats.csv 2> $OUT/summary.err
rm -rf $OUT/rocprof
bash tools/pmc_collect.sh $TAG > $OUT/pmc.log 2>&1
bash tools/pmc_env.sh $TAG > $OUT/envpmc.log 2>&1
bash tools/pmc_env3d.sh $TAG > $OUT/env3dpmc.log 2>&1
cp gpurun_out/env3dpmc_$TAG/summary.json $OUT/env3dpmc_summary.json 2> /dev/null
rm -rf gpurun_out/env3dpmc_$TAG/*/ 2> /dev/null
cp gpurun_out/envpmc_$TAG/summary.json $OUT/envpmc_summary.json 2> /dev/null
rm -rf gpurun_out/envpmc_$TAG/*/ 2> /dev/null
cp gpurun_out/pmc_$TAG/summary.json $OUT/pmc_summary.json 2> /dev/null
cp gpurun_out/pmc_$TAG/passes.txt $OUT/pmc_passes.txt 2> /dev/null
rm -rf gpurun_out/pmc_$TAG/*/ 2> /dev/null
head -c 700 $OUT/bench.json; echo; head -12 $OUT/kernel_stats.csv; tail -3 $OUT/bench_aux.log | cut -c1-400
