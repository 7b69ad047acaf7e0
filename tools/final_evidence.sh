#!/bin/bash
# End of round on the GPU box (from the repo root):  bash tools/final_evidence.sh <tag>
# profile_round (bench, kernel table, counters, gantts, stepper stage clocks, A/Bs), the grouped and PPO counter passes, then the counter
# summaries put where bench.py looks for them and the FINAL bench line (roofline.traffic from this very tree: traffic_stale false), the ten
# grouped refloop seeds with eval_async (acceptance: identical to profiles/r06_returns_async), the K sweep of the grouped weight-gradient tile.
set -u
TAG=${1:-r06c}
RN=${TAG:0:3}
bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
bash tools/pmc_grp.sh $TAG hopper 8 > gpurun_out/pmc_grp_$TAG.log 2>&1
bash tools/pmc_ppo.sh $TAG > gpurun_out/pmc_ppo_$TAG.log 2>&1
cp gpurun_out/pmc_$TAG/summary.json profiles/${RN}_pmc_summary.json 2> /dev/null
cp gpurun_out/grppmc_$TAG/summary.json profiles/${RN}_grppmc_summary.json 2> /dev/null
[ -f gpurun_out/ppopmc_$TAG/summary.json ] && cp gpurun_out/ppopmc_$TAG/summary.json profiles/${RN}_ppopmc_summary.json
rm -rf gpurun_out/grppmc_$TAG/*/ gpurun_out/ppopmc_$TAG/*/ 2> /dev/null
timeout 600 python bench.py > gpurun_out/${RN}_bench_final.json 2> gpurun_out/${RN}_bench_final.err
mkdir -p gpurun_out/${TAG}_returns_async
timeout 700 python tools/grouped_entry_rate.py exp_specs/sac/sac_hopper_refloop_hip.yaml --group 10 --epochs 102 --set rl_alg_params.eval_async=true --keep gpurun_out/${TAG}_returns_async > gpurun_out/${TAG}_returns_async/launcher_0_9_async.json 2> gpurun_out/${TAG}_returns_async/err.log
: > gpurun_out/${TAG}_dw_low_ksweep.txt
for K in 2 4 16; do for kv in ILSX_DW_GRP_LOW=0 ILSX_DW_GRP_LOW=1; do
  env $kv timeout 120 python tools/grp_sweep.py hopper $K 1000 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k.split('<')[0]+('<'+k.split('<')[1][:14] if '<' in k else ''): round(v['avg_us'],1) for k,v in d['kernels'].items()}
print('$kv', 'hopper K=$K', 'us/lockstep %.1f'%d['us_per_lockstep'], 'agg %.0f'%d['aggregate_grad_steps_per_s'], ks)" >> gpurun_out/${TAG}_dw_low_ksweep.txt
done; done
head -c 500 gpurun_out/${RN}_bench_final.json; echo; cat gpurun_out/${TAG}_dw_low_ksweep.txt; head -c 400 gpurun_out/${TAG}_returns_async/launcher_0_9_async.json
