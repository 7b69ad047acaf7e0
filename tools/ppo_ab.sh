#!/bin/bash
# PPO 8192 x 128 update (bench_aux.py ppo) with the env knobs given as arguments, one process each -> gpurun_out/ppo_ab.txt
# usage: tools/ppo_ab.sh "ILSX_DW_BIG=0" "ILSX_DW_BIG=1" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/ppo_ab.txt
for kv in "$@"; do
  env $kv python bench_aux.py ppo 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$kv', 'value %.1fM' % (d['value']/1e6), 'update_s %.4f' % d['detail']['mb32768']['train_step_s'], r['kernel'], 'frac %.3f' % r['frac'], {k: round(v,1) for k,v in r['kernel_ms'].items()})
" >> gpurun_out/ppo_ab.txt
done
cat gpurun_out/ppo_ab.txt
