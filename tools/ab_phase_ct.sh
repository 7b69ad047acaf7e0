# A/B of the phase kernels' descriptor blocks: kernel arguments (ILSX_PHASE_CT=0) against the constant-memory copy (default), interleaved
for rep in 1 2 3; do for v in 0 1; do
  ILSX_PHASE_CT=$v timeout 300 python bench.py --no-aux --no-seeds --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k={x['kernel'].split('<')[0]: round(x['avg_launch_us'],2) for x in d['roofline']['kernels']}
print('ILSX_PHASE_CT=$v', 'grad-steps/s %.0f'%d['value'], 'us/step %.2f'%(1e3*d['ms_per_step']/1000), d['roofline']['kernel'][:16], k, d['phase_kernels']['fallbacks'])"
done; done
