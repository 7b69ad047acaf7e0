# A/B of the grouped weight-gradient launch (K = 8 Hopper, K = 4 Humanoid): table pointers as global accesses (default) vs FLAT (libilsx_flat.so:
# make -C ilswiss_amd/csrc VAR=flat VARFLAGS=-DILSX_FLAT_TABLE_PTRS), each with the 104-register tile (ILSX_DW_GRP_LOW=0) and the 63-register one (=1)
for rep in 1 2; do
for lib in ilswiss_amd/libilsx_flat.so ilswiss_amd/libilsx.so; do
[ -f $lib ] || continue
for kv in "ILSX_DW_GRP_LOW=0" "ILSX_DW_GRP_LOW=1" "ILSX_DW_TILE_GRP=12"; do
  for cfg in "hopper 8" "humanoid 4"; do
    env ILSX_LIB=$lib $kv timeout 120 python tools/grp_sweep.py $cfg 1500 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k.split('<')[0]+('<'+k.split('<')[1][:14] if '<' in k else ''): round(v['avg_us'],1) for k,v in d['kernels'].items()}
print('$lib', '$kv', '$cfg', 'us/lockstep %.1f'%d['us_per_lockstep'], 'agg %.0f'%d['aggregate_grad_steps_per_s'], ks)"
  done
done
done
done
