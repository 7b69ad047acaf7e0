#!/bin/bash
# VERDICT r3 item 6 on the GPU box: returns on the reference's own schedule next to the 4096-env schedule, several seeds per task, ALL runs
# side by side on one GPU (a single run is latency-bound and leaves most of the chip idle; the launcher's children leave the phase kernels).
#   bash tools/returns_run.sh [cap_seconds] [tasks...]      tasks: hopper_ref hopper walker halfcheetah ant humanoid
# -> gpurun_out/r04_returns/<exp>__<run dir>.csv (progress files, also of runs the cap cut short) + table.md
set -u
CAP=${1:-1300}; shift || true
TASKS=${@:-hopper_ref walker halfcheetah ant humanoid}
OUT=gpurun_out/r04_returns
mkdir -p $OUT
rm -rf logs
seeds3() { sed -e 's/seed: \[0\]/seed: [0, 1, 2]/' -e 's/num_workers: 1/num_workers: 3/' "$1" > "$2"; }
pids=()
for t in $TASKS; do
  case $t in
    hopper_ref) spec=exp_specs/sac/sac_hopper_refschedule_hip.yaml ;;
    hopper) spec=exp_specs/sac/sac_hopper_hip_5seeds.yaml ;;
    walker|halfcheetah|ant) seeds3 exp_specs/sac/sac_${t}_hip.yaml /tmp/ret_$t.yaml; spec=/tmp/ret_$t.yaml ;;
    humanoid) sed -e 's/seed: \[0\]/seed: [0, 1]/' -e 's/num_workers: 1/num_workers: 2/' exp_specs/sac/sac_humanoid_hip.yaml > /tmp/ret_$t.yaml; spec=/tmp/ret_$t.yaml ;;
  esac
  ( timeout $CAP python run_experiment.py -e $spec -g 0 > $OUT/run_$t.log 2>&1 ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for f in $(find logs -name progress.csv); do
  d=$(dirname $f); e=$(basename $(dirname $d))
  cp $f $OUT/${e}__$(basename $d | tr ' ' '_').csv
done
python - <<'PY' | tee gpurun_out/r04_returns/table.md
import csv, glob, os, collections
import numpy as np
runs = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r04_returns/*__*.csv")):
    rows = list(csv.DictReader(open(f)))
    if not rows: continue
    r = [float(x["Test Returns Mean"]) for x in rows]
    runs[os.path.basename(f).split("__")[0]].append((len(rows), float(np.mean(r[-10:])), max(r), rows[-1].get("Number of env steps total", "?"), rows[-1].get("Number of train steps total", "?")))
print("| experiment (own simulator) | seeds | epochs done | env steps | grad steps | mean of last-10-epoch AverageReturn: mean +- std over seeds | best epoch: mean +- std |")
print("|---|---|---|---|---|---|---|")
for e, v in runs.items():
    fin, best = [x[1] for x in v], [x[2] for x in v]
    print(f"| {e} | {len(v)} | {'/'.join(str(x[0]) for x in v)} | {v[0][3]} | {v[0][4]} | {np.mean(fin):.0f} +- {np.std(fin):.0f} | {np.mean(best):.0f} +- {np.std(best):.0f} |")
PY
