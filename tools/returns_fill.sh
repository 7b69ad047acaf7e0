#!/bin/bash
# Re-run the seeds that the 16-way shared call of tools/returns_run.sh cut short, side by side on one GPU:
#   [SPEC=exp_specs/sac/sac_walker_hip.yaml] [OUT=gpurun_out/<dir>] bash tools/returns_fill.sh <cap_seconds> <seed> [<seed> ...]   ->  gpurun_out/r04_returns_fill/seed<k>.csv
# (default SPEC: the reference-schedule SAC Hopper spec)
set -u
CAP=${1:-1400}; shift
SEEDS=$(echo "$@" | tr ' ' ',')
N=$#
OUT=${OUT:-gpurun_out/r04_returns_fill}
mkdir -p $OUT
rm -rf logs $OUT/seed*.csv
SPEC=${SPEC:-exp_specs/sac/sac_hopper_refschedule_hip.yaml}
sed -e "s/seed: \[[0-9, ]*\]/seed: [$SEEDS]/" -e "s/num_workers: [0-9]*/num_workers: $N/" $SPEC > /tmp/ret_fill.yaml
timeout $CAP python run_experiment.py -e /tmp/ret_fill.yaml -g 0 > $OUT/run.log 2>&1
for f in $(find logs -name progress.csv); do
  d=$(dirname $f)
  s=$(echo $d | sed -n 's/.*--s-\([0-9]*\).*/\1/p')
  cp $f $OUT/seed$s.csv
  # KEEP_SNAPSHOTS=1: the run's last periodic snapshot (policy, critics, optimiser state) too — tools/eval_snapshot_cpu.py plays its policy in the CPU stepper
  if [ "${KEEP_SNAPSHOTS:-0}" = 1 ] && [ -f $d/params.pkl ]; then cp $d/params.pkl $OUT/seed$s.params.pkl; fi
done
OUT=$OUT python - <<'PY'
import csv, glob, os
import numpy as np
for f in sorted(glob.glob(os.environ["OUT"] + "/seed*.csv")):
    rows = list(csv.DictReader(open(f)))
    r = [float(x["Test Returns Mean"]) for x in rows]
    print(f, "epochs", len(rows), "env steps", rows[-1].get("Number of env steps total"), "last-10 %.0f" % np.mean(r[-10:]), "best %.0f" % max(r))
PY
