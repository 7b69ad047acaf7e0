#!/bin/bash
# Config 3 at the reference's full schedule on the GPU box: GAIL Walker2d (gail_walker_hip.yaml) for `epochs` epochs (gail_walker.yaml: 562)
# under a wall-clock cap; progress.csv is written every epoch, so a capped run still leaves its curve.   bash tools/gail_full_run.sh [epochs] [cap_s]
set -u
GE=${1:-562}
CAP=${2:-1500}
mkdir -p gpurun_out/r03_gail_full
python - <<PY
import pickle, numpy as np
d = np.load("demos/walker_sac_expert_policy.npz")
pickle.dump(dict(policy=d["policy"]), open("/tmp/walker_expert.pkl", "wb"))
PY
python run_scripts/gen_expert_demos.py --snapshot /tmp/walker_expert.pkl --env walker --num-trajs 16 --out demos/walker_sac.pkl > gpurun_out/r03_gail_full/demos.log 2>&1
sed "s/num_epochs: 100/num_epochs: $GE/" exp_specs/gail/gail_walker_hip.yaml > /tmp/gail_walker_run.yaml
( time timeout $CAP python run_experiment.py -e /tmp/gail_walker_run.yaml -g 0 ) > gpurun_out/r03_gail_full/run.log 2>&1
for d in logs/gail-walker-hip/*/; do cp "$d/progress.csv" gpurun_out/r03_gail_full/gail_walker_progress.csv; done
tail -4 gpurun_out/r03_gail_full/run.log
python - <<'PY'
import csv
import numpy as np
rows = list(csv.DictReader(open("gpurun_out/r03_gail_full/gail_walker_progress.csv")))
r = [float(x["Test Returns Mean"]) for x in rows]
print("GAIL Walker:", len(rows), "epochs; best", round(max(r), 1), "at", int(np.argmax(r)), "; last-10 mean", round(float(np.mean(r[-10:])), 1),
      "; last-50 mean", round(float(np.mean(r[-50:])), 1), "; gradient steps", rows[-1]["Number of gradient steps total"], "; wall", rows[-1].get("Total Train Time (s)"))
for i in range(0, len(r), 25):
    print(i, round(float(np.mean(r[i:i + 25])), 1))
PY
