#!/bin/bash
# VERDICT r2 item 7 on the GPU box: the 5-seed SAC Hopper run (exp_specs/sac/sac_hopper_hip_5seeds.yaml: 100 epochs = 4.1 M env steps,
# 1.0 M gradient steps per seed) and one GAIL Walker2d run (gail_walker_hip.yaml) side by side on ONE GPU, progress files collected
# under gpurun_out/.  ILSX_NO_PHASE=1: six processes share the GPU, the merged phase kernels want it to themselves.
#   bash tools/seeds_run.sh [gail_epochs]
set -u
GE=${1:-150}
export ILSX_NO_PHASE=1
mkdir -p gpurun_out/r03_seeds
python - <<PY
import pickle, numpy as np
d = np.load("demos/walker_sac_expert_policy.npz")
pickle.dump(dict(policy=d["policy"]), open("/tmp/walker_expert.pkl", "wb"))
PY
python run_scripts/gen_expert_demos.py --snapshot /tmp/walker_expert.pkl --env walker --num-trajs 16 --out demos/walker_sac.pkl > gpurun_out/r03_seeds/demos.log 2>&1
sed "s/num_epochs: 100/num_epochs: $GE/" exp_specs/gail/gail_walker_hip.yaml > /tmp/gail_walker_run.yaml
( time python run_experiment.py -e /tmp/gail_walker_run.yaml -g 0 ) > gpurun_out/r03_seeds/gail_run.log 2>&1 &
GP=$!
( time python run_experiment.py -e exp_specs/sac/sac_hopper_hip_5seeds.yaml -g 0 ) > gpurun_out/r03_seeds/sac_run.log 2>&1
wait $GP
i=0
for d in logs/sac-hopper-hip-5seeds/*/; do cp "$d/progress.csv" gpurun_out/r03_seeds/sac_hopper_seed_$(basename "$d" | sed 's/.*--s-//')_progress.csv 2>/dev/null || cp "$d/progress.csv" gpurun_out/r03_seeds/sac_hopper_run${i}_progress.csv; i=$((i+1)); done
for d in logs/gail-walker-hip/*/; do cp "$d/progress.csv" gpurun_out/r03_seeds/gail_walker_progress.csv; done
ls gpurun_out/r03_seeds; tail -3 gpurun_out/r03_seeds/sac_run.log; tail -3 gpurun_out/r03_seeds/gail_run.log
python - <<'PY'
import csv, glob
import numpy as np
fin, best = [], []
for f in sorted(glob.glob("gpurun_out/r03_seeds/sac_hopper_*_progress.csv")):
    rows = list(csv.DictReader(open(f)))
    r = [float(x["Test Returns Mean"]) for x in rows]
    print(f.split("/")[-1], len(rows), "epochs; last", round(r[-1], 1), "mean of last 10", round(float(np.mean(r[-10:])), 1), "best", round(max(r), 1), "env steps", rows[-1]["Number of env steps total"])
    fin.append(np.mean(r[-10:])); best.append(max(r))
if fin:
    print("5-seed: mean of (last-10-epoch mean) %.1f +- %.1f (std over seeds); best-epoch mean %.1f +- %.1f" % (np.mean(fin), np.std(fin), np.mean(best), np.std(best)))
rows = list(csv.DictReader(open("gpurun_out/r03_seeds/gail_walker_progress.csv")))
r = [float(x["Test Returns Mean"]) for x in rows]
print("GAIL Walker:", len(rows), "epochs; best", round(max(r), 1), "at", int(np.argmax(r)), "last-10 mean", round(float(np.mean(r[-10:])), 1))
PY
