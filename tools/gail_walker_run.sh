#!/bin/bash
# Config 3 end to end on the GPU box: expert = the SAC Walker2d policy of demos/walker_sac_expert_policy.npz (trained by
# exp_specs/sac/sac_walker_hip.yaml) -> demonstrations in the reference's pickle format -> GAIL (gail_walker_hip.yaml).
set -u
SPEC=${1:-gail_walker_hip}          # spec name under exp_specs/gail/ (also names the log dir and the progress file)
LOGNAME=$(echo $SPEC | tr _ -)
python - <<PY
import pickle, numpy as np
d = np.load("demos/walker_sac_expert_policy.npz")
pickle.dump(dict(policy=d["policy"]), open("/tmp/walker_expert.pkl", "wb"))
print("expert policy from epoch", int(d["epoch"]), "AverageReturn", float(d["average_return"]))
PY
python run_scripts/gen_expert_demos.py --snapshot /tmp/walker_expert.pkl --env walker --num-trajs 16 --out demos/walker_sac.pkl
python run_experiment.py -e exp_specs/gail/$SPEC.yaml -g 0 > gpurun_out/${SPEC}_run.log 2>&1
f=$(ls -d logs/$LOGNAME/*/ | head -1)
cp $f/progress.csv gpurun_out/r02_${SPEC}_progress.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r02_${SPEC}_progress.csv")))
print(len(rows), "epochs")
for r in rows[::6] + rows[-3:]:
    print(r["Epoch"], r["Number of env steps total"], r.get("Number of train steps total", r.get("Number of gradient steps total")), round(float(r["AverageReturn"]), 1),
          round(float(r["Test Ep. Len. Mean"]), 1), round(float(r["Disc Acc"]), 3), round(float(r["Disc Rew Mean"]), 3), "alpha", round(float(r["Alpha"]), 3), round(float(r["Total Train Time (s)"]), 1))
PY
