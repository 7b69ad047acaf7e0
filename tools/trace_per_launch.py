"""Per-launch durations of the SAC step from a rocprofv3 kernel trace of tools/tail_trace.py (median over the steady-state steps).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tt_<tag> -- python $REPO/tools/tail_trace.py
    python tools/trace_per_launch.py <tag>        # F = forward, B = backward, D = weight gradients (+Adam), T = tail
"""
import csv, glob, sys, statistics
mode = sys.argv[1]
f = glob.glob(f"/root/repo/gpurun_out/tt_{mode}/runc/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_mlp" in r["Kernel_Name"] or "k_sac" in r["Kernel_Name"] or "nop" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: "F" if "fwd_split" in n else "B" if "bwd_split" in n else "D" if "bwd_dw" in n else "T" if "tail" in n else "N" if "nop" in n else "?"
seq = "".join(short(r["Kernel_Name"]) for r in rows)
for pat in ("FFBDFBBDT", "FFBDFBBDN", "FFBDFBBD"):
    i = seq.find(pat * 3)
    if i >= 0:
        break
L = len(pat); out = []
while seq[i:i + L] == pat:
    out.append([(int(rows[i + j]["End_Timestamp"]) - int(rows[i + j]["Start_Timestamp"])) / 1000 for j in range(L)]); i += L
med = [statistics.median(o[j] for o in out) for j in range(L)]
print(mode, len(out), " ".join(f"{pat[j]}{j}:{med[j]:.2f}" for j in range(L)), "sum", round(sum(med), 2))
