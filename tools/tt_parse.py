import csv, glob, sys, collections
mode = sys.argv[1]
f = glob.glob(f"/root/repo/gpurun_out/tt_{mode}/runc/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_mlp" in r["Kernel_Name"] or "k_sac" in r["Kernel_Name"]]
short = lambda n: "F" if "fwd_split" in n else "B" if "bwd_split" in n else "D" if "bwd_dw" in n else "T" if "tail" in n else "?"
seq = "".join(short(k[0]) for k in ks)
pat = "FFBDFBBDT" if "T" in seq[20:60] else "FFBDFBBD"
L = len(pat)
i = seq.find(pat * 3)
per = collections.defaultdict(list)
cnt = 0
while seq[i:i + L] == pat and cnt < 50:
    for j in range(L):
        per[j].append((ks[i + j][2] - ks[i + j][1]) / 1000)
    i += L; cnt += 1
meds = [sorted(per[j])[len(per[j]) // 2] for j in range(L)]
print(mode, cnt, " ".join(f"{pat[j]}{j}:{meds[j]:.2f}" for j in range(L)), "sum", round(sum(meds), 2))
