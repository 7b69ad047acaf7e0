#!/usr/bin/env python
"""Fold the per-pass rocprofv3 counter CSVs written by tools/pmc_collect.sh into per-kernel averages (JSON).
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; the gfx950 correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE
tallies 128-B requests at 64 B: double it for wide coalesced reads) is applied in `hbm_read_bytes_corrected`."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "").split("(")[0]
            k = k.replace("void ", "").split("<")[0].strip()
            c, v = row.get("Counter_Name"), float(row.get("Counter_Value", 0))
            acc[k][c][0] += v
            acc[k][c][1] += 1
out = {}
for k, cs in acc.items():
    d = {c: s / n for c, (s, n) in cs.items()}
    d["launches"] = max(n for _, n in cs.values())
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024.0
        d["hbm_read_bytes_corrected"] = 2.0 * d["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes_raw"] = d["WRITE_SIZE"] * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_over_cu_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CU_CYCLES"]
    out[k] = d
# which kernels these counters describe: a hash of the library's sources (bench.py recomputes it and flags `traffic_stale` when the tree moved on)
import hashlib
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(repo, "ilswiss_amd", "csrc", "*.hip")) + glob.glob(os.path.join(repo, "ilswiss_amd", "csrc", "*.h"))
                + glob.glob(os.path.join(repo, "ilswiss_amd", "csrc", "*.inc")) + [os.path.join(repo, "include", "ilsx.h")]):
    h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
out["_meta"] = dict(csrc_sha256=h.hexdigest())
json.dump(out, sys.stdout, indent=1, sort_keys=True)
