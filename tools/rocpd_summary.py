#!/usr/bin/env python
"""Dump the per-kernel stats of a rocprofv3 (rocpd sqlite) result as CSV: name,calls,total_us,avg_us,pct.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.split("(")[0].replace("void ", "")
    print(f"\"{short}\",{calls},{total:.1f},{avg:.3f},{pct:.2f}")
