"""Durations and inter-kernel gaps of consecutive dispatches from a rocprofv3 --kernel-trace CSV.
usage: python scripts/trace_gaps.py <dir-or-csv> [kernel-name-substring]"""
import csv
import glob
import os
import sys

import numpy as np

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_lj_step"
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sel = [(s, e, k) for s, e, k in rows if pat in k]
print(f"{len(rows)} dispatches, {len(sel)} match '{pat}'")
by = {}
for i in range(1, len(rows)):
    s0, e0, k0 = rows[i - 1]
    s1, e1, k1 = rows[i]
    if pat in k0 and pat in k1:
        short = k1.split("(")[0][-40:]
        by.setdefault(short, []).append((e1 - s1, s1 - e0, s1 - s0))
for k, v in by.items():
    a = np.array(v, dtype=np.float64) * 1e-3
    print(f"{k:42s} n={len(a):6d}  duration med {np.median(a[:,0]):6.2f} us  gap med {np.median(a[:,1]):6.2f} us  start-to-start med {np.median(a[:,2]):6.2f} us  (p10 {np.percentile(a[:,2],10):.2f}, p90 {np.percentile(a[:,2],90):.2f})")
