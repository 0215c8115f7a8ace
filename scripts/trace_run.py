"""Kernel timeline of ONE short run() call from a rocprofv3 --kernel-trace CSV (the last `n` dispatches before the
final k_lj_fl of bench's kernel timing are not needed: we take a window by kernel names).
usage: python scripts/trace_run.py <dir> """
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-46:]))
rows.sort()
# find the LAST occurrence of a cell-build kernel followed by exactly the short run's step launches
idx = [i for i, r in enumerate(rows) if "k_cell_assign" in r[2]]
want = int(sys.argv[2]) if len(sys.argv) > 2 else -1
i0 = idx[want]
t0 = rows[i0][0]
prev_end = t0
for s, e, k in rows[max(0, i0 - 3): i0 + 60]:
    print(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  {k}")
    prev_end = e
