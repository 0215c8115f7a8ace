"""Analyse a rocprofv3 kernel_trace.csv: busy time, gaps and duration drift of the step kernel."""
import csv, glob, os, sys
import numpy as np
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*_kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
step = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in rows if "k_lj_step" in r["Kernel_Name"]]
step = np.array(step, dtype=np.float64) / 1e3  # us
dur = step[:, 1] - step[:, 0]
gap = step[1:, 0] - step[:-1, 1]
print(f"{len(step)} step launches; total span {1e-3*(step[-1,1]-step[0,0]):.2f} ms; busy {1e-3*dur.sum():.2f} ms")
real = dur > 3.0
print(f"real launches {real.sum()} avg {dur[real].mean():.2f} us; no-op launches {(~real).sum()} avg {dur[~real].mean() if (~real).any() else 0:.2f} us")
n = len(dur)
for a, b in [(0, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4), (3 * n // 4, n)]:
    d = dur[a:b][real[a:b]]
    print(f"  launches {a:5d}-{b:5d}: avg real duration {d.mean():7.2f} us, median gap {np.median(gap[a:b-1]):6.2f} us, gaps>50us: {(gap[a:b-1] > 50).sum()} totalling {gap[a:b-1][gap[a:b-1] > 50].sum()/1e3:.2f} ms")
big = np.argsort(gap)[-8:]
print("largest gaps (us) at launch index:", [(int(i), round(float(gap[i]), 1)) for i in sorted(big)])
