"""Per-kernel duration statistics from a rocprofv3 kernel trace CSV, beyond the --stats average: how the average of the
step kernel splits into no-op launches (after a halt), the clock warm-up at the start of the process and the steady
state the bench line's event-timed `kernel_us` is measured in.  usage: trace_summary.py <rocprof output dir> [pattern]"""
import csv, glob, os, sys
import numpy as np

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_lj_step<4, true, false, false>"
f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[-1]
start, dur = [], []
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        start.append(s)
        dur.append(e - s)
start, dur = np.array(start), np.array(dur) / 1e3
order = np.argsort(start)
start, dur = start[order], dur[order]
t = (start - start[0]) / 1e9
work = dur > 3.0
print(f"{pat}: {len(dur)} launches over {t[-1]:.1f} s")
print(f"  all launches:                 mean {dur.mean():.3f} us, median {np.median(dur):.3f} us")
print(f"  no-op launches (< 3 us):      {int((~work).sum())} ({100.0 * (~work).mean():.1f} %), mean {dur[~work].mean() if (~work).any() else 0:.3f} us")
print(f"  working launches:             mean {dur[work].mean():.3f} us, median {np.median(dur[work]):.3f} us, "
      f"99th percentile {np.percentile(dur[work], 99):.2f} us, max {dur.max():.1f} us")
for lo, hi in ((0, 1), (1, 3), (3, 6), (6, 1e9)):
    m = work & (t >= lo) & (t < hi)
    if m.any():
        print(f"  working launches, {lo:g}-{hi if hi < 1e8 else t[-1]:.0f} s after the first: {int(m.sum()):7d} launches, mean {dur[m].mean():.3f} us, median {np.median(dur[m]):.3f} us")
