"""One system state, one table row: rocprofv3's average kernel duration on that state against the kernel's algorithmic
bytes (pair force 24 N + 16 P_nl; fused step + 72 N) and the 8 TB/s HBM peak.

    python scripts/state_roofline.py <rocprofv3 --stats dir of a `kernel_bench.py <workload>` pass> <its log> <workload>

The pass runs kernel_bench.py on ONE state (the fluid equilibrated for KB_WARM steps, loaded from the on-disk snapshot,
list built from it), so every k_lj_fl launch in the directory is a launch on that state and the fraction recomputes
from this one row: frac = bytes / (avg_us * 1e-6) / 8e12.  (The pass also launches the fused step kernel, but with its
skin check disabled: the list ages while it is timed and the pruned rows grow shorter than a live run's -- no row is
printed for it; its live figure is `roofline` in the bench line and the driver-command table.)"""
import csv, glob, os, re, sys

prof, log, wl = sys.argv[1:4]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n, rho = bench.WORKLOADS[wl]
line = [l for l in open(log) if l.startswith(wl + " g=")][-1]
p_nl = int(re.search(r"P_nl=(\d+)", line).group(1))
g = int(re.search(r"g=(\d+)", line).group(1))
ev_force = float(re.search(r"force\s+([\d.]+) us", line).group(1))
rows = {}
for path in glob.glob(os.path.join(prof, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
force = [(k, v) for k, v in rows.items() if k.startswith("void k_lj_fl<") and "true, false>" in k]
step = [(k, v) for k, v in rows.items() if k.startswith("void k_lj_step<") and ", true, false, false>" in k]
bf, bs = 24 * n + 16 * p_nl, 96 * n + 16 * p_nl
for kind, found, b in (("pair force `k_lj_fl`", force, bf),):
    for name, (calls, avg) in found:
        print(f"| {wl} (N = {n}, rho* = {rho}) | {kind} | {g} | {p_nl} | {b} | {calls} | {avg:.2f} | {b / (avg * 1e-6) / 1e9:.0f} | "
              f"**{b / (avg * 1e-6) / 8e12:.3f}** | {ev_force:.2f} |" if kind.startswith("pair") else
              f"| {wl} (N = {n}, rho* = {rho}) | {kind} | {g} | {p_nl} | {b} | {calls} | {avg:.2f} | {b / (avg * 1e-6) / 1e9:.0f} | "
              f"{b / (avg * 1e-6) / 8e12:.3f} | |")
