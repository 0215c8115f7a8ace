"""Pretty-print a rocprofv3 *_kernel_stats.csv (usage: show_stats.py <dir-or-file> [rows])."""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*_kernel_stats.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(p)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:>9.2f} "
          f"min={float(r['MinNs'])/1e3:>8.2f} max={float(r['MaxNs'])/1e3:>9.2f} pct={r['Percentage']}")
