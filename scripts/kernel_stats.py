"""Per-kernel average duration from a rocprofv3 --kernel-trace --stats output directory.
usage: python scripts/kernel_stats.py /tmp/dir [substring ...]"""
import csv, glob, os, sys
root = sys.argv[1]
want = sys.argv[2:]
for path in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0]
        if want and not any(w in name for w in want):
            continue
        print(f"{name[:70]:70s} calls {int(r['Calls']):8d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {float(r['Percentage']):5.1f} %")
