"""Average each PMC counter per dispatch for the libchx kernels (reads rocprofv3 counter_collection CSVs)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if not (name.startswith("void k_") or name.startswith("k_")):
            continue
        key = name.split("(")[0].replace("void ", "") + " grid=" + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:34s} n={len(v):5d} avg={sum(v)/len(v):16.1f}")
