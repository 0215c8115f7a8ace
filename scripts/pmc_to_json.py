"""Reduce the rocprofv3 --pmc passes of scripts/pmc_collect.sh to profiles/pmc_traffic.json and a
markdown table (usage: pmc_to_json.py <pmc-dir> <out.json> <out.md>).

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (counters in KiB; FETCH_SIZE under-reports by 2x
on gfx950 -- MI355X_MICROARCH.md, HBM / rocprofv3 section), averaged over the dispatches of a kernel.
"""
import collections, csv, glob, json, os, re, sys

root, out_json, out_md = sys.argv[1:4]
# the system state the passes ran on: P_nl per workload from the benchmark's own output, equilibration from meta.json
pairs = {}
for f in sorted(glob.glob(os.path.join(root, "pass*.log"))):
    for line in open(f, errors="replace"):
        m = re.match(r"(c\d\w*) g=\d+ .*P_nl=(\d+)", line)
        if m:
            pairs.setdefault(m.group(1), set()).add(int(m.group(2)))
for wl, v in pairs.items():
    assert len(v) == 1, f"the passes of {wl} did not run on one state: P_nl = {sorted(v)}"
try:
    meta = json.load(open(os.path.join(root, "meta.json")))
except OSError:
    meta = {}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        m = re.match(r"(?:void )?(k_lj_fl|k_lj_step)<(\d+)((?:, (?:true|false))*)>", name)
        if not m:
            continue
        g = int(m.group(2))
        flags = [f.strip() for f in m.group(3).split(",") if f.strip()]
        n = int(r["Grid_Size"]) // g
        wl = {32768: "c3", 262144: "c4"}.get(n)
        if wl is None:
            continue
        # k_lj_step<G, HOT, PRUNE>: the launches that also refresh the inner list are a kernel of their own
        kern = m.group(1) + ("+refresh" if m.group(1) == "k_lj_step" and len(flags) >= 2 and flags[1] == "true" else "")
        acc[(wl, kern, g)][r["Counter_Name"]].append(float(r["Counter_Value"]))

res = {}
rows = []
for (wl, kern, g), c in sorted(acc.items()):
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    fetch = avg.get("FETCH_SIZE", 0.0) * 1024
    write = avg.get("WRITE_SIZE", 0.0) * 1024
    e = {
        "lanes_per_particle": g,
        "list_pairs": (sorted(pairs[wl])[0] if wl in pairs else None),
        "equilibration_steps": meta.get("equilibration_steps"),
        "source": f"profiles/{os.path.basename(out_md)}",
        "fetch_size_bytes_raw": fetch,
        "write_size_bytes": write,
        "traffic_bytes": 2 * fetch + write,
        "tcc_hit_rate": avg.get("TCC_HIT_sum", 0.0) / max(1.0, avg.get("TCC_REQ_sum", 0.0)),
        "l1_hit_rate": 1.0 - avg.get("TCP_TCC_READ_REQ_sum", 0.0) / max(1.0, avg.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)),
        "sq_wait_any_frac": avg.get("SQ_WAIT_ANY", 0.0) / max(1.0, avg.get("SQ_WAVE_CYCLES", 0.0)),
        "sq_active_frac": avg.get("SQ_ACTIVE_INST_ANY", 0.0) / max(1.0, avg.get("SQ_WAVE_CYCLES", 0.0)),
        "valu_wave_insts": avg.get("SQ_INSTS_VALU", 0.0),
        "dispatches": max(len(v) for v in c.values()),
        "counters": avg,
    }
    res[f"{wl}:{kern}"] = e
    rows.append(f"| {wl} | `{kern}<{g}>` | {e['traffic_bytes']/1e6:.1f} | {fetch/1e6:.1f} | {write/1e6:.1f} | "
                f"{100*e['tcc_hit_rate']:.0f} % | {100*e['l1_hit_rate']:.0f} % | {100*e['sq_wait_any_frac']:.0f} % | "
                f"{100*e['sq_active_frac']:.0f} % | {e['valu_wave_insts']/1e6:.2f} M |")
json.dump(res, open(out_json, "w"), indent=1)
with open(out_md, "w") as fh:
    fh.write("# PMC passes over the hot kernels (scripts/pmc_collect.sh: one counter group per rocprofv3 --pmc pass,\n"
             f"# `python3 scripts/kernel_bench.py c3 c4` on the fluid equilibrated for {meta.get('equilibration_steps')} steps; "
             f"P_nl = {dict((k, sorted(v)[0]) for k, v in pairs.items())})\n\n"
             "HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters, gfx950 FETCH_SIZE correction).\n\n"
             "| workload | kernel | HBM traffic MB | FETCH_SIZE MB (raw) | WRITE_SIZE MB | L2 hit | L1 hit | SQ_WAIT_ANY / wave cycles | "
             "SQ_ACTIVE_INST_ANY / wave cycles | VALU wave instructions |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    fh.write("\n".join(rows) + "\n")
print("\n".join(rows))
