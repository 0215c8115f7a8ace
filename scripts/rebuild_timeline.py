"""Device time line of the list rebuilds inside a Langevin run, from a rocprofv3 kernel trace (GPU box only).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python3 $REPO/scripts/rebuild_timeline.py run
    python3 scripts/rebuild_timeline.py report /tmp/rt

`run`: 3 s of clock warm-up, then 6000 steps at C3 (about 28 rebuilds).  `report`: for every rebuild of the last 6000
steps -- the stretch between the last step launch that did real work and the first one of the next generation -- the
wall span on the device, the time the device spent in kernels, and what those kernels were.
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import time

    import torch

    torch.set_num_threads(1)
    from chiron_amd._device import pin_process_cpus

    pin_process_cpus()
    import bench

    wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
    state, nbr = integ.run(state, thermo, number_of_steps=6000, nbr_list=nbr)
    torch.cuda.synchronize()
    print("rebuilds in the last run:", integ.n_rebuilds)


def report(root):
    f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    rows = rows[-9000:]
    spans = []
    i = 0
    while i < len(rows):
        if "k_nl_build_cell" in rows[i][2]:
            # walk back to the last step launch that took more than 3 us (real work), forward to the first step launch
            b = i
            while b > 0 and not ("k_lj_step" in rows[b][2] and rows[b][1] - rows[b][0] > 3000):
                b -= 1
            e = i
            while e < len(rows) - 1 and "k_lj_step" not in rows[e][2]:
                e += 1
            seg = rows[b + 1:e]
            busy = sum(r[1] - r[0] for r in seg)  # (kernels strictly between the two step launches)
            seg = [rows[b]] + seg + [rows[e]]  # (bracketed by the step launches either side, for the gap table)
            spans.append((rows[e][0] - rows[b][1], busy, seg))
            i = e
        i += 1
    if not spans:
        print("no rebuild found")
        return
    spans = spans[2:]
    n = len(spans)
    wall = sorted(s[0] for s in spans)[n // 2] / 1e3
    busy = sorted(s[1] for s in spans)[n // 2] / 1e3
    print(f"{n} rebuilds: median device span {wall:.1f} us from the end of the last working step launch to the start of the next, "
          f"{busy:.1f} us of it inside kernels")
    acc = {}
    for _, _, seg in spans:
        for s, e, name in seg:
            k = name.split("(")[0].replace("void ", "")[:44]
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += (e - s) / 1e3
    for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:46s} {c / n:5.1f} launches per rebuild, {t / n:6.1f} us")
    # where the idle time sits: the gap in FRONT of each kernel of the chain (median over the rebuilds)
    gaps = {}
    for _, _, seg in spans:
        prev_end = None
        for s_, e_, name in seg:
            k = name.split("(")[0].replace("void ", "")[:44]
            if prev_end is not None:
                gaps.setdefault(k, []).append((s_ - prev_end) / 1e3)
            prev_end = e_
    print("  idle gap in front of (median us):")
    for k, v in gaps.items():
        v.sort()
        print(f"    {k:44s} {v[len(v) // 2]:6.1f}   (n = {len(v)})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run()
