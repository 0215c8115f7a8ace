"""Device time line of short run() calls from a rocprofv3 kernel trace (GPU box only).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/srt -- python3 $REPO/scripts/short_run_timeline.py run c3 20 [carry]
    python3 scripts/short_run_timeline.py report /tmp/srt

`run`: 3 s of clock warm-up, then 200 x run(K) (build on entry unless `carry` is given), each followed by a device
synchronisation; prints the median wall time per call.  `report`: the calls are cut apart at the idle gaps between them;
for the median call: every kernel's duration and the gap in front of it, device span and busy time.
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
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    carry = len(sys.argv) > 4 and sys.argv[4] == "carry"
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    integ.list_carry = carry
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        time.sleep(0.0005)  # (an idle gap the report can cut at)
        t1 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=nbr)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t1)
    ts.sort()
    print(f"{wl} run({k}) carry={carry}: median wall {1e6 * ts[len(ts) // 2]:.1f} us per call")


def report(root):
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]))
    rows.sort()
    calls, cur = [], []
    for s, e, name in rows:
        if cur and s - cur[-1][1] > 300_000:  # > 0.3 ms idle: a new call (or the warm-up's end)
            calls.append(cur)
            cur = []
        cur.append((s, e, name))
    if cur:
        calls.append(cur)
    calls = [c for c in calls if 5 <= len(c) <= 400][-200:]
    if not calls:
        print("no short calls found")
        return
    spans = sorted((c[-1][1] - c[0][0], i) for i, c in enumerate(calls))
    c = calls[spans[len(spans) // 2][1]]
    print(f"{len(calls)} calls; median device span {spans[len(spans) // 2][0] / 1e3:.1f} us, {len(c)} kernels in the median call")
    busy = 0
    prev = c[0][0]
    steps = []
    for s, e, name in c:
        busy += e - s
        if name.startswith("void k_lj_step") or name.startswith("k_lj_step"):
            steps.append((s - prev, e - s))
        else:
            if steps:
                print(f"   {len(steps)} x k_lj_step: first gap {steps[0][0] / 1e3:.1f} + {steps[0][1] / 1e3:.1f} us, then mean gap "
                      f"{sum(g for g, _ in steps[1:]) / max(1, len(steps) - 1) / 1e3:.2f} + mean {sum(d for _, d in steps[1:]) / max(1, len(steps) - 1) / 1e3:.2f} us")
                steps = []
            print(f"   gap {(s - prev) / 1e3:6.1f} us  {name[:60]:60s} {(e - s) / 1e3:6.1f} us")
        prev = e
    if steps:
        print(f"   {len(steps)} x k_lj_step: first gap {steps[0][0] / 1e3:.1f} + {steps[0][1] / 1e3:.1f} us, mean of the rest gap "
              f"{sum(g for g, _ in steps[1:]) / max(1, len(steps) - 1) / 1e3:.2f} + {sum(d for _, d in steps[1:]) / max(1, len(steps) - 1) / 1e3:.2f} us")
    print(f"   device busy {busy / 1e3:.1f} us of span {(c[-1][1] - c[0][0]) / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
