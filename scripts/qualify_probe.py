"""Does a rank's stand-alone rate, measured in its first ~1.5 s, predict the rate it keeps?  (GPU box only.)

The requalification of bench.py's launcher (bench.requalify) replaces ranks whose qualification rate is below 0.97 x the
median.  One GPU is all this box has, so the ranks of an 8-GPU run are rehearsed as fresh processes ONE AFTER ANOTHER
(nothing shares the card): each measures (a) the qualification rate exactly as bench.py's children do (1 s of stepping,
median of three 10 000-step windows) and then (b) what it would deliver afterwards (3 s more, median of three
20 000-step windows).  Printed: both per process, and the table  flagged x slow-afterwards.

usage: python scripts/qualify_probe.py 40        (parent)      python scripts/qualify_probe.py child   (one process)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch

    torch.set_num_threads(1)
    from chiron_amd._device import pin_process_cpus

    pin_process_cpus()
    import bench

    fluid, pot, st, th, nb, ig = bench.make_system("c3", 300.0, 32, "cell")
    ig.list_carry = False

    def windows(warm_s, steps):
        nonlocal st, nb
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < warm_s:
            st, nb = ig.run(st, th, number_of_steps=2000, nbr_list=nb)
            torch.cuda.synchronize()
        out = []
        for _ in range(3):
            t1 = time.perf_counter()
            st, nb = ig.run(st, th, number_of_steps=steps, nbr_list=nb)
            torch.cuda.synchronize()
            out.append(steps / (time.perf_counter() - t1))
        return sorted(out)[1]

    q = windows(1.0, 10000)
    later = windows(3.0, 20000)
    print(json.dumps({"qualify": q, "later": later}), flush=True)


def parent(n):
    rows = []
    for i in range(n):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"process {i}: failed ({r.returncode}) {r.stderr[-300:]}", flush=True)
            continue
        d = json.loads(line[-1])
        rows.append(d)
        print(f"process {i:3d}: qualification {d['qualify'] / 1e3:7.1f} k steps/s, afterwards {d['later'] / 1e3:7.1f} k", flush=True)
    q_med = sorted(r["qualify"] for r in rows)[len(rows) // 2]
    l_med = sorted(r["later"] for r in rows)[len(rows) // 2]
    flagged = [r["qualify"] < 0.97 * q_med for r in rows]
    slow = [r["later"] < 0.97 * l_med for r in rows]
    both = sum(f and s for f, s in zip(flagged, slow))
    print(f"{len(rows)} processes; medians: qualification {q_med / 1e3:.1f} k, afterwards {l_med / 1e3:.1f} k steps/s")
    print(f"slow afterwards (< 0.97 x median): {sum(slow)}; flagged by the qualification: {sum(flagged)}; flagged AND slow: {both}; "
          f"slow but NOT flagged (what requalification would let through): {sum(slow) - both}; flagged but fine afterwards: {sum(flagged) - both}")
    kept = [r["later"] for r, f in zip(rows, flagged) if not f]
    print(f"kept ranks: slowest {min(kept) / 1e3:.1f} k = {min(kept) / l_med:.3f} x median; all ranks: slowest {min(r['later'] for r in rows) / 1e3:.1f} k "
          f"= {min(r['later'] for r in rows) / l_med:.3f} x median")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        parent(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
