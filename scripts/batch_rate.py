"""Aggregate rate of R replicas of a bench workload stepped side by side on one card (integrators.run_batch) against
the serial loop.    python scripts/batch_rate.py [R=8] [steps=4000] [workload=c3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from chiron_amd.integrators import run_batch
from chiron_amd.rng import PRNGKey

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"


def make(k):
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0 * 1.05 ** k, 32, "cell")
    integ.list_carry = False
    state.current_PRNG_key = PRNGKey(1234, k)
    if os.environ.get("BATCH_HINT", "1") != "0":
        nbr.set_replicas_sharing_card(R)  # (what MultiStateSampler tells its lists)
    return [integ, state, thermo, nbr]


reps = [make(k) for k in range(R)]


def batched(n):
    out = run_batch([r[0] for r in reps], [r[1] for r in reps], [r[2] for r in reps], n, [r[3] for r in reps])
    for r, (s, l) in zip(reps, out):
        r[1], r[3] = s, l


def serial(n):
    for r in reps:
        r[1], r[3] = r[0].run(r[1], r[2], number_of_steps=n, nbr_list=r[3])


modes = os.environ.get("BATCH_MODES", "batched,serial,batched").split(",")
for name, fn in [(m, {"batched": batched, "serial": serial}[m]) for m in modes]:
    fn(2000)
    torch.cuda.synchronize()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        rates.append(R * steps / (time.perf_counter() - t0))
    print(f"{name:8s} R={R} {wl}: aggregate replica-steps/s {[round(v) for v in rates]}  rebuilds/replica {reps[0][0].n_rebuilds}", flush=True)
