"""Deferred list rebuilds against the halting scheme (GPU box only).

    python scripts/deferred_probe.py [workload] [steps] [windows]

In ONE process, alternating: the same system, seed and step count under CHX_DEFERRED_REBUILD on and off.  Prints, per
mode, the median steps/s over `windows` back-to-back windows, the reference's rebuild count and steps (which must be
the same in both modes as long as the trajectories have not drifted apart), the side builds / switches / internal halts
/ late triggers of the deferred schedule, and the largest position difference between the two trajectories after the
first window.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus

print("cpus:", pin_process_cpus(), file=sys.stderr)
import bench
from chiron_amd import integrators

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5


def run(deferred: bool):
    integrators._DEFERRED[0] = deferred
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
    for _ in range(150):  # clock (a FIXED number of runs from the same state: both modes then start from the same bits)
        integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
    rates, first = [], None
    for k in ("n_side_builds", "n_adoptions", "n_internal_halts", "late_triggers"):
        setattr(integ, k, 0)
    integ.__dict__.pop("defer_profile", None)
    rebuilds = 0
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
        torch.cuda.synchronize()
        rates.append(steps / (time.perf_counter() - t0))
        rebuilds += integ.n_rebuilds
        if r == 0:
            first = (state.positions.clone(), list(integ.rebuild_steps), nbr.ref_positions.clone(), nbr.n_neighbors.clone())
    info = {k: getattr(integ, k, 0) for k in ("n_side_builds", "n_adoptions", "n_internal_halts", "late_triggers")}
    prof = getattr(integ, "defer_profile", None)
    if prof and info["n_adoptions"]:
        info["host_us_per_rebuild"] = {k: round(1e6 * v / info["n_adoptions"], 1) for k, v in prof.items()}
    return sorted(rates)[len(rates) // 2], rates, rebuilds, info, first


res = {}
for mode in (True, False, True, False):  # (fixed warm-up: both modes start from the same state)
    res.setdefault(mode, []).append(run(mode))
for mode in (True, False):
    for med, rates, rebuilds, info, first in res[mode]:
        print(f"{wl} deferred={int(mode)}: median {med:9.0f} steps/s {[round(r) for r in rates]} reference rebuilds {rebuilds} {info}")
a, b = res[True][0][4], res[False][0][4]
same = a[1] == b[1]
k = min(len(a[1]), len(b[1]))
agree = next((i for i in range(k) if a[1][i] != b[1][i]), k)
print(f"rebuild steps of the first window: deferred {len(a[1])}, halting {len(b[1])}, identical: {same} (first {agree} agree) "
      f"{a[1][:8]} vs {b[1][:8]}")
print(f"max |dx| after {steps} steps: {(a[0] - b[0]).abs().max().item():.3e} nm; public list after the window: ref positions equal "
      f"{torch.equal(a[2], b[2])}, n_neighbors equal {torch.equal(a[3], b[3])}")
