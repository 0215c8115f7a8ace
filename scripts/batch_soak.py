"""Soak of the batched replica launch (GPU box only): R replicas of a workload for `steps` steps in calls of `call` steps,
stepped side by side (integrators.run_batch) and one after the other; the final states must hash the same.
usage: python scripts/batch_soak.py [R=8] [steps=50000] [call=5000] [workload=c3]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from chiron_amd.integrators import run_batch
from chiron_amd.rng import PRNGKey

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
call = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
wl = sys.argv[4] if len(sys.argv) > 4 else "c3"


def make(k):
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0 * 1.05 ** k, 32, "cell")
    integ.list_carry = False
    state.current_PRNG_key = PRNGKey(1234, k)
    nbr.set_replicas_sharing_card(R)
    return [integ, state, thermo, nbr]


def digest(reps):
    h = hashlib.sha256()
    for r in reps:
        h.update(r[1].positions.cpu().numpy().tobytes() + r[1].velocities.cpu().numpy().tobytes() + r[3].ref_positions.cpu().numpy().tobytes())
    return h.hexdigest()


out = {}
for mode in ("batched", "serial"):
    reps = [make(k) for k in range(R)]
    rebuilds = 0
    t0 = time.perf_counter()
    for _ in range(steps // call):
        if mode == "batched":
            res = run_batch([r[0] for r in reps], [r[1] for r in reps], [r[2] for r in reps], call, [r[3] for r in reps])
        else:
            res = [r[0].run(r[1], r[2], number_of_steps=call, nbr_list=r[3]) for r in reps]
        for r, (s, l) in zip(reps, res):
            r[1], r[3] = s, l
        rebuilds += sum(r[0].n_rebuilds for r in reps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[mode] = (digest(reps), rebuilds, R * steps / dt)
    print(f"{mode:8s} {R} x {wl} x {steps} steps: hash {out[mode][0][:16]}  list rebuilds {rebuilds}  {out[mode][2]:.0f} replica-steps/s", flush=True)
same = out["batched"][0] == out["serial"][0] and out["batched"][1] == out["serial"][1]
print("IDENTICAL" if same else "DIFFERENT")
sys.exit(0 if same else 1)
