"""Fixed cost of short run() calls with and without list carry-over (GPU box only).

    python scripts/carry_probe.py [workload] [steps per run ...]

Consecutive run(K) calls on one integrator, each continuing from the state the previous one returned; median wall
time per call between two device synchronisations, carry-over on and off alternating in one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus

pin_process_cpus()
import bench
from chiron_amd import integrators

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
ks = [int(a) for a in sys.argv[2:]] or [20, 100, 400]
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:  # clock warm-up, and an equilibrated fluid
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
torch.cuda.synchronize()
for k in ks:
    for rep in range(2):
        for carry in (False, True):
            integ.list_carry = carry
            integ.n_carried_runs = integ.n_carry_rejected = integ.n_internal_halts = 0
            state, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=nbr)
            torch.cuda.synchronize()
            ts = []
            for _ in range(300):
                t1 = time.perf_counter()
                state, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=nbr)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            ts.sort()
            med = ts[len(ts) // 2]
            print(f"{wl} run({k}) carry-over {'on ' if carry else 'off'}: median {1e6 * med:7.1f} us per call = {k / med:9.0f} steps/s"
                  f"   (carried {integ.n_carried_runs}, rejected {integ.n_carry_rejected}, force-list renewals {integ.n_internal_halts},"
                  f" reference rebuilds in the last call {integ.n_rebuilds})", flush=True)
integ.list_carry = True
