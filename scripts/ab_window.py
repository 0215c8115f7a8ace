"""In-process ABAB over the integrator's launch window (0 = two whole chunks ahead, the older scheme).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
wins = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "8", "12", "16", "24"])]
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:  # clock warm-up
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
res = {w: [] for w in wins}
for rep in range(4):
    for w in wins:
        integ.launch_window = w
        torch.cuda.synchronize(); t0 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
        torch.cuda.synchronize(); res[w].append(steps / (time.perf_counter() - t0))
for w in wins:
    print(wl, "window", w, [round(v) for v in res[w]], "mean", round(sum(res[w]) / len(res[w])), "rebuilds/run", integ.n_rebuilds)
