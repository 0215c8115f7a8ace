"""In-process ABAB over the integrator's chunk size (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
sizes = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["32", "16", "8"])]
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
res = {c: [] for c in sizes}
for rep in range(4):
    for c in sizes:
        integ.chunk_size = c
        torch.cuda.synchronize(); t0 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
        torch.cuda.synchronize(); res[c].append(steps / (time.perf_counter() - t0))
for c in sizes:
    print(wl, "chunk", c, [round(v) for v in res[c]], "mean", round(sum(res[c]) / len(res[c])))
