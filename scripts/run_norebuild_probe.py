"""Steps/s of LangevinIntegrator.run() when no rebuild can happen (huge skin is not an option -- it changes the list --
so: count steps between rebuilds instead).  Prints the time of run(n) for n below the first rebuild.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_num_threads(1)
import bench

fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
for n in (32, 64, 96, 128, 160):
    res = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=n, nbr_list=nbr)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if integ.n_rebuilds == 0:
            res.append(dt * 1e6)
    print(f"run({n:3d}) without a rebuild inside: median {np.median(res):7.1f} us ({len(res)} samples)")
