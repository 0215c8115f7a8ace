"""Langevin steps/s on the all-pairs paths (nbr_list=None: no PBC; PairListNsqrd: PBC), config C2 (N = 512), GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import unit
from chiron_amd.neighbors import PairListNsqrd, OrthogonalPeriodicSpace

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
fluid, pot, state, thermo, nbr, integ = bench.make_system("c2", 300.0, 32, "cell")
for name, lst in (("NeighborList (fused step kernel)", nbr), ("PairListNsqrd (all pairs, PBC)", PairListNsqrd(OrthogonalPeriodicSpace(), cutoff=1.02 * unit.nanometer)),
                  ("nbr_list=None (all pairs, no PBC)", None)):
    out = integ.run(state, thermo, number_of_steps=200, nbr_list=lst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = integ.run(out[0], thermo, number_of_steps=n_steps, nbr_list=out[1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"N=512 {name:36s}: {n_steps/dt:9.0f} steps/s  ({1e6*dt/n_steps:6.2f} us/step)")
