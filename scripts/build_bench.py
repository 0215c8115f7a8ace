"""Time of one neighbor-list build (all launches of NeighborList.build incl. its one host sync),
GPU box only: python scripts/build_bench.py [c3 c4 c4_liquid]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

for wl in sys.argv[1:] or ["c3", "c4"]:
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    for _ in range(3):
        nbr.build_from_state(state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        nbr.build_from_state(state)
    torch.cuda.synchronize()
    print(f"{wl}: build: {1e6*(time.perf_counter()-t0)/reps:8.1f} us   pairs {int(nbr.n_neighbors.sum())}")
