"""cProfile of NeighborListCell.build + _RunState._load on the host (GPU box only)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
nbr.build_from_state(state)
x = state.positions
box = state.box_vectors
for _ in range(3):
    nbr.build(x, box)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    nbr.build(x, box)
torch.cuda.synchronize()
print(f"{wl}: build {1e6*(time.perf_counter()-t0)/20:.0f} us each (20 back-to-back)")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    nbr.build(x, box)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
