"""cProfile of the fixed cost of LangevinIntegrator.run (1-step runs; GPU box only)."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
out = integ.run(state, thermo, number_of_steps=10, nbr_list=nbr)
st, nb = out
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    st, nb = integ.run(st, thermo, number_of_steps=1, nbr_list=nb)
torch.cuda.synchronize(); print(f"{wl}: run(1 step) = {1e6*(time.perf_counter()-t0)/50:.0f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    st, nb = integ.run(st, thermo, number_of_steps=1, nbr_list=nb)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
