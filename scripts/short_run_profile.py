import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=5, nbr_list=nbr)
for rep in range(3):
    state, nbr = integ.run(state, thermo, number_of_steps=20, nbr_list=nbr)
torch.cuda.synchronize()
# host-only time of a call (no sync inside the timed region beyond what run() does itself)
t0 = time.perf_counter()
for rep in range(50):
    state, nbr = integ.run(state, thermo, number_of_steps=20, nbr_list=nbr)
torch.cuda.synchronize()
print("run(20) x50: %.1f us per call" % (1e6 * (time.perf_counter() - t0) / 50))
pr = cProfile.Profile(); pr.enable()
for rep in range(50):
    state, nbr = integ.run(state, thermo, number_of_steps=20, nbr_list=nbr)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
