"""cProfile of LangevinIntegrator.run on the host (GPU box only): python run_host_profile.py c3 3000"""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
wl = sys.argv[1]; steps = int(sys.argv[2])
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
out = integ.run(state, thermo, number_of_steps=300, nbr_list=nbr)
pr = cProfile.Profile(); pr.enable()
out = integ.run(out[0], thermo, number_of_steps=steps, nbr_list=out[1])
torch.cuda.synchronize(); pr.disable()
print("rebuilds", integ.n_rebuilds)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
