import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=5, nbr_list=nbr)
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=20, nbr_list=nbr)
    torch.cuda.synchronize(); print("run(20):", round(1e6 * (time.perf_counter() - t0), 1), "us")
