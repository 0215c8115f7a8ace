"""In-process ABAB: control block published by the step launches themselves vs by one publish launch per chunk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import integrators as I

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
res = {True: [], False: []}
for rep in range(10):
    flag = rep % 2 == 0
    I._HOST_CTRL_IN_KERNEL[0] = flag
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    torch.cuda.synchronize(); res[flag].append(steps / (time.perf_counter() - t0))
print(wl, "in-kernel:", [round(v) for v in res[True]], " publish launch:", [round(v) for v in res[False]])
