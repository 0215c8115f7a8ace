"""Median steps/s of one workload at the loaded clock (3 s warm-up, then `reps` windows).  GPU box only.
usage: python scripts/steps_rate.py c3 20000 5      (CHX_LIB=<path> selects a variant build of libchx.so)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
print("cpus:", pin_process_cpus(), file=sys.stderr)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
res = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    torch.cuda.synchronize(); res.append(steps / (time.perf_counter() - t0))
print(f"{wl} [{os.environ.get('CHX_LIB', 'product')}]: median {sorted(res)[len(res) // 2]:.0f} steps/s  {[round(r) for r in res]}")
