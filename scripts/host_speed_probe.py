"""Is the per-process spread of the C3 figure a HOST effect?  Prints, for this process: the CPU it runs on, a pure-Python
loop time, the host time of a list rebuild inside run(), and steps/s.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import integrators as I

p = psutil.Process()


def pyloop():
    t0 = time.perf_counter()
    s = 0
    for i in range(2_000_000):
        s += i & 7
    return time.perf_counter() - t0


cpu0 = p.cpu_num()
t_py = min(pyloop() for _ in range(3))
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
host = []
orig = I._RunState._rebuild


def reb(self, step=0, speculative=False):
    t = time.perf_counter()
    orig(self, step, speculative)
    host.append(time.perf_counter() - t)


I._RunState._rebuild = reb
res = []
cpus = set()
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=20000, nbr_list=nbr)
    torch.cuda.synchronize(); res.append(20000 / (time.perf_counter() - t0))
    cpus.add(p.cpu_num())
print(f"cpu {cpu0} -> {sorted(cpus)}  python loop {1e3 * t_py:6.1f} ms  rebuild host time median {1e6 * np.median(host):6.1f} us  "
      f"steps/s median {sorted(res)[2]:.0f}  {[round(r) for r in res]}  affinity {len(os.sched_getaffinity(0))} cpus", flush=True)
