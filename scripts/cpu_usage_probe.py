"""steps/s and host CPU seconds of a Langevin run (GPU box only): python cpu_usage_probe.py c3 6000"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
wl = sys.argv[1]; steps = int(sys.argv[2])
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
out = integ.run(state, thermo, number_of_steps=300, nbr_list=nbr)
torch.cuda.synchronize()
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
out = integ.run(out[0], thermo, number_of_steps=steps, nbr_list=out[1])
torch.cuda.synchronize()
dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
print(f"{wl}: {steps/dt:.0f} steps/s, wall {1e3*dt:.1f} ms, process CPU {1e3*cpu:.1f} ms ({cpu/dt:.2f} cores)")
