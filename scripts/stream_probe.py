"""Does the per-process spread of the step time (4.7 .. 6.4 us) depend on the HIP stream / hardware queue?
(GPU box only.)  Times the fused step on the default stream and on several fresh streams in ONE process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
torch.cuda.synchronize()
res = []
t, _, _, _ = bench.time_kernels(fluid, pot, state, nbr, integ, 640)
res.append(("default", t * 1e6))
for i in range(5):
    s = torch.cuda.Stream(priority=-1 if i == 4 else 0)
    with torch.cuda.stream(s):
        t, _, _, _ = bench.time_kernels(fluid, pot, state, nbr, integ, 640)
    res.append((f"stream{i}" + ("(high prio)" if i == 4 else ""), t * 1e6))
t, _, _, _ = bench.time_kernels(fluid, pot, state, nbr, integ, 640)
res.append(("default again", t * 1e6))
print("  ".join(f"{n}: {v:.2f}" for n, v in res))
