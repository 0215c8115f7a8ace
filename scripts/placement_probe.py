"""Is the per-process 'slow mode' of the C3 loop a property of WHERE the integrator's long-lived buffers landed in
device memory?  One process, several fresh systems with dummy allocations of varying size in between (so the caching
allocator hands out different addresses); steps/s per instance plus the addresses of the sticky buffers.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

keep = []
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
for inst in range(8):
    keep.append(torch.empty(((inst * 7919) % 23 + 1) * 1024 * 257, dtype=torch.uint8, device="cuda"))  # shifts later addresses
    fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    res = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=20000, nbr_list=nbr)
        torch.cuda.synchronize(); res.append(20000 / (time.perf_counter() - t0))
    sc = integ.__dict__.get("_inner_scratch", {})
    addr = sc["fl"].data_ptr() if "fl" in sc else 0
    print(f"instance {inst}: {sorted(res)[1]:8.0f} steps/s  {[round(r) for r in res]}   inner list @ {addr:#x} (mod 2 MiB = {addr % (2 << 20):#x})", flush=True)
