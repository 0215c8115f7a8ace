"""Does launch_regime_probe tell the slow launch regime?  One process: probe (20 000 launches, spinning window), then the
usual rate measurement.  Run it many times (scripts/regime_sweep.sh style) and compare.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import launch_regime_probe, pin_process_cpus
pin_process_cpus()
import bench

fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
torch.cuda.synchronize()
probe = launch_regime_probe(lambda k: integ.run(state, thermo, number_of_steps=k, nbr_list=nbr))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
import ctypes
from chiron_amd import _lib
lib = _lib.load()
lib.chx_poll_stats(None, 1)
res = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=20000, nbr_list=nbr)
    torch.cuda.synchronize(); res.append(20000 / (time.perf_counter() - t0))
st = (ctypes.c_int64 * 4)()
lib.chx_poll_stats(st, 0)
waits, naps, dry, ns = [int(v) for v in st]
print(f"probe {probe}  ->  median {sorted(res)[1]:.0f} steps/s {[round(r) for r in res]}  poll: {waits} waits, {naps} naps "
      f"({ns / max(1, naps) / 1e3:.1f} us each), {dry} dry wake-ups ({100.0 * dry / max(1, waits):.2f} %)")
