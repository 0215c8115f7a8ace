"""Where does the time of a neighbor-list rebuild inside run() go?  (GPU box only, diagnostic.)

Wraps _RunState._rebuild with timing events (no profiler attached, so the host runs at its real speed) and prints,
per rebuild: the no-op launches queued behind the halting one, the GPU time from the host's detection of the halt
(= an event behind the last no-op it had enqueued) to the end of the rebuild's kernels, and the host time of
_rebuild itself.  usage: python scripts/rebuild_timeline.py [c3|c4] [steps] [launch_window]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import integrators as I

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
if len(sys.argv) > 3:
    integ.launch_window = int(sys.argv[3])
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)

rebuilds = []  # (halt step, host seconds, event before, event after)
orig_reb = I._RunState._rebuild


def reb(self, step=0, speculative=False):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    orig_reb(self, step, speculative)
    t1 = time.perf_counter()
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    rebuilds.append((step, t1 - t0, e0, e1))


I._RunState._rebuild = reb
if os.environ.get("NOGC") == "1":
    import gc
    gc.collect(); gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
I._RunState._rebuild = orig_reb
gpu = np.array([a.elapsed_time(b) * 1e3 for _, _, a, b in rebuilds])
host = np.array([h * 1e6 for _, h, _, _ in rebuilds])
print(f"{wl}: window {integ.launch_window}: {steps} steps in {1e3 * wall:.1f} ms ({steps / wall:.0f} steps/s with the timing events), {len(rebuilds)} rebuilds")
for name, v in (("halt seen by the host -> rebuild kernels done (GPU)", gpu), ("host time of _rebuild", host)):
    print(f"  {name:52s} median {np.median(v):7.1f} us   mean {v.mean():7.1f}   p10 {np.percentile(v, 10):7.1f}   p90 {np.percentile(v, 90):7.1f}")
print(f"  sum of the GPU intervals {gpu.sum() / 1e3:.1f} ms of {1e3 * wall:.1f} ms")
