"""Where does the time of a neighbor-list rebuild inside run() go?  (GPU box only, diagnostic.)

Wraps _RunState._enqueue_fused / _rebuild with timing events (no profiler attached, so the host side runs at
its real speed) and prints, per rebuild: the GPU time of the halted chunk beyond its real steps (the no-op tail
inside that chunk), the GPU time from the end of the halted chunk to the end of the rebuild's kernels (no-op
chunks queued behind + waiting for the host + the build), and the host time of _rebuild itself.
usage: python scripts/rebuild_timeline.py [c3|c4] [steps]"""
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
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)

log = []          # (s0, n, start event, end event)
rebuilds = []     # (halt step, host seconds, event after the rebuild's kernels)
orig_enq, orig_reb = I._RunState._enqueue_fused, I._RunState._rebuild


def enq(self, s0, n, do_aoa=1):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rec = orig_enq(self, s0, n, do_aoa)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    log.append((s0, n, e0, e1))
    return rec


def reb(self, step=0, speculative=False):
    t0 = time.perf_counter()
    orig_reb(self, step, speculative)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    rebuilds.append((step, time.perf_counter() - t0, e))


I._RunState._enqueue_fused, I._RunState._rebuild = enq, reb
torch.cuda.synchronize(); t0 = time.perf_counter()
state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
I._RunState._enqueue_fused, I._RunState._rebuild = orig_enq, orig_reb

full = np.array([a.elapsed_time(b) * 1e3 / n for s0, n, a, b in log if n == integ.chunk_size])
t_step = float(np.median(full))
print(f"{wl}: {steps} steps in {1e3 * wall:.1f} ms ({steps / wall:.0f} steps/s, with the timing events), {len(rebuilds)} rebuilds, "
      f"median chunk = {t_step:.2f} us per step")
tails, afters, hosts = [], [], []
for halt, host_s, e_after in rebuilds:
    # the halted chunk: the one with s0 <= halt-1 < s0+n enqueued last before this rebuild
    cand = [r for r in log if r[0] < halt <= r[0] + r[1]]
    if not cand:
        continue
    s0, n, a, b = cand[-1]
    real = halt - s0
    tails.append(a.elapsed_time(b) * 1e3 - real * t_step)
    afters.append(b.elapsed_time(e_after) * 1e3)
    hosts.append(host_s * 1e6)
for name, v in (("no-op tail inside the halted chunk", tails), ("end of halted chunk -> rebuild kernels done", afters), ("host time of _rebuild", hosts)):
    v = np.array(v)
    print(f"  {name:44s} median {np.median(v):7.1f} us   mean {v.mean():7.1f}   p10 {np.percentile(v, 10):7.1f}   p90 {np.percentile(v, 90):7.1f}")
tot = np.array(tails) + np.array(afters)
print(f"  per rebuild (sum of the two GPU intervals)   median {np.median(tot):7.1f} us   x {len(tot)} = {tot.sum() / 1e3:.1f} ms of {1e3 * wall:.1f} ms")
print(f"  steps x median step time = {steps * t_step / 1e3:.1f} ms")
