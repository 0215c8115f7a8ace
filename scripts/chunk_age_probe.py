"""GPU time per step of the real run() loop as a function of the steps since the last list rebuild (events around every
32-step C call).  GPU box only."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import integrators as I

fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
log = []
orig = I._RunState._enqueue_window


def enq(self, s0, n):
    a = torch.cuda.Event(enable_timing=True); a.record()
    sent = orig(self, s0, n)
    b = torch.cuda.Event(enable_timing=True); b.record()
    log.append((s0 - self._gen_step0, sent, n, a, b))
    return sent


I._RunState._enqueue_window = enq
state, nbr = integ.run(state, thermo, number_of_steps=20000, nbr_list=nbr)
torch.cuda.synchronize()
I._RunState._enqueue_window = orig
by = collections.defaultdict(list)
for age, sent, n, a, b in log:
    if sent == n == 32:
        by[age // 32].append(a.elapsed_time(b) * 1e3 / 32)
for k in sorted(by):
    v = np.array(by[k])
    print(f"steps {32 * k:4d}-{32 * k + 31:4d} after a rebuild: {np.median(v):5.2f} us per step (n = {len(v)})")
