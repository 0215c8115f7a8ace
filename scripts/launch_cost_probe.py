"""Host cost of one step launch vs its GPU time: is the Langevin loop bound by the host's launch rate?  GPU box only.
Enqueues N launches in ONE C call on an idle stream and reads the host time of the call and the GPU completion time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import ops, unit
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd.utils import get_list_of_mass
from chiron_amd._device import as_tensor

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
nbr.build_from_state(state)
half_dt, a, b = langevin_constants(1.0 * unit.femtoseconds, 1.0 / unit.picoseconds)
mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
st = _RunState(integ, pot, nbr, state.positions, state.velocities, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "fused")
st.params.skin = 1e6
step = [0]


def call(n):
    even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
    ops.langevin_fused_chunk(st.params, even, odd, step[0], n, 2, 1)
    step[0] += n


t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    call(64)
    torch.cuda.synchronize()
for n in (64, 256, 1024):
    host, total = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) / n * 1e6)
        total.append((t2 - t0) / n * 1e6)
    print(f"{wl}: {n:5d} launches per call: host {np.median(host):5.2f} us per launch, until the GPU is done {np.median(total):5.2f} us per launch")


def call_window(n, window):
    even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
    st._gen.host[0] = step[0] - 1
    sent = ops.langevin_fused_run(st.params, even, odd, step[0], n, 2, window)
    assert sent == n
    step[0] += n


if st._windowed():
    for window in (4, 8, 12, 16, 32, 64):
        total = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            call_window(1024, window)
            torch.cuda.synchronize()
            total.append((time.perf_counter() - t0) / 1024 * 1e6)
        print(f"{wl}: bounded launch queue, window {window:3d}: {np.median(total):5.2f} us per launch")
    def call_window2(n, window):  # (progress word left alone: as the integrator's loop does between rebuilds)
        even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
        sent = ops.langevin_fused_run(st.params, even, odd, step[0], n, 2, window)
        assert sent == n
        step[0] += n

    for chunk in (1024, 256, 64, 32, 8):
        total = []
        for _ in range(10):
            torch.cuda.synchronize()
            st._gen.host[0] = step[0] - 1
            t0 = time.perf_counter()
            for _ in range(1024 // chunk):
                call_window2(chunk, 12)
            torch.cuda.synchronize()
            total.append((time.perf_counter() - t0) / 1024 * 1e6)
        print(f"{wl}: window 12, one C call per {chunk} launches from a Python loop: {np.median(total):5.2f} us per launch")
    # the real loop, no rebuild possible (huge skin): run() segments
    st.params.skin = 1e6
