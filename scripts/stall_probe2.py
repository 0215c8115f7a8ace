"""Diagnostic: fused-step launches in chunks of 32 with the per-chunk 16-byte read-back, three variants:
 A: async D2H copy into pinned memory + event (what integrators.py does)
 B: no read-back at all (event only)
 C: read-back by a device->pinned store from a tiny kernel is not available here; instead .cpu() sync per chunk"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from chiron_amd import ops, unit
from chiron_amd._device import as_tensor
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd.utils import get_list_of_mass

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
NCH = int(sys.argv[2]) if len(sys.argv) > 2 else 60
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
nbr.build_from_state(state)
half_dt, a, b = langevin_constants(integ.timestep, integ.collision_rate)
mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
v = torch.zeros((fluid.n_particles, 3), device="cuda")
st = _RunState(integ, pot, nbr, state.positions, v, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "fused")
st.params.skin = 1.0e6
ring = [torch.empty((4,), dtype=torch.int32, pin_memory=True) for _ in range(4)]

def run(variant):
    torch.cuda.synchronize()
    pend = collections.deque()
    waits = []
    t0 = time.perf_counter()
    step = 0
    for c in range(NCH):
        ops.langevin_fused_chunk(st.params, st.x_a, st.x_b, step, 32, 2, 1)
        step += 32
        if variant == "A":
            ring[c % 4].copy_(st.ctrl, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(); pend.append(ev)
        if len(pend) >= 2:
            e = pend.popleft(); tw = time.perf_counter(); e.synchronize(); waits.append(time.perf_counter() - tw)
            if variant == "C":
                st.ctrl.cpu()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w = np.array(waits) * 1e3
    print(f"variant {variant}: {NCH*32} steps {1e3*dt:.2f} ms ({1e6*dt/(NCH*32):.1f} us/step); chunk waits ms: median {np.median(w):.2f} max {w.max():.2f} (>3x median: {(w > 3*np.median(w)).sum()})", flush=True)

for rep in range(3):
    for variant in ("A", "B", "C"):
        run(variant)
