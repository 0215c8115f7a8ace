"""Diagnostic: launch N fused steps back to back; record a GPU event every 10 launches and the host
time of every launch; report where GPU-side or host-side stalls happen."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from chiron_amd import ops, unit
from chiron_amd._device import as_tensor
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd.utils import get_list_of_mass

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
nbr.build_from_state(state)
half_dt, a, b = langevin_constants(integ.timestep, integ.collision_rate)
mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
v = torch.zeros((fluid.n_particles, 3), device="cuda")
st = _RunState(integ, pot, nbr, state.positions, v, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "fused")
st.params.skin = 1.0e6
for rep in range(3):
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N // 10 + 1)]
    host = np.zeros(N + 1)
    evs[0].record()
    t0 = time.perf_counter()
    for i in range(N):
        even, odd = (st.x_a, st.x_b) if (i & 1) == 0 else (st.x_b, st.x_a)
        ops.langevin_fused_chunk(st.params, even, odd, i, 1, 2, 1)
        host[i + 1] = time.perf_counter() - t0
        if (i + 1) % 10 == 0:
            evs[(i + 1) // 10].record()
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    gpu = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(N // 10)]) * 1e3 / 10  # us per launch
    hd = np.diff(host) * 1e6
    print(f"rep {rep}: total {1e3*total:.2f} ms; host enqueue {1e3*host[-1]:.2f} ms (max single call {hd.max():.0f} us at {hd.argmax()}); "
          f"gpu us/launch per 10-block: median {np.median(gpu):.1f} max {gpu.max():.1f} at block {gpu.argmax()}; blocks>2x median: {(gpu > 2*np.median(gpu)).sum()}")
