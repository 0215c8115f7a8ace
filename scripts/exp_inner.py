"""Feasibility probe for a dynamically pruned inner list: fused-step kernel time on an EVOLVED
configuration as a function of the list skin (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import unit, ops
from chiron_amd.neighbors import NeighborListCell, OrthogonalPeriodicSpace
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd.utils import get_list_of_mass
from chiron_amd._device import as_tensor

def ev_time(fn, reps=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

for wl in sys.argv[1:] or ["c3", "c4"]:
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    out = integ.run(state, thermo, number_of_steps=600, nbr_list=nbr)
    st0 = out[0]
    n = fluid.n_particles
    mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
    half_dt, a, b = langevin_constants(1.0 * unit.femtoseconds, 1.0 / unit.picoseconds)
    for skin in (0.5, 0.2, 0.12, 0.08):
        nb = NeighborListCell(OrthogonalPeriodicSpace(), cutoff=3 * 0.34 * unit.nanometer, skin=skin * unit.nanometer, n_max_neighbors=180)
        nb.build(st0.positions, st0.box_vectors)
        p_nl = int(nb.n_neighbors.sum())
        rs = _RunState(integ, pot, nb, st0.positions, st0.velocities, mass, 2.494, half_dt, a, b, st0.current_PRNG_key, None, "fused")
        rs.params.skin = 1e6
        step = [0]
        def one():
            # 8 launches per call from C: the measurement is not bound by the Python launch rate
            even, odd = (rs.x_a, rs.x_b) if (step[0] & 1) == 0 else (rs.x_b, rs.x_a)
            ops.langevin_fused_chunk(rs.params, even, odd, step[0], 8, 2, 1)
            step[0] += 8
        t = ev_time(one) / 8
        print(f"{wl} skin={skin}: P_nl={p_nl} ({p_nl/n:.1f}/particle) g={rs.fl.g} chunks={rs.fl.chunks} fused step {t:.2f} us", flush=True)
