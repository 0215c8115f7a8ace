"""Inner-list row lengths in the equilibrated C3 fluid (GPU box only): how many rows need a second chunk?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chiron_amd import ops, unit
from chiron_amd._device import as_tensor
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd.utils import get_list_of_mass

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
for warm in (0, 2000, 20000):
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    if warm:
        state, nbr = integ.run(state, thermo, number_of_steps=warm, nbr_list=nbr)
    else:
        nbr.build_from_state(state)
    half_dt, a, b = langevin_constants(1.0 * unit.femtoseconds, 1.0 / unit.picoseconds)
    mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
    v = state.velocities if state.velocities is not None else torch.zeros_like(state.positions)
    st = _RunState(integ, pot, nbr, state.positions, v, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "fused")
    st.params.skin = 1e6
    ops.langevin_fused_chunk(st.params, st.x_a, st.x_b, 0, 2, 2, 1)  # step 0 refreshes the inner list
    torch.cuda.synchronize()
    fl = nbr.force_list
    g = fl.g
    W = 4 * g
    degi = st.deg_inner.cpu().numpy() if hasattr(st, "deg_inner") else None
    if degi is None:
        for name in dir(st):
            if "deg" in name and "inner" in name:
                degi = getattr(st, name).cpu().numpy()
    deg = fl.deg.cpu().numpy()
    ppb = 256 // g
    nblk = len(degi) // ppb
    per_blk = degi[: nblk * ppb].reshape(nblk, ppb).max(axis=1)
    print(f"{wl} after {warm} steps: G={g} W={W}  full list mean {deg.mean():.1f} max {deg.max()}  inner mean {degi.mean():.1f} "
          f"p99 {np.percentile(degi, 99):.0f} max {degi.max()};  rows > W: {100 * (degi > W).mean():.2f} %, "
          f"blocks with such a row: {100 * (per_blk > W).mean():.1f} %, rows > W/2: {100 * (degi > W // 2).mean():.1f} %")
