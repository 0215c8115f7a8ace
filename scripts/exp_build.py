"""GPU time of chx_nl_build_cell_fl (all its launches) for the timing decomposition of scripts/exp_build.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import ops

for wl in sys.argv[1:] or ["c3", "c4"]:
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    nbr.build_from_state(state)
    warm = int(os.environ.get("EXP_WARM", "0"))  # equilibrate first: the clustered fluid the bench's numbers are quoted on
    if warm:
        state, nbr = integ.run(state, thermo, number_of_steps=warm, nbr_list=nbr)
        torch.cuda.synchronize()
    half = os.environ.get("EXP_HALF_ROWS", "1") != "0"  # 0: the integrator's rebuilds (half rows counted, not written)
    x = state.positions
    n = x.shape[0]
    cs = nbr._cs_md
    ncell, nc = ops.cell_grid(nbr._box_h, cs, n)
    g, chunks = nbr._fl_geometry(n)
    scratch = {}
    m = int(nbr.n_max_neighbors)
    for _ in range(3):
        ops.nl_build_cell_fl(x, nbr._box_h, cs, m, ncell, nc, g, chunks, scratch, half_rows=half)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        ops.nl_build_cell_fl(x, nbr._box_h, cs, m, ncell, nc, g, chunks, scratch, half_rows=half)
    e1.record()
    torch.cuda.synchronize()
    print(f"{wl}: chx_nl_build_cell_fl {1e3 * e0.elapsed_time(e1) / reps:8.1f} us per call (g={g}, chunks={chunks}, m={m}, half_rows={half}, warm={warm})")
