"""Host time of the pieces of a list rebuild inside run() (GPU box only, diagnostic)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import integrators as I, ops, neighbors as NB

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
acc = collections.defaultdict(lambda: [0.0, 0])


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[label]
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(obj, name, g)


fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
wrap(I._RunState, "_rebuild")
wrap(I._RunState, "positions_user")
wrap(I._RunState, "velocities_user")
wrap(I._RunState, "_load")
wrap(I._RunState, "_open_generation")
wrap(I._RunState, "_close_generation")
wrap(I._RunState, "_inner_list_settings")
wrap(type(nbr), "_build_speculative")
wrap(ops, "nl_build_cell_fl", "ops.nl_build_cell_fl")
wrap(ops, "cell_grid", "ops.cell_grid")
wrap(ops, "publish", "ops.publish")
wrap(ops, "pack_langevin", "ops.pack_langevin")
wrap(ops, "ctrl_reset", "ops.ctrl_reset")
wrap(ops._lib.load(), "chx_nl_build_cell_fl", "C chx_nl_build_cell_fl")
wrap(ops._lib.load(), "chx_pack_langevin", "C chx_pack_langevin")
torch.cuda.synchronize(); t0 = time.perf_counter()
state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"{wl}: {steps / wall:.0f} steps/s with the wrappers, {integ.n_rebuilds} rebuilds")
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:28s} {1e6 * t / max(n, 1):8.1f} us x {n}")
