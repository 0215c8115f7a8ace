"""Upper bound of a wider INTERNAL skin (round-3 review item 5): steady-state steps/s of the halting scheme with the
Verlet skin itself widened -- the kernels and the rebuild chain a force list of radius cutoff + skin_int would cost,
WITHOUT the decoupled step kernel's extra loads (snap4) that a wider internal skin under the reference's 0.5 nm would
add on top.  If the wider list does not pay here it cannot pay there.  GPU box only.
usage: python scripts/skin_sweep.py c3 20000 0.5,0.6,0.7,0.8,0.9"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench
from chiron_amd import unit

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
skins = [float(s) for s in (sys.argv[3] if len(sys.argv) > 3 else "0.5,0.6,0.7,0.8,0.9").split(",")]
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
state, nbr = integ.run(state, thermo, number_of_steps=20000 if wl == "c3" else 8000, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
for rep in range(2):
    for skin in skins:
        nb, ig = bench.make_list_and_integrator(wl)
        nb.skin = skin * unit.nanometer
        nb.n_max_neighbors = 400
        st, nb = ig.run(state, thermo, number_of_steps=2000, nbr_list=nb)
        torch.cuda.synchronize()
        res = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st, nb = ig.run(st, thermo, number_of_steps=steps, nbr_list=nb)
            torch.cuda.synchronize(); res.append((steps / (time.perf_counter() - t0), ig.n_rebuilds))
        res.sort()
        print(f"{wl} skin {skin:.2f} nm: median {res[1][0]:.0f} steps/s, {res[1][1]} rebuilds / {steps} steps, "
              f"list pairs {int(nb.n_neighbors.sum())}, windows {[round(r[0]) for r in res]}", flush=True)
