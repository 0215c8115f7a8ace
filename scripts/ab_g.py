"""Steps/s as a function of the lanes-per-particle override (CHX_FL_G), measured at the loaded clock: one process, a
3 s clock warm-up, then for every G a fresh system and the median of three windows.  GPU box only.
usage: python scripts/ab_g.py c4 4000 1,2,4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
gs = (sys.argv[3] if len(sys.argv) > 3 else "1,2,4").split(",")


def make(g):
    if g == "auto":
        os.environ.pop("CHX_FL_G", None)
    else:
        os.environ["CHX_FL_G"] = g
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
    state, nbr = integ.run(state, thermo, number_of_steps=1000, nbr_list=nbr)
    return state, thermo, nbr, integ


state, thermo, nbr, integ = make(gs[0])
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
for rep in range(2):
    for g in gs:
        state, thermo, nbr, integ = make(g)
        res = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
            torch.cuda.synchronize(); res.append(steps / (time.perf_counter() - t0))
        fl = nbr.force_list
        print(f"{wl} G={g} (list g={fl.g}, chunks={fl.chunks}): {sorted(res)[1]:.0f} steps/s  {[round(r) for r in res]}  rebuilds {integ.n_rebuilds}", flush=True)
