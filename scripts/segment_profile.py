"""Fixed cost of one run() segment + energy evaluation (what a replica-exchange cycle adds), GPU box only.
usage: python scripts/segment_profile.py [c3] [segments]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
segs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
st, nb = integ.run(state, thermo, number_of_steps=300, nbr_list=nbr)


def cycle(st, nb, steps):
    st, nb = integ.run(st, thermo, number_of_steps=steps, nbr_list=nb)
    e = pot.compute_energy(st.positions, nb)
    return st, nb, float(e)


for steps in (1, 2):
    st, nb, _ = cycle(st, nb, steps)
torch.cuda.synchronize()
for steps in (1, 33, 500):
    t0 = time.perf_counter()
    for _ in range(segs):
        st, nb, e = cycle(st, nb, steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / segs
    print(f"{wl}: cycle of {steps:4d} steps + energy: {1e6*dt:8.1f} us  ({1e6*dt/steps:7.2f} us/step)")
pr = cProfile.Profile(); pr.enable()
for _ in range(segs):
    st, nb, e = cycle(st, nb, 1)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
