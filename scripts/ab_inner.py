"""Steps/s as a function of the pruned inner list's settings (CHX_INNER_SKIN [nm], CHX_PRUNE_EVERY [steps]) at the
loaded clock: one process, a 3 s warm-up, median of three windows per setting, the whole grid twice.  GPU box only.
usage: python scripts/ab_inner.py c3 20000 0.08,0.12,0.16 16,32,64"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
skins = (sys.argv[3] if len(sys.argv) > 3 else "0.08,0.12,0.16").split(",")
everys = (sys.argv[4] if len(sys.argv) > 4 else "16,32,64").split(",")
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
for rep in range(2):
    for sk in skins:
        for ev in everys:
            os.environ["CHX_INNER_SKIN"], os.environ["CHX_PRUNE_EVERY"] = sk, ev
            res = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
                torch.cuda.synchronize(); res.append(steps / (time.perf_counter() - t0))
            print(f"{wl} inner_skin={sk} every={ev}: {sorted(res)[1]:.0f} steps/s {[round(r) for r in res]} full-list launches {getattr(integ, 'full_list_launches', 0)}", flush=True)
            integ.full_list_launches = 0
