"""Diagnostic: does running c3 first slow a later c4 run in the same process?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench

def run(wl, steps, label):
    fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
    state, nbr = integ.run(state, thermo, number_of_steps=100, nbr_list=nbr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {wl} {steps} steps {1e3*dt:.2f} ms ({1e6*dt/steps:.1f} us/step) rebuilds {integ.n_rebuilds} rebuild_s {getattr(integ,'rebuild_seconds',0):.4f}", flush=True)
    return fluid, pot, state, thermo, nbr, integ

a = run("c4", 1000, "c4 alone")
del a
b = run("c3", 3000, "c3")
c = run("c4", 1000, "c4 after c3 (c3 objects alive)")
del b, c
d = run("c4", 1000, "c4 again")
