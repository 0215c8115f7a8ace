"""cProfile of LangevinIntegrator.run on the GPU box (diagnostic): where does host time go?"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
state, nbr = integ.run(state, thermo, number_of_steps=100, nbr_list=nbr)
torch.cuda.synchronize()
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    torch.cuda.synchronize()
    pr.disable()
    print(f"rep {rep}: {steps} steps in {1e3*(time.perf_counter()-t0):.2f} ms, rebuilds {integ.n_rebuilds}")
    if rep == 1:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
