"""Where the fixed cost of a short LangevinIntegrator.run() goes (GPU box only): wall time per run(K) for a few K, and a
cProfile of 400 runs of 20 steps -- the driver's own bench invocation is `--steps 20 --warmup 5`.
usage: python scripts/short_run_profile.py [workload]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
state, nbr = integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
for k in (0, 1, 20, 100, 1000):
    torch.cuda.synchronize()
    reps = 300 if k <= 100 else 60
    t0 = time.perf_counter()
    for _ in range(reps):
        state, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=nbr)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"run({k:4d}): {1e6 * dt:8.1f} us per call" + (f" = {k / dt:9.0f} steps/s" if k else ""))
pr = cProfile.Profile()
pr.enable()
for _ in range(400):
    state, nbr = integ.run(state, thermo, number_of_steps=20, nbr_list=nbr)
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
