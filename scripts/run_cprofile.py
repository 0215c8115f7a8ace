"""Where the Python time of a short run() goes (GPU box only): cProfile over many run(K) calls, top functions by own time.
usage: python scripts/run_cprofile.py [K] [workload] [calls]     (CHX_LIST_CARRY=0: build on every entry)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
for _ in range(200):
    state, nbr = integ.run(state, thermo, number_of_steps=K, nbr_list=nbr)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    state, nbr = integ.run(state, thermo, number_of_steps=K, nbr_list=nbr)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"{calls} x run({K}) at {wl}: own time per call in us (profiler overhead included), top 45")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:45]
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print(f"  {1e6 * tt / calls:7.2f} us own  {1e6 * ct / calls:7.2f} us cum  {nc / calls:6.1f} calls  {os.path.basename(fn)}:{line} {name}")
