import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench

def timed_run(integ, state, thermo, nbr, steps, label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {steps} steps {1e3*dt:.2f} ms ({1e6*dt/steps:.1f} us/step) rebuilds {integ.n_rebuilds} "
          f"rebuild_ms {1e3*getattr(integ, 'rebuild_seconds', 0.0):.2f} gc {gc.get_count()} {[s['collections'] for s in gc.get_stats()]}", flush=True)
    integ.rebuild_seconds = 0.0
    return state, nbr

fluid4, pot4, state4, thermo4, nbr4, integ4 = bench.make_system("c4", 300.0, 32, "cell")
nbr4.build_from_state(state4)
if "--tk" in sys.argv:
    print(bench.time_kernels(fluid4, pot4, state4, nbr4, integ4, 200)[:3])
state4, nbr4 = timed_run(integ4, state4, thermo4, nbr4, 100, "warmup")
state4, nbr4 = timed_run(integ4, state4, thermo4, nbr4, 1000, "run A")
state4, nbr4 = timed_run(integ4, state4, thermo4, nbr4, 1000, "run B")
