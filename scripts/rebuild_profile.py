"""Where a list rebuild's wall time goes inside LangevinIntegrator.run (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import integrators

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, int(os.environ.get("CHUNK", "32")), "cell")
out = integ.run(state, thermo, number_of_steps=300, nbr_list=nbr)

T = {"drain": 0.0, "unpack": 0.0, "build": 0.0, "load": 0.0, "n": 0}
RS = integrators._RunState
orig_rebuild = RS._rebuild

def timed_rebuild(self, step=0, speculative=False):
    if speculative and os.environ.get("CHX_SYNC_REBUILD") != "1":
        t0 = time.perf_counter()
        orig_rebuild(self, step, speculative=True)  # nothing waits here: host enqueue time only
        T["build"] += time.perf_counter() - t0; T["n"] += 1
        return
    t0 = time.perf_counter()
    x_user, v_user = self.positions_user(), self.velocities_user()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    self.nbr_list._adopt_positions = True
    self.nbr_list.build(x_user, self.nbr_list.box_vectors)
    self.nbr_list._adopt_positions = False
    torch.cuda.synchronize(); t2 = time.perf_counter()
    self._load(x_user, v_user, step)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    self.n_rebuilds += 1
    T["unpack"] += t1 - t0; T["build"] += t2 - t1; T["load"] += t3 - t2; T["n"] += 1

RS._rebuild = timed_rebuild
orig_sync = torch.cuda.synchronize
torch.cuda.synchronize(); t0 = time.perf_counter()
out = integ.run(out[0], thermo, number_of_steps=steps, nbr_list=out[1])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = max(T["n"], 1)
print(f"{wl}: {steps} steps in {1e3*dt:.2f} ms ({steps/dt:.0f} steps/s), rebuilds {T['n']}")
print(f"  per rebuild: unpack {1e6*T['unpack']/n:.0f} us, build {1e6*T['build']/n:.0f} us, load {1e6*T['load']/n:.0f} us"
      f"  -> {1e6*(T['unpack']+T['build']+T['load'])/n:.0f} us; all rebuilds = {100*(T['unpack']+T['build']+T['load'])/dt:.1f}% of the run")
print(f"  inner list: {getattr(integ, 'prune_passes', 0)} pruning passes, {getattr(integ, 'full_list_launches', 0)} launches on the full list")
