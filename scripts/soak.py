"""Long-run consistency check (GPU box): the same trajectory with speculative and with synchronous rebuilds,
and with two launch batchings, must end in identical bits; reports rebuild counts and the kinetic temperature.
usage: python scripts/soak.py [c3] [steps]     (run once per mode: CHX_SYNC_REBUILD=0/1, CHUNK=16/32)"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, int(os.environ.get("CHUNK", "32")), "cell")
t0 = time.perf_counter()
out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
x, v = out.positions.cpu().numpy(), out.velocities.cpu().numpy()
kT = 0.00831446261815324 * 300.0
h = hashlib.sha256(x.tobytes() + v.tobytes()).hexdigest()[:16]
print(f"{wl} steps={steps} sync={os.environ.get('CHX_SYNC_REBUILD','0')} chunk={os.environ.get('CHUNK','32')}: "
      f"{steps/dt:.0f} steps/s, rebuilds {integ.n_rebuilds}, refreshes {getattr(integ,'prune_passes',0)}, "
      f"full-list launches {getattr(integ,'full_list_launches',0)}, T_kin/T = {(39.948*v**2).mean()/kT:.4f}, "
      f"E = {float(pot.compute_energy(out.positions, nbr)):.3f}, hash {h}")
