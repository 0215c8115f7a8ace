"""Monte Carlo displacement trials per second at C3 with the device-resident loop (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
print('cpus:', pin_process_cpus(), file=sys.stderr)  # CHX_PIN_CPUS=0: unpinned
import bench
from chiron_amd import unit as u, mcmc

fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
nbr.build_from_state(state)
mv = mcmc.MonteCarloDisplacementMove(displacement_sigma=0.0005 * u.nanometer, number_of_moves=2000)
state, thermo, nbr = mv.update(state, thermo, nbr)
torch.cuda.synchronize(); t0 = time.perf_counter()
state, thermo, nbr = mv.update(state, thermo, nbr)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"c3 MC displacement: {2000 / dt:.0f} trials/s ({1e6 * dt / 2000:.2f} us per trial), acceptance {mv.statistics['n_accepted'] / mv.statistics['n_proposed']:.2f}")
