"""Langevin dynamics of an argon-like LJ fluid -- the workflow of the reference's
Examples/LJ_langevin.py with ``chiron`` replaced by ``chiron_amd`` (needs the MI355X).

    python examples/lj_langevin.py [nparticles] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from chiron_amd._device import pin_process_cpus

pin_process_cpus()  # a few CPUs near the GPU for this process and the HIP runtime's threads

from chiron_amd import unit
from chiron_amd.integrators import LangevinIntegrator
from chiron_amd.neighbors import NeighborListNsqrd, OrthogonalPeriodicSpace
from chiron_amd.potential import LJPotential
from chiron_amd.reporters import LangevinDynamicsReporter
from chiron_amd.states import SamplerState, ThermodynamicState
from chiron_amd.testsystems import LennardJonesFluid  # stands in for openmmtools
from chiron_amd.utils import PRNG

nparticles = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000

fluid = LennardJonesFluid(reduced_density=0.1, nparticles=nparticles)
sigma, epsilon = 0.34 * unit.nanometer, 0.238 * unit.kilocalories_per_mole
cutoff, skin = 3.0 * sigma, 0.5 * unit.nanometer
potential = LJPotential(fluid.topology, sigma=sigma, epsilon=epsilon, cutoff=cutoff)

PRNG.set_seed(1234)
state = SamplerState(positions=fluid.positions, current_PRNG_key=PRNG.get_random_key(),
                     box_vectors=fluid.system.getDefaultPeriodicBoxVectors())
thermo = ThermodynamicState(potential=potential, temperature=300 * unit.kelvin)

# same class name as the reference; the build runs the O(N) cell kernel whenever the box allows
nbr_list = NeighborListNsqrd(OrthogonalPeriodicSpace(), cutoff=cutoff, skin=skin, n_max_neighbors=180)
nbr_list.build_from_state(state)
print("initial energy [kJ/mol]:", float(potential.compute_energy(state.positions, nbr_list)))

reporter = LangevinDynamicsReporter("lj_langevin", 1, fluid.topology)
integrator = LangevinIntegrator(reporter=reporter, report_interval=100)
torch.cuda.synchronize()
t0 = time.perf_counter()
state, nbr_list = integrator.run(state, thermo, number_of_steps=steps, nbr_list=nbr_list)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} steps of N={nparticles} in {1e3 * dt:.1f} ms ({steps / dt:.0f} steps/s), "
      f"{integrator.n_rebuilds} list rebuilds")
print("reported potential energies:", reporter.get_property("potential_energy"))
