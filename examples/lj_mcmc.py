"""NPT Monte Carlo of united-atom methane (UA-TraPPE): Langevin + displacement + barostat moves,
the schedule of the reference's Examples/LJ_MCMC.py (needs the MI355X).

    python examples/lj_mcmc.py [iterations]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from chiron_amd import unit
from chiron_amd._device import pin_process_cpus
from chiron_amd.mcmc import (LangevinDynamicsMove, MCMCSampler, MonteCarloBarostatMove,
                             MonteCarloDisplacementMove, MoveSchedule)
from chiron_amd.neighbors import NeighborListNsqrd, OrthogonalPeriodicSpace
from chiron_amd.potential import LJPotential
from chiron_amd.reporters import LangevinDynamicsReporter, MCReporter
from chiron_amd.states import SamplerState, ThermodynamicState
from chiron_amd.topology import Element, Topology
from chiron_amd.utils import PRNG

pin_process_cpus()  # a few CPUs near the GPU for this process and the HIP runtime's threads

iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_particles = 1100
topology = Topology()
bead = Element(1000, "CH4", "CH4", unit.Quantity(16.04, unit.gram / unit.mole))
chain = topology.addChain()
for _ in range(n_particles):
    topology.addAtom("CH4", bead, topology.addResidue("CH4", chain))

positions = np.load(os.path.join(ROOT, "tests", "golden", "methane_coords.npy")) * unit.nanometer
edge = 4.275021399280942
box_vectors = np.diag([edge, edge, edge]) * unit.nanometer

cutoff = 1.4 * unit.nanometer
potential = LJPotential(topology, sigma=0.373 * unit.nanometer, epsilon=0.2941 * unit.kilocalories_per_mole, cutoff=cutoff)
thermo = ThermodynamicState(potential=potential, temperature=140 * unit.kelvin, pressure=13.00765 * unit.atmosphere)
PRNG.set_seed(1234)
state = SamplerState(positions=positions, current_PRNG_key=PRNG.get_random_key(), box_vectors=box_vectors)
nbr_list = NeighborListNsqrd(OrthogonalPeriodicSpace(), cutoff=cutoff, skin=0.3 * unit.nanometer, n_max_neighbors=500)
nbr_list.build_from_state(state)

rep_disp, rep_baro, rep_md = MCReporter("mc_disp", 10), MCReporter("mc_baro", 1), LangevinDynamicsReporter("md", 10)
schedule = MoveSchedule([
    ("LangevinDynamicsMove", LangevinDynamicsMove(timestep=1.0 * unit.femtoseconds, collision_rate=1.0 / unit.picoseconds,
                                                  number_of_steps=1000, reporter=rep_md, report_interval=10)),
    ("MonteCarloDisplacementMove", MonteCarloDisplacementMove(displacement_sigma=0.001 * unit.nanometer, number_of_moves=100,
                                                              reporter=rep_disp, report_interval=10, autotune=True,
                                                              autotune_interval=100)),
    ("MonteCarloBarostatMove", MonteCarloBarostatMove(volume_max_scale=0.1, number_of_moves=10, reporter=rep_baro,
                                                      report_interval=1, autotune=True, autotune_interval=50)),
])
state, thermo, nbr_list = MCMCSampler(schedule).run(state, thermo, n_iterations=iterations, nbr_list=nbr_list)

volume = rep_baro.get_property("volume")
density = 16.04 * n_particles / (volume * 6.02214076e23 * 1e-27) * 1e-3  # g/mol / nm^3 -> kg/m^3
print(f"{iterations} iterations: <V> = {volume.mean():.3f} nm^3, density = {density.mean():.1f} kg/m^3 "
      f"(UA-TraPPE methane at 140 K / 13 atm: ~372 kg/m^3 once equilibrated)")
for name, move in schedule.move_schedule[1:]:
    print(name, move.statistics)
