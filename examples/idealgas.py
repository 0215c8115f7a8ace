"""NPT ideal gas: barostat + displacement moves, volume statistics against the analytic
expectation -- the workflow of the reference's Examples/Idealgas.py (needs the MI355X).

    python examples/idealgas.py [iterations]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from chiron_amd import unit
from chiron_amd._device import pin_process_cpus
from chiron_amd.mcmc import MCMCSampler, MonteCarloBarostatMove, MonteCarloDisplacementMove, MoveSchedule
from chiron_amd.neighbors import OrthogonalPeriodicSpace, PairListNsqrd
from chiron_amd.potential import IdealGasPotential
from chiron_amd.reporters import MCReporter
from chiron_amd.states import SamplerState, ThermodynamicState
from chiron_amd.testsystems import IdealGas
from chiron_amd.utils import PRNG

pin_process_cpus()  # a few CPUs near the GPU for this process and the HIP runtime's threads

iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gas = IdealGas(nparticles=216, temperature=298 * unit.kelvin, pressure=1 * unit.atmosphere)
potential = IdealGasPotential(gas.topology)
thermo = ThermodynamicState(potential=potential, temperature=298 * unit.kelvin, pressure=1 * unit.atmosphere)
PRNG.set_seed(1234)
state = SamplerState(positions=gas.positions, current_PRNG_key=PRNG.get_random_key(),
                     box_vectors=gas.system.getDefaultPeriodicBoxVectors())
pairs = PairListNsqrd(OrthogonalPeriodicSpace(), cutoff=0 * unit.nanometer)  # wraps particles, no interactions
pairs.build_from_state(state)

rep_baro = MCReporter("ideal_baro", 1)
schedule = MoveSchedule([
    ("MonteCarloDisplacementMove", MonteCarloDisplacementMove(displacement_sigma=0.1 * unit.nanometer, number_of_moves=100)),
    ("MonteCarloBarostatMove", MonteCarloBarostatMove(volume_max_scale=0.2, number_of_moves=10, reporter=rep_baro,
                                                      report_interval=1)),
])
MCMCSampler(schedule).run(state, thermo, n_iterations=iterations, nbr_list=pairs)
volume = rep_baro.get_property("volume")
expected = gas.get_volume_expectation(thermo).value_in_unit(unit.nanometer ** 3)
spread = gas.get_volume_standard_deviation(thermo).value_in_unit(unit.nanometer ** 3)
print(f"<V> = {np.mean(volume):.1f} nm^3 (ideal gas law {expected:.1f}), sigma_V = {np.std(volume):.1f} (expected {spread:.1f})")
