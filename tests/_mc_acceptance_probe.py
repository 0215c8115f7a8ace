"""Acceptance statistics of the test-sized MC systems for a few displacement sigmas (GPU box only; a helper for
choosing the parameters of the Monte Carlo tests -- it uses the oracle's lattice generator, so it lives under tests/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import chiron_amd as ca
from chiron_amd import unit as u, mcmc, neighbors, states, potential
from oracle import chiron_oracle as O
from chiron_amd.topology import Topology

for skin, sd, moves in ((0.5, 0.002, 60), (0.5, 0.01, 60), (0.5, 0.03, 60), (0.05, 0.004, 80), (0.05, 0.01, 80), (0.5, 0.05, 40)):
    n = 500
    x, box = O.lj_lattice(n, 0.1, 0.34, seed=71)
    pot = potential.LJPotential(Topology.homogeneous(n, "Ar"), sigma=0.34 * u.nanometer, epsilon=0.238 * u.kilocalories_per_mole, cutoff=1.02 * u.nanometer)
    st = states.SamplerState(u.Quantity(x, u.nanometer), current_PRNG_key=ca.rng.PRNGKey(5, 6), box_vectors=u.Quantity(box, u.nanometer))
    th = states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    nb = neighbors.NeighborListCell(neighbors.OrthogonalPeriodicSpace(), cutoff=1.02 * u.nanometer, skin=skin * u.nanometer, n_max_neighbors=180)
    nb.build_from_state(st)
    mv = mcmc.MonteCarloDisplacementMove(displacement_sigma=sd * u.nanometer, number_of_moves=moves)
    s2, _, n2 = mv.update(st, th, nb)
    print(skin, sd, moves, mv.statistics, "rebuilt" if not torch.equal(n2.ref_positions, nb.ref_positions) else "same list", flush=True)
