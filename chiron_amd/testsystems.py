"""Synthetic test systems standing in for ``openmmtools.testsystems`` (not installed).

Each object exposes what the reference's examples/tests read from an openmmtools
system: ``positions`` (Quantity, nm), ``topology`` and ``box_vectors`` (Quantity 3x3),
plus ``system.getDefaultPeriodicBoxVectors()``.  Geometry follows SURVEY.md section 8d:
jittered simple-cubic lattice, cubic box of edge (N sigma^3 / rho*)^(1/3), seed 1234.
"""
from __future__ import annotations

import math

import numpy as np

from . import unit
from .topology import Topology


class _System:
    def __init__(self, box_vectors):
        self._box = box_vectors

    def getDefaultPeriodicBoxVectors(self):
        return self._box


def lj_lattice(n: int, reduced_density: float, sigma_nm: float, seed: int = 1234, jitter: float = 0.1):
    """(positions fp32 (n,3) nm, box fp32 (3,3) nm) -- same generator as the oracle's."""
    L = (n * sigma_nm ** 3 / reduced_density) ** (1.0 / 3.0)
    m = int(math.ceil(n ** (1.0 / 3.0) - 1e-9))
    a = L / m
    idx = np.arange(n)
    ix, iy, iz = idx % m, (idx // m) % m, idx // (m * m)
    rng = np.random.default_rng(seed)
    pos = (np.stack([ix, iy, iz], axis=1) + 0.5) * a + rng.uniform(-jitter, jitter, size=(n, 3)) * a
    pos = pos - np.floor(pos / L) * L
    return pos.astype(np.float32), np.diag([L, L, L]).astype(np.float32)


class LennardJonesFluid:
    """Argon-like LJ fluid (defaults of openmmtools: sigma 3.4 A, eps 0.238 kcal/mol)."""

    def __init__(self, nparticles: int = 1000, reduced_density: float = 0.05,
                 sigma=3.4 * unit.angstrom, epsilon=0.238 * unit.kilocalories_per_mole,
                 seed: int = 1234, symbol: str = "Ar", **_ignored):
        self.sigma, self.epsilon = sigma, epsilon
        s = float(sigma.value_in_unit(unit.nanometer))
        pos, box = lj_lattice(nparticles, reduced_density, s, seed)
        self.positions = unit.Quantity(pos, unit.nanometer)
        self.box_vectors = unit.Quantity(box, unit.nanometer)
        self.topology = Topology.homogeneous(nparticles, symbol)
        self.system = _System(self.box_vectors)
        self.n_particles = nparticles


class IdealGas:
    """N non-interacting particles at (T, p): cubic box of volume N kT / p."""

    def __init__(self, nparticles: int = 216, temperature=298.0 * unit.kelvin, pressure=1.0 * unit.atmosphere,
                 seed: int = 1234, symbol: str = "Ar"):
        self.temperature, self.pressure = temperature, pressure
        self.n_particles = nparticles
        kT = unit.BOLTZMANN_CONSTANT_kB * temperature
        volume = (nparticles * kT / pressure).value_in_unit(unit.nanometer ** 3)
        L = float(volume) ** (1.0 / 3.0)
        rng = np.random.default_rng(seed)
        self.positions = unit.Quantity(rng.uniform(0.0, L, size=(nparticles, 3)).astype(np.float32), unit.nanometer)
        self.box_vectors = unit.Quantity(np.diag([L, L, L]).astype(np.float32), unit.nanometer)
        self.topology = Topology.homogeneous(nparticles, symbol)
        self.system = _System(self.box_vectors)

    def get_volume_expectation(self, thermodynamic_state=None):
        """<V> = (N+1) kT / p for the ideal-gas NPT ensemble (openmmtools convention)."""
        kT = unit.BOLTZMANN_CONSTANT_kB * self.temperature
        return ((self.n_particles + 1) * kT / self.pressure).in_units_of(unit.nanometer ** 3)

    def get_volume_standard_deviation(self, thermodynamic_state=None):
        kT = unit.BOLTZMANN_CONSTANT_kB * self.temperature
        return (math.sqrt(self.n_particles + 1) * kT / self.pressure).in_units_of(unit.nanometer ** 3)
