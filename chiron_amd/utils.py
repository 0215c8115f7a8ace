"""PRNG bookkeeping, masses and Maxwell-Boltzmann velocities.

Reference: chiron/utils.py:6-38 (PRNG), :101-113 (topology helpers),
:116-144 (initialize_velocities).
"""
from __future__ import annotations

import numpy as np

from . import unit
from .rng import JaxKey, PRNGKey, split


class PRNG:
    """Process-global seed -> stream of independent keys (chiron/utils.py:6-38)."""

    _key: PRNGKey = None
    _seed: int = None

    @classmethod
    def set_seed(cls, seed: int, stream: str = "philox") -> None:
        """``stream="threefry"`` makes every key handed out a ``JaxKey``: splits, uniforms and normals then follow
        ``jax.random`` (chiron/utils.py:29-38), reproducing the reference's draws (chiron_amd/rng.py)."""
        if stream not in ("philox", "threefry"):
            raise ValueError(f"unknown PRNG stream {stream!r} (philox | threefry)")
        cls._seed = seed
        cls._key = JaxKey.from_seed(seed) if stream == "threefry" else PRNGKey.from_seed(seed)

    @classmethod
    def get_random_key(cls) -> PRNGKey:
        key, subkey = split(cls._key)
        cls._key = key
        return subkey


def get_nr_of_particles(topology) -> int:
    """chiron/utils.py:101-103"""
    return topology.getNumAtoms()


def get_list_of_mass(topology) -> unit.Quantity:
    """Masses (amu) from ``atom.element.mass`` (chiron/utils.py:106-113)."""
    if hasattr(topology, "masses_amu"):
        return topology.masses_amu() * unit.amu  # cached: the reference re-walks the atoms on every run()
    mass = [atom.element.mass.value_in_unit(unit.amu) for atom in topology.atoms()]
    return np.asarray(mass, dtype=np.float64) * unit.amu


def initialize_velocities(temperature: unit.Quantity, topology, key: PRNGKey) -> unit.Quantity:
    """v0 = sqrt(kT/m) * N(0,1) per component, in nm/ps (chiron/utils.py:116-144).

    The normals come from the device Philox stream ``STREAM_VELOCITY`` keyed by
    ``key`` (HIP kernel ``chx_init_velocities``).
    """
    from . import ops
    from ._device import as_tensor

    mass = get_list_of_mass(topology)
    kB = unit.BOLTZMANN_CONSTANT_kB * unit.AVOGADRO_CONSTANT_NA
    kbT_unitless = (kB * temperature).value_in_unit_system(unit.md_unit_system)
    mass_unitless = as_tensor(mass.value_in_unit_system(unit.md_unit_system))
    normals = None
    if isinstance(key, JaxKey):  # random.normal(key, [len(mass), 3]), chiron/utils.py:142
        normals = ops.jax_normal(key, (mass_unitless.shape[0], 3), device=mass_unitless.device)
    v0 = ops.init_velocities(mass_unitless, float(kbT_unitless), key, normals)
    return v0 * (unit.nanometer / unit.picosecond)
