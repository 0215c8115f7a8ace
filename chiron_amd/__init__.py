"""chiron_amd: MI355X (gfx950) implementation of chiron's Lennard-Jones /
neighbor-list / Langevin / Metropolis hot path behind chiron's own class API.

Modules mirror the reference's flat layout (``chiron.potential`` -> ``chiron_amd.potential``
and so on); the numerics live in ``csrc/*.hip`` behind the C ABI of ``include/chx.h``.
"""
from . import unit  # noqa: F401

__all__ = [
    "unit", "topology", "states", "utils", "neighbors", "potential", "integrators", "mcmc",
    "multistate", "reporters", "testsystems", "build",
]
