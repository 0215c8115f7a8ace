"""Potentials: same classes and call signatures as chiron/potential.py.

``LJPotential.compute_energy`` / ``compute_force`` (reference :215-300) dispatch to
the HIP pair kernels in ``csrc/chx_lj.hip``:

* ``nbr_list`` is a ``NeighborListNsqrd`` / ``NeighborListCell``: the list kernel
  over the force list derived from the reference half list (pairs with
  d_ref < cutoff+skin and d_now < cutoff, reference neighbors.py:782).
* ``nbr_list`` is a ``PairListNsqrd``: the all-pairs kernel with the list's space.
* ``nbr_list is None``: the all-pairs kernel without PBC (reference :26-63, :235-258).

The force is the closed-form gradient of the same energy (what ``-jax.grad`` of the
reference evaluates to, reference :21-24 and :322-332); no autodiff is involved.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, unit
from ._device import as_tensor
from .topology import Topology


class NeuralNetworkPotential:
    """Base class (chiron/potential.py:7-63)."""

    def __init__(self, model=None, **kwargs):
        if model is not None:
            self.model = model
            self.topology = model.potential.topology
        else:
            self.topology = kwargs.get("topology")

    def compute_energy(self, positions, nbr_list=None):
        raise NotImplementedError

    def compute_force(self, positions, nbr_list=None):
        raise NotImplementedError

    def compute_pairlist(self, positions, cutoff):
        """All i<j with |x_i - x_j| < cutoff, no PBC (chiron/potential.py:26-63).

        Returns (distance[P], displacement_vectors[P,3], pairs[2,P]); the mask is
        produced by the pair-list kernel, the dynamic-size compaction is host indexing.
        """
        positions = as_tensor(positions)
        n = positions.shape[0]
        n_pairs, mask, dist, rij = ops.pl_calculate(positions, ops.make_box(None, False), float(cutoff))
        ii, kk = torch.nonzero(mask, as_tuple=True)
        jj = torch.where(kk < ii, kk, kk + 1)
        return dist[ii, kk], rij[ii, kk], torch.stack((ii, jj), dim=0)


class IdealGasPotential(NeuralNetworkPotential):
    """Energy and force identically zero (chiron/potential.py:66-127)."""

    def __init__(self, topology):
        if not isinstance(topology, (Topology, property)) and topology is not None:
            raise TypeError(
                f"Topology must be a Topology object, a property, or None, got type(topology) = {type(topology)}"
            )
        self.topology = topology

    def compute_energy(self, positions, nbr_list=None, debug_mode=False):
        return 0.0

    def compute_force(self, positions, nbr_list=None):
        return 0.0


class HarmonicOscillatorPotential(NeuralNetworkPotential):
    """U(x) = (k/2) sum (x - x0)^2 + U0 (chiron/potential.py:335-430).

    Not a pair path: kept so that the reference's integrator tests (a Harmonic oscillator
    propagated with LangevinIntegrator, chiron/tests/test_mcmc.py:12-84) run on the BAOAB kernels
    through the integrator's generic-potential branch.  Energy and the analytic force are plain
    device tensor expressions, not hand-written kernels.
    """

    def __init__(
        self,
        topology,
        k: unit.Quantity = 1.0 * unit.kilocalories_per_mole / unit.angstrom**2,
        x0: unit.Quantity = None,
        U0: unit.Quantity = 0.0 * unit.kilocalories_per_mole,
    ):
        if x0 is None:
            x0 = unit.Quantity(np.array([[0.0, 0.0, 0.0]]), unit.angstrom)
        # validation order and messages follow chiron/potential.py:358-389
        if not isinstance(topology, Topology):
            if not isinstance(topology, property):
                if topology is not None:
                    raise TypeError(
                        f"Topology must be a Topology object or None, type(topology) = {type(topology)}"
                    )
        if not isinstance(k, unit.Quantity):
            raise TypeError(f"k must be a unit.Quantity, type(k) = {type(k)}")
        if not isinstance(x0, unit.Quantity):
            raise TypeError(f"positions must be a unit.Quantity, type(positions) = {type(x0)}")
        if not isinstance(U0, unit.Quantity):
            raise TypeError(f"U0 must be a unit.Quantity, type(U0) = {type(U0)}")
        if not k.unit.is_compatible(unit.kilocalories_per_mole / unit.angstrom**2):
            raise ValueError(
                f"k must be a unit.Quantity with units of energy per distance squared, k.unit = {k.unit}"
            )
        if not x0.unit.is_compatible(unit.angstrom):
            raise ValueError(
                f"positions must be a unit.Quantity with units of distance, positions.unit = {x0.unit}"
            )
        assert x0.shape[1] == 3, f"positions must be a NX3 vector, positions.shape = {x0.shape}"
        if not U0.unit.is_compatible(unit.kilocalories_per_mole):
            raise ValueError(f"U0 must be a unit.Quantity with units of energy, U0.unit = {U0.unit}")
        self.k = float(k.value_in_unit_system(unit.md_unit_system))
        self.x0 = as_tensor(x0.value_in_unit_system(unit.md_unit_system))
        self.U0 = float(U0.value_in_unit_system(unit.md_unit_system))
        self.topology = topology

    def compute_energy(self, positions, nbr_list=None, debug_mode=False):
        d = as_tensor(positions) - self.x0.to(as_tensor(positions).device)
        return 0.5 * self.k * (d * d).sum() + self.U0  # :413-418

    def compute_force(self, positions, nbr_list=None):
        d = as_tensor(positions) - self.x0.to(as_tensor(positions).device)
        return -self.k * d  # -grad U (the reference differentiates :413-418 with jax.grad)


class LJPotential(NeuralNetworkPotential):
    def __init__(
        self,
        topology,
        sigma: unit.Quantity = 3.350 * unit.angstroms,
        epsilon: unit.Quantity = 1.0 * unit.kilocalories_per_mole,
        cutoff: unit.Quantity = unit.Quantity(1.0, unit.nanometer),
    ):
        # validation order and messages follow chiron/potential.py:154-178
        if not isinstance(topology, Topology):
            if not isinstance(topology, property):
                if topology is not None:
                    raise TypeError(
                        f"Topology must be a Topology object or None, type(topology) = {type(topology)}"
                    )
        if not isinstance(sigma, unit.Quantity):
            raise TypeError(f"sigma must be a unit.Quantity, type(sigma) = {type(sigma)}")
        if not isinstance(epsilon, unit.Quantity):
            raise TypeError(f"epsilon must be a unit.Quantity, type(epsilon) = {type(epsilon)}")
        if not isinstance(cutoff, unit.Quantity):
            raise TypeError(f"cutoff must be a unit.Quantity, type(cutoff) = {type(cutoff)}")
        if not sigma.unit.is_compatible(unit.angstrom):
            raise ValueError(f"sigma must have units of distance, got {sigma.unit}")
        if not epsilon.unit.is_compatible(unit.kilocalories_per_mole):
            raise ValueError(f"epsilon must have units of energy, got  {epsilon.unit}")
        if not cutoff.unit.is_compatible(unit.nanometer):
            raise ValueError(f"cutoff must have units of distance, got {cutoff.unit}")

        self.sigma = float(sigma.value_in_unit_system(unit.md_unit_system))
        self.epsilon = float(epsilon.value_in_unit_system(unit.md_unit_system))
        self.cutoff = float(cutoff.value_in_unit_system(unit.md_unit_system))
        self.topology = topology

    # -- dispatch ----------------------------------------------------------
    def _evaluate(self, positions, nbr_list, want_force: bool, want_energy: bool):
        from .neighbors import NeighborListNsqrd, PairListNsqrd

        positions = as_tensor(positions)
        if nbr_list is None:
            # N^2 pairs, no PBC (chiron/potential.py:235-258)
            return ops.lj_allpairs(positions, ops.make_box(None, False), self.sigma, self.epsilon,
                                   self.cutoff, want_force, want_energy)
        if not nbr_list.is_built:
            raise ValueError("Neighborlist must be built before use")  # :263-264
        if isinstance(nbr_list, PairListNsqrd):
            list_cutoff = nbr_list._cutoff_md()
            if list_cutoff is not None and list_cutoff != self.cutoff:
                raise ValueError(
                    f"Neighborlist cutoff ({nbr_list.cutoff}) must be the same as the potential cutoff ({self.cutoff})"
                )
            if positions.shape[0] != nbr_list.n_particles:
                raise ValueError(
                    f"Number of particles cannot changes without rebuilding. "
                    f"Positions must have shape ({nbr_list.n_particles}, 3), found {tuple(positions.shape)}"
                )
            return ops.lj_allpairs(positions, nbr_list._box_h, self.sigma, self.epsilon, list_cutoff,
                                   want_force, want_energy)
        if nbr_list.cutoff.value_in_unit_system(unit.md_unit_system) != self.cutoff:  # :267-270
            raise ValueError(
                f"Neighborlist cutoff ({nbr_list.cutoff}) must be the same as the potential cutoff ({self.cutoff})"
            )
        if not isinstance(nbr_list, NeighborListNsqrd):
            raise TypeError(f"unsupported pair structure {type(nbr_list)}")
        fl = nbr_list.force_list
        x4 = fl.pack(positions, 4)
        f_int, e = fl.energy_force(x4, nbr_list._box_h, self.sigma, self.epsilon, self.cutoff,
                                   want_force, want_energy)
        f = fl.unpack(f_int) if want_force else None
        return f, e

    def compute_energy(self, positions, nbr_list=None, debug_mode=False):
        """Total LJ energy in kJ/mol as a 0-d device tensor (chiron/potential.py:215-279)."""
        _, e = self._evaluate(positions, nbr_list, False, True)
        return e[0]

    def compute_force(self, positions, nbr_list=None) -> torch.Tensor:
        """Forces (N,3) in kJ/mol/nm (chiron/potential.py:281-300)."""
        f, _ = self._evaluate(positions, nbr_list, True, False)
        return f

    def compute_energy_and_force(self, positions, nbr_list=None):
        f, e = self._evaluate(positions, nbr_list, True, True)
        return e[0], f

    def compute_force_analytical(self, positions) -> torch.Tensor:
        """Test helper of the reference (chiron/potential.py:302-332): same kernel, no list."""
        return self.compute_force(positions, None)
