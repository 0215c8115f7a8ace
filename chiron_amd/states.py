"""State containers: ``SamplerState`` and ``ThermodynamicState``.

Same names, constructor arguments, properties and error behaviour as the
reference (chiron/states.py:8-174 and :177-329); arrays are ``torch`` tensors
(fp32, on the ROCm device when one is visible) instead of ``jax`` arrays, and
the PRNG key is a Philox key pair (see ``chiron_amd.rng``).
"""
from __future__ import annotations

import threading

from typing import List, Optional, Union

import numpy as np
import torch

from . import unit
from ._device import as_tensor
from .rng import PRNGKey, split


def _shape_of(q) -> tuple:
    v = q._value if isinstance(q, unit.Quantity) else q
    if hasattr(v, "shape"):
        return tuple(v.shape)
    return tuple(np.asarray(v).shape)


class SamplerState:
    """Positions / velocities / box vectors / PRNG key of one replica.

    Reference: chiron/states.py:8-174.
    """

    def __init__(
        self,
        positions: unit.Quantity,
        current_PRNG_key: PRNGKey = None,
        velocities: Optional[unit.Quantity] = None,
        box_vectors: Optional[unit.Quantity] = None,
    ) -> None:
        # validation order follows chiron/states.py:44-87
        if not isinstance(positions, unit.Quantity):
            raise TypeError(f"positions must be a unit.Quantity, got {type(positions)} instead.")
        if velocities is not None and not isinstance(velocities, unit.Quantity):
            raise TypeError(f"velocities must be a unit.Quantity, got {type(velocities)} instead.")
        if box_vectors is not None and not isinstance(box_vectors, unit.Quantity):
            if isinstance(box_vectors, list):
                try:
                    box_vectors = self._convert_from_openmm_box(box_vectors)
                except Exception:
                    raise TypeError(f"Unable to parse box_vectors {box_vectors}.")
            else:
                raise TypeError(
                    f"box_vectors must be a unit.Quantity or openMM box, got {type(box_vectors)} instead."
                )
        if not positions.unit.is_compatible(unit.nanometer):
            raise ValueError(f"positions must have units of distance, got {positions.unit} instead.")
        if velocities is not None and not velocities.unit.is_compatible(unit.nanometer / unit.picosecond):
            raise ValueError(
                f"velocities must have units of distance/time, got {velocities.unit} instead."
            )
        if box_vectors is not None and not box_vectors.unit.is_compatible(unit.nanometer):
            raise ValueError(f"box_vectors must have units of distance, got {box_vectors.unit} instead.")
        if box_vectors is not None and _shape_of(box_vectors) != (3, 3):
            raise ValueError(f"box_vectors must be a 3x3 array, got {_shape_of(box_vectors)} instead.")
        if velocities is not None and _shape_of(positions) != _shape_of(velocities):
            raise ValueError(
                f"positions and velocities must have the same shape, got {_shape_of(positions)} and {_shape_of(velocities)} instead."
            )
        if current_PRNG_key is None:
            raise ValueError("random_seed must be set.")

        self._distance_unit = unit.nanometer
        self._time_unit = unit.picosecond
        # stored canonically as fp32 device tensors in nm / nm ps^-1, so the
        # properties below are O(1) instead of a unit conversion per access
        self._positions = self._to_md(positions)
        self._velocities = None if velocities is None else self._to_md(velocities)
        self._box_vectors = None if box_vectors is None else self._to_md(box_vectors)
        self._box_token = object()  # identity of the box VALUES: copies share it, assigning a box renews it
        self._current_PRNG_key = current_PRNG_key

    def __deepcopy__(self, memo):
        """Same result as the generic deep copy (every caller of the reference copies states
        freely, chiron/integrators.py:210, chiron/mcmc.py:726), without walking the object graph:
        three tensor clones."""
        new = SamplerState.__new__(SamplerState)
        memo[id(self)] = new
        for name, value in self.__dict__.items():
            if name == "_box_pool":
                continue  # (spare box copies: the deep copy starts its own pool when it needs one)
            if isinstance(value, unit.Quantity) and isinstance(value._value, torch.Tensor):
                value = unit.Quantity(value._value.clone(), value.unit)
            elif name not in ("_distance_unit", "_time_unit", "_current_PRNG_key", "_box_token", "_v_unit"):
                import copy

                value = copy.deepcopy(value, memo)
            setattr(new, name, value)
        return new

    def _copy_with(self, positions: torch.Tensor, velocities: torch.Tensor, box_copy=None) -> "SamplerState":
        """``deepcopy(self)`` followed by assigning fresh md-unit fp32 device arrays (what
        chiron/integrators.py:210-214 does), without cloning the two arrays that are replaced.
        ``box_copy``: a copy of the box values the caller has made already (``_box_copy``: the integrator takes it
        while its steps run instead of behind them)."""
        if self._box_vectors is not None and box_copy is None:
            box_copy = self._box_copy()  # (before the attributes are handed on: the copy inherits the pool of spare boxes)
        new = SamplerState.__new__(SamplerState)
        new.__dict__.update(self.__dict__)
        if self._box_vectors is not None:
            new._box_vectors = box_copy
        new._positions = unit.Quantity(positions, self._distance_unit)
        new._velocities = unit.Quantity(velocities, self._velocity_unit())
        return new

    _BOX_POOL = 32
    _box_pool_lock = threading.Lock()

    def _box_copy(self):
        """A private copy of the box values for a state that is about to be returned (chiron/integrators.py:210: the
        reference deep-copies the state).  Copies are made 32 at a time -- one small launch fills a pool of independent
        3x3 blocks, and the states of one trajectory (they share the pool through ``_copy_with``) take one each: a clone
        per ``run()`` put a copy kernel into every call's dependent launch chain (2-5 us of a 230 us run(20), measured
        in round 4).  A new box (the setter renews ``_box_token``) starts a new pool, and so does a box that was edited
        IN PLACE since the pool was filled (``state.box_vectors.mul_(s)``: the property hands out the tensor itself):
        the pool remembers the version counter of the tensor it was filled from and its own -- a state's box is either
        that tensor or a block of the pool, and an in-place write to either moves the counter -- so the copy always
        carries the values the state holds NOW, as ``clone()`` did (round-4 review).  The blocks are dealt under a lock
        (concurrent replicas derive their states from one pool)."""
        if self._box_vectors is None:
            return None
        box = self._box_vectors._value
        with SamplerState._box_pool_lock:
            pool = self.__dict__.get("_box_pool")
            fresh = pool is not None and pool[0] is self._box_token and pool[2] < self._BOX_POOL
            if fresh:
                if box is pool[3]:
                    fresh = box._version == pool[4]
                else:  # a block of the pool (a view of it shares its version counter), or a tensor from elsewhere
                    fresh = box._base is pool[1] and pool[1]._version == pool[5]
            if not fresh:
                filled = box.unsqueeze(0).repeat(self._BOX_POOL, 1, 1)
                pool = self.__dict__["_box_pool"] = [self._box_token, filled, 0, box, box._version, filled._version]
            out = pool[1][pool[2]]
            pool[2] += 1
        return unit.Quantity(out, self._box_vectors.unit)

    def _velocity_unit(self):
        """distance / time (unit arithmetic is exact-rational Python: once per pair of units, not per run() call)"""
        cached = self.__dict__.get("_v_unit")
        if cached is None or cached[0] is not self._distance_unit or cached[1] is not self._time_unit:
            cached = self.__dict__["_v_unit"] = (self._distance_unit, self._time_unit, self._distance_unit / self._time_unit)
        return cached[2]

    def _to_md(self, q: unit.Quantity) -> unit.Quantity:
        value = q.value_in_unit_system(unit.md_unit_system)
        target = self._distance_unit
        if q.unit.is_compatible(unit.nanometer / unit.picosecond):
            target = self._velocity_unit()
        return unit.Quantity(as_tensor(value), target)

    # -- read-only views ---------------------------------------------------
    @property
    def number_of_particles(self) -> int:
        return self._positions.shape[0]

    @property
    def positions(self) -> torch.Tensor:
        return self._positions._value

    @property
    def velocities(self) -> Optional[torch.Tensor]:
        if self._velocities is None:
            return None
        return self._velocities._value

    @property
    def box_vectors(self) -> Optional[torch.Tensor]:
        if self._box_vectors is None:
            return None
        return self._box_vectors._value

    # -- setters (accept raw arrays in md units or Quantities) -------------
    @positions.setter
    def positions(self, x0: Union[torch.Tensor, np.ndarray, unit.Quantity]) -> None:
        if isinstance(x0, unit.Quantity):
            self._positions = self._to_md(x0)
        else:
            self._positions = unit.Quantity(as_tensor(x0), self._distance_unit)

    @box_vectors.setter
    def box_vectors(self, box_vectors) -> None:
        self._box_token = object()
        if isinstance(box_vectors, unit.Quantity):
            self._box_vectors = self._to_md(box_vectors)
        else:
            self._box_vectors = unit.Quantity(as_tensor(box_vectors), self._distance_unit)

    @velocities.setter
    def velocities(self, velocities) -> None:
        if _shape_of(velocities) != _shape_of(self._positions):
            raise ValueError(
                f"velocities must have the same shape as positions, got {_shape_of(velocities)} and {_shape_of(self._positions)} instead."
            )
        if isinstance(velocities, unit.Quantity):
            self._velocities = self._to_md(velocities)
        else:
            self._velocities = unit.Quantity(as_tensor(velocities), self._velocity_unit())

    @property
    def distance_unit(self) -> unit.Unit:
        return self._distance_unit

    def velocity_unit(self) -> unit.Unit:
        return self._velocity_unit()

    @property
    def current_PRNG_key(self) -> PRNGKey:
        return self._current_PRNG_key

    @current_PRNG_key.setter
    def current_PRNG_key(self, key: PRNGKey) -> None:
        self._current_PRNG_key = key

    @property
    def new_PRNG_key(self) -> PRNGKey:
        """Split the stored key, keep one half, return the other (chiron/states.py:150-154)."""
        key, subkey = split(self._current_PRNG_key)
        self._current_PRNG_key = key
        return subkey

    def _convert_from_openmm_box(self, openmm_box_vectors: List) -> unit.Quantity:
        # chiron/states.py:165-174
        box_vec = []
        u0 = openmm_box_vectors[0].unit
        for i in range(3):
            layer = []
            for j in range(3):
                layer.append(float(openmm_box_vectors[i][j].value_in_unit(u0)))
            box_vec.append(layer)
        return unit.Quantity(np.asarray(box_vec), u0)


class ThermodynamicState:
    """Potential + (T, V, p); provides the reduced potential (chiron/states.py:177-329)."""

    def __init__(
        self,
        potential,
        temperature: Optional[unit.Quantity] = None,
        volume: Optional[unit.Quantity] = None,
        pressure: Optional[unit.Quantity] = None,
    ):
        self.potential = potential

        if temperature is not None and not isinstance(temperature, unit.Quantity):
            raise TypeError(f"temperature must be a unit.Quantity, got {type(temperature)} instead.")
        elif temperature is not None:
            if not temperature.unit.is_compatible(unit.kelvin):
                raise ValueError(
                    f"temperature must have units of temperature, got {temperature.unit} instead."
                )
        if volume is not None and not isinstance(volume, unit.Quantity):
            raise TypeError(f"volume must be a unit.Quantity, got {type(volume)} instead.")
        elif volume is not None:
            if not volume.unit.is_compatible(unit.nanometer**3):
                raise ValueError(f"volume must have units of distance**3, got {volume.unit} instead.")
        if pressure is not None and not isinstance(pressure, unit.Quantity):
            raise TypeError(f"pressure must be a unit.Quantity, got {type(pressure)} instead.")
        elif pressure is not None:
            if not pressure.unit.is_compatible(unit.atmosphere):
                raise ValueError(f"pressure must have units of pressure, got {pressure.unit} instead.")

        self.temperature = temperature
        if temperature is not None:
            self.beta = 1.0 / (unit.BOLTZMANN_CONSTANT_kB * self.temperature)
        else:
            self.beta = None
        self.volume = volume
        self.pressure = pressure

        from .utils import get_nr_of_particles

        self.nr_of_particles = get_nr_of_particles(self.potential.topology)

    def check_variables(self) -> List[str]:
        variables = ["temperature", "volume", "pressure"]
        return [var for var in variables if getattr(self, var) is not None]

    def get_reduced_potential(self, sampler_state: SamplerState, nbr_list=None):
        """u = beta (U/N_A + p V)  (chiron/states.py:275-325).

        The potential energy comes back from the HIP kernel as a device scalar
        and is read to the host here (the reference also synchronises at this
        point: every caller branches on the value).
        """
        if self.beta is None:
            self.beta = 1.0 / (unit.BOLTZMANN_CONSTANT_kB * (self.temperature * unit.kelvin))
        energy = self.potential.compute_energy(sampler_state.positions, nbr_list)
        return self.reduced_potential_from_energy(float(energy), sampler_state)

    def reduced_potential_from_energy(self, energy: float, sampler_state: SamplerState):
        """The arithmetic of ``get_reduced_potential`` for an already known potential energy (kJ/mol)."""
        if self.beta is None:
            self.beta = 1.0 / (unit.BOLTZMANN_CONSTANT_kB * (self.temperature * unit.kelvin))
        reduced_potential = unit.Quantity(float(energy), unit.kilojoule_per_mole) / unit.AVOGADRO_CONSTANT_NA
        if self.pressure is not None:
            box = sampler_state.box_vectors
            if isinstance(box, torch.Tensor):
                box = box.detach().cpu().numpy()
            self.volume = (float(box[0][0]) * float(box[1][1]) * float(box[2][2])) * unit.nanometer**3
            reduced_potential = reduced_potential + self.pressure * self.volume
        return self.beta * reduced_potential

    def kT_to_kJ_per_mol(self, energy):
        energy = energy * unit.AVOGADRO_CONSTANT_NA
        return energy / self.beta


def calculate_reduced_potential_at_states(
    sampler_state: SamplerState,
    thermodynamic_states: List[ThermodynamicState],
    nbr_list=None,
) -> np.ndarray:
    """Reduced potential of one configuration at every state (chiron/states.py:335-366)."""
    reduced_potentials = np.zeros(len(thermodynamic_states))
    for state_idx, state in enumerate(thermodynamic_states):
        reduced_potentials[state_idx] = state.get_reduced_potential(sampler_state, nbr_list)
    return reduced_potentials
