"""Minimal ``openmm.unit``-compatible unit algebra for the chiron hot path.

The reference passes every physical parameter as an ``openmm.unit.Quantity``
(reference: chiron/potential.py:154-188, chiron/neighbors.py:227-232,
chiron/states.py:44-87,156-163).  ``openmm`` is not installed in this image nor
on the GPU box, so this module provides exactly the subset of the API the hot
path touches: ``Quantity``, ``Unit`` arithmetic, ``is_compatible``,
``value_in_unit``, ``value_in_unit_system(md_unit_system)`` and the named units
and constants used by the reference sources/tests/examples.

Conventions follow OpenMM: ``dalton``/``amu`` is *gram/mole* (mass per amount),
so that ``md_unit_system`` energies come out in kJ/mol, and a product of
quantities whose units cancel collapses to a plain number.
"""
from __future__ import annotations

import math
from fractions import Fraction
from typing import Any, Tuple

import numpy as np

# base dimensions: length, mass, time, temperature, amount, angle, charge
_NDIM = 7
_L, _M, _T, _TH, _N, _A, _Q = range(_NDIM)
_DIM_NAMES = ("length", "mass", "time", "temperature", "amount", "angle", "charge")


def _dims(**kw) -> Tuple[Fraction, ...]:
    d = [Fraction(0)] * _NDIM
    for k, v in kw.items():
        d[_DIM_NAMES.index(k)] = Fraction(v)
    return tuple(d)


def _is_number(x: Any) -> bool:
    return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, bool)


class Unit:
    """A physical unit: SI scale factor plus exponents over the base dimensions."""

    __array_ufunc__ = None  # make numpy defer to __rmul__/__rtruediv__
    __array_priority__ = 1000

    __slots__ = ("dims", "scale", "name", "_sys_factor")

    def __init__(self, dims, scale, name: str):
        self.dims = tuple(Fraction(d) for d in dims)
        # exact rational scale whenever possible so that e.g. 10 angstrom == 1 nm exactly
        self.scale = _exact(scale)
        self.name = name
        self._sys_factor = None  # (unit system id, float factor): exact-rational arithmetic is slow

    # -- algebra -----------------------------------------------------------
    def is_compatible(self, other: "Unit") -> bool:
        return isinstance(other, Unit) and self.dims == other.dims

    def is_dimensionless(self) -> bool:
        return all(d == 0 for d in self.dims)

    def conversion_factor_to(self, other: "Unit") -> float:
        if not self.is_compatible(other):
            raise TypeError(f'Unit "{self.name}" is not compatible with Unit "{other.name}".')
        return float(self.scale / other.scale)

    def __mul__(self, other):
        if isinstance(other, Unit):
            return Unit(
                tuple(a + b for a, b in zip(self.dims, other.dims)),
                self.scale * other.scale,
                f"{self.name}*{other.name}",
            )
        if isinstance(other, Quantity):
            return Quantity(other._value, self * other.unit)._collapse()
        return Quantity(other, self)

    def __rmul__(self, other):
        # value * unit -> Quantity
        return Quantity(other, self)

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return Unit(
                tuple(a - b for a, b in zip(self.dims, other.dims)),
                self.scale / other.scale,
                f"{self.name}/({other.name})",
            )
        if isinstance(other, Quantity):
            return Quantity(1.0 / other._value, self / other.unit)._collapse()
        return Quantity(1.0 / other, self)

    def __rtruediv__(self, other):
        # value / unit -> Quantity in inverse unit
        return Quantity(other, self ** -1)

    def __pow__(self, p):
        p = Fraction(p).limit_denominator(1000)
        if p.denominator == 1 and isinstance(self.scale, Fraction):
            scale = self.scale ** int(p)
        else:
            scale = float(self.scale) ** float(p)
        return Unit(tuple(d * p for d in self.dims), scale, f"({self.name})**{p}")

    def __eq__(self, other):
        return isinstance(other, Unit) and self.dims == other.dims and math.isclose(
            float(self.scale), float(other.scale), rel_tol=1e-12
        )

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self.dims, round(math.log(float(self.scale)) if self.scale > 0 else 0.0, 9)))

    def __repr__(self):
        return f"Unit({self.name})"

    def __str__(self):
        return self.name


def _exact(scale):
    if isinstance(scale, Fraction):
        return scale
    if isinstance(scale, int):
        return Fraction(scale)
    if isinstance(scale, str):
        return Fraction(scale)
    f = float(scale)
    # decimal literals such as 1e-9 or 4184.0 are meant exactly
    fr = Fraction(repr(f))
    return fr if float(fr) == f and fr.denominator <= 10**30 else f


def _scale_value(value, factor: float):
    """Multiply ``value`` (number / list / ndarray / torch tensor) by ``factor``."""
    if factor == 1.0:
        if isinstance(value, (list, tuple)):
            return np.asarray(value)
        return value
    if isinstance(value, (list, tuple)):
        return np.asarray(value) * factor
    return value * factor


class UnitSystem:
    def __init__(self, name: str, base_scales):
        self.name = name
        # SI scale of the system's base unit for each dimension, where the mass
        # slot holds the scale of dalton (= gram/mole, mass*amount^-1).
        self.base_scales = base_scales
        self._scale_cache = {}

    def scale_for(self, dims):
        """SI scale of this system's unit for ``dims`` (memoised: the hot callers convert the same
        handful of dimensions -- length, velocity, energy -- thousands of times)."""
        hit = self._scale_cache.get(dims)
        if hit is None:
            hit = self._scale_cache[dims] = self._scale_for(dims)
        return hit

    def _scale_for(self, dims):
        l, m, t, th, n, a, q = dims
        s = self.base_scales
        # mass^m is expressed as dalton^m * mole^m (dalton = mass/amount)
        out = Fraction(1)
        for base, p in (("length", l), ("dalton", m), ("amount", n + m), ("time", t),
                        ("temperature", th), ("angle", a), ("charge", q)):
            p = Fraction(p)
            if p == 0:
                continue
            b = _exact(s[base])
            if p.denominator == 1 and isinstance(b, Fraction) and not isinstance(out, float):
                out = out * b ** int(p)
            else:
                out = float(out) * float(b) ** float(p)
        return out


class Quantity:
    """A value (number, list, numpy array or torch tensor) tagged with a Unit."""

    __array_ufunc__ = None
    __array_priority__ = 1000

    __slots__ = ("_value", "unit")

    def __init__(self, value=None, unit: Unit = None):
        if isinstance(value, Quantity):
            if unit is None:
                unit = value.unit
                value = value._value
            else:
                value = value.value_in_unit(unit)
        if unit is None:
            unit = dimensionless
        if isinstance(value, (list, tuple)):
            # a list of Quantities (e.g. an OpenMM-style box) -> plain array
            if len(value) > 0 and isinstance(value[0], Quantity):
                u0 = value[0].unit
                value = np.asarray([v.value_in_unit(u0) for v in value])
                unit = u0 * unit if not unit.is_dimensionless() else u0
            else:
                value = np.asarray(value)
        self._value = value
        self.unit = unit

    # -- conversions -------------------------------------------------------
    def value_in_unit(self, unit: Unit):
        return _scale_value(self._value, self.unit.conversion_factor_to(unit))

    def value_in_unit_system(self, system: UnitSystem):
        u = self.unit
        cached = u._sys_factor
        if cached is not None and cached[0] == id(system):
            return _scale_value(self._value, cached[1])
        factor = float(u.scale / system.scale_for(u.dims))
        u._sys_factor = (id(system), factor)
        return _scale_value(self._value, factor)

    def in_units_of(self, unit: Unit) -> "Quantity":
        return Quantity(self.value_in_unit(unit), unit)

    def _collapse(self):
        if self.unit.is_dimensionless():
            return _scale_value(self._value, float(self.unit.scale))
        return self

    # -- container behaviour ----------------------------------------------
    @property
    def shape(self):
        return self._value.shape

    def __len__(self):
        return len(self._value)

    def __getitem__(self, key):
        return Quantity(self._value[key], self.unit)

    def __iter__(self):
        for v in self._value:
            yield Quantity(v, self.unit)

    # -- arithmetic --------------------------------------------------------
    def __add__(self, other):
        if not isinstance(other, Quantity):
            raise TypeError(f"Cannot add Quantity and {type(other)}")
        return Quantity(self._value + other.value_in_unit(self.unit), self.unit)

    def __sub__(self, other):
        if not isinstance(other, Quantity):
            raise TypeError(f"Cannot subtract {type(other)} from Quantity")
        return Quantity(self._value - other.value_in_unit(self.unit), self.unit)

    def __neg__(self):
        return Quantity(-self._value, self.unit)

    def __mul__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value * other._value, self.unit * other.unit)._collapse()
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit * other)._collapse()
        return Quantity(self._value * other, self.unit)

    def __rmul__(self, other):
        return Quantity(other * self._value, self.unit)

    def __truediv__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value / other._value, self.unit / other.unit)._collapse()
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit / other)._collapse()
        return Quantity(self._value / other, self.unit)

    def __rtruediv__(self, other):
        return Quantity(other / self._value, self.unit ** -1)

    def __pow__(self, p):
        return Quantity(self._value ** p, self.unit ** p)

    def __abs__(self):
        return Quantity(abs(self._value), self.unit)

    # -- comparisons -------------------------------------------------------
    def _cmp_value(self, other):
        if not isinstance(other, Quantity):
            raise TypeError(f"Cannot compare Quantity with {type(other)}")
        return other.value_in_unit(self.unit)

    def __eq__(self, other):
        if not isinstance(other, Quantity) or not self.unit.is_compatible(other.unit):
            return False
        return self._value == other.value_in_unit(self.unit)

    def __ne__(self, other):
        r = self.__eq__(other)
        return ~r if hasattr(r, "shape") and getattr(r, "shape", ()) != () else not r

    def __lt__(self, other):
        return self._value < self._cmp_value(other)

    def __le__(self, other):
        return self._value <= self._cmp_value(other)

    def __gt__(self, other):
        return self._value > self._cmp_value(other)

    def __ge__(self, other):
        return self._value >= self._cmp_value(other)

    __hash__ = None

    def __bool__(self):
        return bool(self._value)

    def __float__(self):
        return float(self._value)

    def __repr__(self):
        return f"Quantity(value={self._value!r}, unit={self.unit.name})"

    def __str__(self):
        return f"{self._value} {self.unit.name}"

    def __deepcopy__(self, memo):
        import copy

        return Quantity(copy.deepcopy(self._value, memo), self.unit)


def is_quantity(x) -> bool:
    return isinstance(x, Quantity)


# ---------------------------------------------------------------------------
# named units (SI scale)
# ---------------------------------------------------------------------------
dimensionless = Unit(_dims(), 1.0, "dimensionless")

meter = meters = Unit(_dims(length=1), 1.0, "meter")
nanometer = nanometers = Unit(_dims(length=1), 1e-9, "nanometer")
angstrom = angstroms = Unit(_dims(length=1), 1e-10, "angstrom")
centimeter = centimeters = Unit(_dims(length=1), 1e-2, "centimeter")
liter = liters = Unit(_dims(length=3), 1e-3, "liter")

second = seconds = Unit(_dims(time=1), 1.0, "second")
picosecond = picoseconds = Unit(_dims(time=1), 1e-12, "picosecond")
femtosecond = femtoseconds = Unit(_dims(time=1), 1e-15, "femtosecond")
nanosecond = nanoseconds = Unit(_dims(time=1), 1e-9, "nanosecond")

kilogram = kilograms = Unit(_dims(mass=1), 1.0, "kilogram")
gram = grams = Unit(_dims(mass=1), 1e-3, "gram")

kelvin = kelvins = Unit(_dims(temperature=1), 1.0, "kelvin")
mole = moles = Unit(_dims(amount=1), 1.0, "mole")
radian = radians = Unit(_dims(angle=1), 1.0, "radian")
degree = degrees = Unit(_dims(angle=1), math.pi / 180.0, "degree")
elementary_charge = elementary_charges = Unit(_dims(charge=1), 1.0, "elementary charge")

# OpenMM convention: dalton == amu == gram/mole
dalton = daltons = amu = amus = Unit(_dims(mass=1, amount=-1), 1e-3, "dalton")

joule = joules = Unit(_dims(mass=1, length=2, time=-2), 1.0, "joule")
kilojoule = kilojoules = Unit(_dims(mass=1, length=2, time=-2), 1e3, "kilojoule")
kilocalorie = kilocalories = Unit(_dims(mass=1, length=2, time=-2), 4184.0, "kilocalorie")
kilojoule_per_mole = kilojoules_per_mole = Unit(
    _dims(mass=1, length=2, time=-2, amount=-1), 1e3, "kilojoule/mole"
)
kilocalorie_per_mole = kilocalories_per_mole = Unit(
    _dims(mass=1, length=2, time=-2, amount=-1), 4184.0, "kilocalorie/mole"
)
joule_per_mole = joules_per_mole = Unit(
    _dims(mass=1, length=2, time=-2, amount=-1), 1.0, "joule/mole"
)

pascal = pascals = Unit(_dims(mass=1, length=-1, time=-2), 1.0, "pascal")
bar = bars = Unit(_dims(mass=1, length=-1, time=-2), 1e5, "bar")
atmosphere = atmospheres = Unit(_dims(mass=1, length=-1, time=-2), 101325.0, "atmosphere")

md_unit_system = UnitSystem(
    "md_unit_system",
    dict(
        length=1e-9,  # nanometer
        dalton=1e-3,  # gram/mole
        amount=1.0,  # mole
        time=1e-12,  # picosecond
        temperature=1.0,  # kelvin
        angle=1.0,  # radian
        charge=1.0,  # elementary charge
    ),
)

# constants (CODATA 2018, the values OpenMM ships)
BOLTZMANN_CONSTANT_kB = Quantity(1.380649e-23, joule / kelvin)
AVOGADRO_CONSTANT_NA = Quantity(6.02214076e23, mole ** -1)
MOLAR_GAS_CONSTANT_R = BOLTZMANN_CONSTANT_kB * AVOGADRO_CONSTANT_NA
