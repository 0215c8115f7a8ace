"""Minimal ``openmm.app.Topology`` / ``Element`` stand-in.

The hot path only reads the particle count and per-atom element masses from a
topology (reference: chiron/utils.py:101-113), and the tests build one with
``Topology().addChain()/addResidue()/addAtom()`` (reference:
chiron/tests/test_mcmc.py:382-389, Examples/LJ_MCMC.py:15-21).
"""
from __future__ import annotations

from typing import Iterator, List, Optional

from . import unit

# standard atomic weights (amu) of the elements the reference systems use
_MASSES = {
    "H": (1, "hydrogen", 1.007947),
    "He": (2, "helium", 4.003),
    "C": (6, "carbon", 12.01078),
    "N": (7, "nitrogen", 14.00672),
    "O": (8, "oxygen", 15.99943),
    "F": (9, "fluorine", 18.99840325),
    "Ne": (10, "neon", 20.17976),
    "Na": (11, "sodium", 22.989769282),
    "P": (15, "phosphorus", 30.9737622),
    "S": (16, "sulfur", 32.0655),
    "Cl": (17, "chlorine", 35.4532),
    "Ar": (18, "argon", 39.948),
    "K": (19, "potassium", 39.09831),
    "Kr": (36, "krypton", 83.7982),
    "Xe": (54, "xenon", 131.2936),
}


class Element:
    _by_symbol = {}

    def __init__(self, number: int, name: str, symbol: str, mass):
        self.atomic_number = number
        self.name = name
        self.symbol = symbol
        if not isinstance(mass, unit.Quantity):
            mass = unit.Quantity(mass, unit.dalton)
        self.mass = mass

    @classmethod
    def getBySymbol(cls, symbol: str) -> "Element":
        if symbol not in cls._by_symbol:
            if symbol not in _MASSES:
                raise KeyError(f"Unknown element symbol {symbol}")
            n, name, m = _MASSES[symbol]
            cls._by_symbol[symbol] = Element(n, name, symbol, m)
        return cls._by_symbol[symbol]

    def __repr__(self):
        return f"<Element {self.name}>"


class Atom:
    def __init__(self, name: str, element: Element, index: int, residue):
        self.name = name
        self.element = element
        self.index = index
        self.residue = residue


class Residue:
    def __init__(self, name: str, index: int, chain):
        self.name = name
        self.index = index
        self.chain = chain
        self._atoms: List[Atom] = []


class Chain:
    def __init__(self, index: int, topology):
        self.index = index
        self.topology = topology
        self._residues: List[Residue] = []


class Topology:
    def __init__(self):
        self._chains: List[Chain] = []
        self._segments: list = []  # lists of explicit Atom objects and (n, name, element, residue) runs
        self._n_residues = 0
        self._masses_amu = None

    def addChain(self, id: Optional[str] = None) -> Chain:
        c = Chain(len(self._chains), self)
        self._chains.append(c)
        return c

    def addResidue(self, name: str, chain: Chain) -> Residue:
        r = Residue(name, self._n_residues, chain)
        self._n_residues += 1
        chain._residues.append(r)
        return r

    def addAtom(self, name: str, element: Element, residue: Residue) -> Atom:
        a = Atom(name, element, self.getNumAtoms(), residue)
        if not self._segments or not isinstance(self._segments[-1], list):
            self._segments.append([])
        self._segments[-1].append(a)
        residue._atoms.append(a)
        self._masses_amu = None
        return a

    def addAtoms(self, n: int, name: str, element: Element, residue: Residue) -> None:
        """``n`` identical atoms stored as ONE run instead of n Python objects.

        A 262,144-atom topology made of individual ``Atom`` objects adds ~1e6 tracked objects to
        the interpreter; every full garbage-collection pass then stalls the launch loop for tens of
        milliseconds.  ``atoms()`` still yields (transient) ``Atom`` objects in index order.
        """
        if n > 0:
            self._segments.append((int(n), name, element, residue))
            self._masses_amu = None

    def masses_amu(self):
        """Per-atom masses in amu as a cached float64 array (one unit conversion per element)."""
        import numpy as np

        if self._masses_amu is None:
            per_element = {}

            def mass_of(el):
                m = per_element.get(id(el))
                if m is None:
                    m = per_element[id(el)] = float(el.mass.value_in_unit(unit.amu))
                return m

            parts = []
            for seg in self._segments:
                if isinstance(seg, list):
                    parts.append(np.array([mass_of(a.element) for a in seg], dtype=np.float64))
                else:
                    parts.append(np.full(seg[0], mass_of(seg[2]), dtype=np.float64))
            self._masses_amu = np.concatenate(parts) if parts else np.zeros(0)
        return self._masses_amu

    def atoms(self) -> Iterator[Atom]:
        index = 0
        for seg in self._segments:
            if isinstance(seg, list):
                for a in seg:
                    yield a
                index += len(seg)
            else:
                n, name, element, residue = seg
                for k in range(n):
                    yield Atom(name, element, index + k, residue)
                index += n

    def getNumAtoms(self) -> int:
        return sum(len(seg) if isinstance(seg, list) else seg[0] for seg in self._segments)

    @classmethod
    def homogeneous(cls, n: int, symbol: str = "Ar") -> "Topology":
        """``n`` atoms of one element in one chain/residue (LJ fluid / ideal gas)."""
        top = cls()
        el = Element.getBySymbol(symbol)
        chain = top.addChain()
        res = top.addResidue("system", chain)
        top.addAtoms(n, symbol, el, res)
        return top
