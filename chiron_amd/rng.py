"""Counter-based PRNG keys (Philox4x32-10), host side.

The reference threads a JAX threefry key through ``SamplerState`` and splits it
on every draw (reference: chiron/states.py:150-154, chiron/utils.py:29-38,
chiron/integrators.py:179,185).  Here a key is an opaque pair of 32-bit words;
the device kernels draw from Philox4x32-10 keyed by that pair with the counter
``(particle_index, step, 0, stream_tag)``, so a trajectory never depends on how
steps are batched into launches or on the internal particle ordering.  The JAX
bit stream itself is not reproduced (parity unpinned).
"""
from __future__ import annotations

from typing import Tuple

_M0 = 0xD2511F53
_M1 = 0xCD9E8D57
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = 0xFFFFFFFF

# stream tags (counter word 3); must match csrc/chx_rng.h
STREAM_VELOCITY = 0
STREAM_LANGEVIN = 1
STREAM_MC_DISPLACE = 2
STREAM_UNIFORM = 3
STREAM_SPLIT = 0x5EED


def philox4x32(counter: Tuple[int, int, int, int], key: Tuple[int, int]) -> Tuple[int, int, int, int]:
    c0, c1, c2, c3 = (int(c) & _MASK for c in counter)
    k0, k1 = (int(k) & _MASK for k in key)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> 32, p0 & _MASK
        hi1, lo1 = p1 >> 32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def u32_to_uniform(x: int) -> float:
    """(0,1) uniform with 23 random bits: (x>>9 + 1/2) 2^-23 is exact in fp32, so the
    host, the oracle and the device produce the very same number."""
    return ((int(x) >> 9) + 0.5) * (1.0 / 8388608.0)


class PRNGKey(tuple):
    """Immutable 2x32-bit key; behaves like the opaque key the reference passes around."""

    def __new__(cls, k0: int, k1: int = 0):
        return super().__new__(cls, (int(k0) & _MASK, int(k1) & _MASK))

    def __getnewargs__(self):
        return (self[0], self[1])

    def __copy__(self):
        return self

    def __deepcopy__(self, memo):
        return self  # immutable

    @classmethod
    def from_seed(cls, seed: int) -> "PRNGKey":
        seed = int(seed)
        return cls(seed & _MASK, (seed >> 32) & _MASK)

    @property
    def k0(self) -> int:
        return self[0]

    @property
    def k1(self) -> int:
        return self[1]

    def __repr__(self):
        return f"PRNGKey(0x{self[0]:08x}, 0x{self[1]:08x})"


def split(key: PRNGKey) -> Tuple[PRNGKey, PRNGKey]:
    """Return (new_key, subkey), the analogue of ``jax.random.split(key)``."""
    o = philox4x32((0, 0, 0, STREAM_SPLIT), key)
    return PRNGKey(o[0], o[1]), PRNGKey(o[2], o[3])


def fold_in(key: PRNGKey, data: int) -> PRNGKey:
    o = philox4x32((int(data) & _MASK, (int(data) >> 32) & _MASK, 1, STREAM_SPLIT), key)
    return PRNGKey(o[0], o[1])


def uniform(key: PRNGKey, counter: int = 0) -> float:
    """One (0,1) uniform from ``key`` -- the host twin of the device STREAM_UNIFORM draw."""
    o = philox4x32((0, int(counter) & _MASK, 0, STREAM_UNIFORM), key)
    return u32_to_uniform(o[0])
