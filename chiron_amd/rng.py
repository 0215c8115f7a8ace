"""Counter-based PRNG keys (Philox4x32-10), host side.

The reference threads a JAX threefry key through ``SamplerState`` and splits it
on every draw (reference: chiron/states.py:150-154, chiron/utils.py:29-38,
chiron/integrators.py:179,185).  Here a key is an opaque pair of 32-bit words;
the device kernels draw from Philox4x32-10 keyed by that pair with the counter
``(particle_index, step, 0, stream_tag)``, so a trajectory never depends on how
steps are batched into launches or on the internal particle ordering.  The JAX
bit stream itself is not what the product draws from by default.

Optional: ``PRNG.set_seed(seed, stream="threefry")`` hands out ``JaxKey``s instead.  ``split`` and
``uniform`` then follow ``jax.random`` (classic threefry key handling, restated below from the published
algorithm), and the integrator / moves draw their normals with ``ops.jax_normal`` (device kernel
``chx_jax_normal``), so a run seeded like the reference consumes the reference's random numbers -- at the price
of materialising the noise of a whole ``run()`` call, i.e. meant for reproducing reference results on small
systems, not for production runs.
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


_fast = None  # (callable, output array type) once libchx HAS BEEN loaded by someone else: chx_philox4x32_h, 1 us instead of 7


def _bind_fast():
    """The library's host Philox (include/chx.h: chx_philox4x32_h) in place of the Python loop below: key handling
    (split / fold_in / uniform) sits on the entry and exit path of every short run() call.  Same bits (tested).

    Key handling is host arithmetic and must stay free of side effects: it never LOADS the library itself (loading
    shares torch's HIP runtime and initialises the device context -- a launcher or a parent process that only hands
    out keys must not have touched the GPU before it starts its ranks; round-4 review).  It binds to a library the
    process has already loaded for its kernels, and until then -- or if that load failed -- every call uses the
    pure-Python rounds, which are the definition anyway; nothing is latched, the next call looks again."""
    global _fast
    import ctypes

    from . import _lib

    lib = _lib._lib
    if lib is None:
        return None
    # (no shared output buffer: ctypes releases the GIL for the call, and MultiStateSampler(concurrent_replicas > 1) splits
    # keys from two threads at once -- a buffer kept here handed one thread the other's words, found in round 4)
    _fast = (lib.chx_philox4x32_h, ctypes.c_uint32 * 4)
    return _fast


def philox4x32(counter: Tuple[int, int, int, int], key: Tuple[int, int]) -> Tuple[int, int, int, int]:
    fast = _fast or _bind_fast()
    if fast:
        fn, out_t = fast
        out = out_t()
        fn(int(counter[0]) & _MASK, int(counter[1]) & _MASK, int(counter[2]) & _MASK, int(counter[3]) & _MASK,
           int(key[0]) & _MASK, int(key[1]) & _MASK, out)
        return out[0], out[1], out[2], out[3]
    return philox4x32_py(counter, key)


def philox4x32_py(counter: Tuple[int, int, int, int], key: Tuple[int, int]) -> Tuple[int, int, int, int]:
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


class JaxKey(PRNGKey):
    """A classic ``jax.random`` threefry key (two uint32 words); selects the JAX-compatible stream."""

    @classmethod
    def from_seed(cls, seed: int) -> "JaxKey":
        seed = int(seed)  # jax.random.PRNGKey(seed) = (seed >> 32, seed & 0xFFFFFFFF)
        return cls((seed >> 32) & _MASK, seed & _MASK)

    def __repr__(self):
        return f"JaxKey(0x{self[0]:08x}, 0x{self[1]:08x})"


def _rotl(x: int, d: int) -> int:
    return ((x << d) | (x >> (32 - d))) & _MASK


def threefry2x32(key, x0: int, x1: int) -> Tuple[int, int]:
    """One Threefry-2x32-20 block (Salmon, Moraes, Dror, Shaw, SC'11)."""
    k0, k1 = int(key[0]) & _MASK, int(key[1]) & _MASK
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    x0 = (x0 + ks[0]) & _MASK
    x1 = (x1 + ks[1]) & _MASK
    for i in range(5):
        for r in rot[i % 2]:
            x0 = (x0 + x1) & _MASK
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(i + 1) % 3]) & _MASK
        x1 = (x1 + ks[(i + 2) % 3] + i + 1) & _MASK
    return x0, x1


def _jax_bits(key, n: int):
    """jax's random_bits for counters 0..n-1: halves in, concatenated halves out (odd n: one padding counter 0)."""
    half = (n + 1) // 2
    lo, hi = [], []
    for b in range(half):
        y0, y1 = threefry2x32(key, b, b + half if b + half < n else 0)
        lo.append(y0)
        hi.append(y1)
    return (lo + hi)[:n]


def _jax_split(key: "JaxKey") -> Tuple["JaxKey", "JaxKey"]:
    b = _jax_bits(key, 4)  # jax.random.split(key, 2): bits reshaped (2, 2)
    return JaxKey(b[0], b[1]), JaxKey(b[2], b[3])


def _jax_uniform01(key: "JaxKey") -> float:
    """float(jax.random.uniform(key)): 23 mantissa bits, [0, 1)."""
    return (_jax_bits(key, 1)[0] >> 9) * (1.0 / 8388608.0)


def split(key: PRNGKey) -> Tuple[PRNGKey, PRNGKey]:
    """Return (new_key, subkey), the analogue of ``jax.random.split(key)`` (and exactly it for a JaxKey)."""
    if isinstance(key, JaxKey):
        return _jax_split(key)
    o = philox4x32((0, 0, 0, STREAM_SPLIT), key)
    return PRNGKey(o[0], o[1]), PRNGKey(o[2], o[3])


def fold_in(key: PRNGKey, data: int) -> PRNGKey:
    o = philox4x32((int(data) & _MASK, (int(data) >> 32) & _MASK, 1, STREAM_SPLIT), key)
    return PRNGKey(o[0], o[1])


def uniform(key: PRNGKey, counter: int = 0) -> float:
    """One (0,1) uniform from ``key`` -- the host twin of the device STREAM_UNIFORM draw
    (``jax.random.uniform(key)`` for a JaxKey; ``2 u - 1`` of it is jax's ``uniform(key, minval=-1, maxval=1)``
    exactly, both being exact in fp32)."""
    if isinstance(key, JaxKey):
        return _jax_uniform01(key)
    o = philox4x32((0, int(counter) & _MASK, 0, STREAM_UNIFORM), key)
    return u32_to_uniform(o[0])
