"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.

NumPy restatement of the parts of ``jax.random`` the reference's hot path draws from, so that the
two known answers in the reference's tests that depend on the JAX PRNG stream
(chiron/tests/test_mcmc.py:81-84 -- five Langevin energies; :451-452 -- 8 of 10 barostat moves
accepted) can pin the oracle and, through injected noise / patched draws, the HIP path.

Third-party dependency, absent from /root/reference: ``jax`` (version UNPINNED in the reference's
devtools/conda-envs/test_env.yaml:10).  Restated from the published algorithms:

* Threefry-2x32, 20 rounds -- Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as
  1, 2, 3" (SC'11); rotation constants (13,15,26,6 / 17,29,16,24), key-schedule parity 0x1BD11BDA.
* ``jax.random.PRNGKey(seed)`` = (seed >> 32, seed & 0xFFFFFFFF); ``split`` and ``random_bits`` of
  the classic (non-"partitionable") threefry implementation, the default of every JAX release
  contemporary with the reference: the counter array 0..n-1 (padded to even length) is split in
  HALVES, the halves are the two Threefry input words, and the two output halves are concatenated.
* ``uniform``: 23 mantissa bits ``bits >> 9 | 0x3F800000`` viewed as float32 minus 1, then
  ``max(minval, u * (maxval - minval) + minval)`` in float32.
* ``normal``: ``sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1))`` with the single-precision
  polynomial of M. Giles, "Approximating the erfinv function" (GPU Computing Gems, 2010), which is
  what XLA evaluates for float32.

Both reference literals are reproduced (tests/test_oracle_golden.py), which is the evidence that
this restatement matches the stream the reference was run with.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

u32 = np.uint32
f32 = np.float32
Key = Tuple[int, int]


def PRNGKey(seed: int) -> Key:
    seed = int(seed)
    return ((seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF)


def _rotl(x, d):
    return ((x << u32(d)) | (x >> u32(32 - d))).astype(u32)


def threefry2x32(key: Key, x0, x1):
    """One Threefry-2x32-20 block per element of the uint32 arrays ``x0``, ``x1``."""
    with np.errstate(over="ignore"):
        k0, k1 = u32(key[0]), u32(key[1])
        ks = (k0, k1, u32(k0 ^ k1 ^ u32(0x1BD11BDA)))
        rot = ((13, 15, 26, 6), (17, 29, 16, 24))
        x0 = (np.asarray(x0, dtype=u32) + ks[0]).astype(u32)
        x1 = (np.asarray(x1, dtype=u32) + ks[1]).astype(u32)
        for i in range(5):
            for r in rot[i % 2]:
                x0 = (x0 + x1).astype(u32)
                x1 = _rotl(x1, r) ^ x0
            x0 = (x0 + ks[(i + 1) % 3]).astype(u32)
            x1 = (x1 + ks[(i + 2) % 3] + u32(i + 1)).astype(u32)
    return x0, x1


def _threefry_2x32(key: Key, count) -> np.ndarray:
    """jax._src.prng.threefry_2x32: halves of ``count`` in, concatenated halves out."""
    count = np.asarray(count, dtype=u32).ravel()
    odd = count.size % 2
    if odd:
        count = np.concatenate([count, np.zeros(1, u32)])
    h = count.size // 2
    y0, y1 = threefry2x32(key, count[:h], count[h:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def split(key: Key, num: int = 2) -> List[Key]:
    """``jax.random.split(key, num)`` -> list of ``num`` keys (``key, subkey = split(key)``)."""
    bits = _threefry_2x32(key, np.arange(2 * num, dtype=u32)).reshape(num, 2)
    return [(int(a), int(b)) for a, b in bits]


def random_bits(key: Key, n: int) -> np.ndarray:
    return _threefry_2x32(key, np.arange(n, dtype=u32))


def uniform(key: Key, shape=(), minval=0.0, maxval=1.0) -> np.ndarray:
    """``jax.random.uniform(key, shape, float32, minval, maxval)``."""
    n = int(np.prod(shape)) if shape != () else 1
    bits = random_bits(key, n)
    floats = ((bits >> u32(9)) | f32(1.0).view(u32)).view(f32) - f32(1.0)
    lo, hi = f32(minval), f32(maxval)
    out = np.maximum(lo, (floats * f32(hi - lo) + lo).astype(f32))
    return out.reshape(shape) if shape != () else out[0]


def erfinv32(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=f32)
    w = (-np.log(((f32(1) - x) * (f32(1) + x)).astype(f32))).astype(f32)

    def horner(t, coeffs):
        p = np.full_like(t, f32(coeffs[0]))
        for c in coeffs[1:]:
            p = (f32(c) + p * t).astype(f32)
        return p

    small = horner((w - f32(2.5)).astype(f32),
                   (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
                    -0.00125372503, -0.00417768164, 0.246640727, 1.50140941))
    with np.errstate(invalid="ignore"):
        large = horner((np.sqrt(w).astype(f32) - f32(3)).astype(f32),
                       (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
                        -0.0076224613, 0.00943887047, 1.00167406, 2.83297682))
    return (np.where(w < f32(5), small, large).astype(f32) * x).astype(f32)


def normal(key: Key, shape) -> np.ndarray:
    """``jax.random.normal(key, shape, float32)``."""
    lo = np.nextafter(f32(-1), f32(0), dtype=f32)
    u = uniform(key, tuple(shape), lo, 1.0)
    return (f32(np.sqrt(2)) * erfinv32(u)).astype(f32)
