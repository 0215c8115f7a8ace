"""ctypes wrapper of oracle/_build/liboracle.so (TEST INFRASTRUCTURE / cpu_baseline only)."""
import ctypes as C
import os

import numpy as np

from . import build_oracle


class OrcBox(C.Structure):
    _fields_ = [("periodic", C.c_int), ("lx", C.c_float), ("ly", C.c_float), ("lz", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # always through build(): it compares the hash baked into the binary with oracle.c and recompiles on any
        # difference (a stale checker must never load)
        _lib = C.CDLL(build_oracle.build())
        _lib.orc_source_hash.restype = C.c_char_p
        assert _lib.orc_source_hash().decode().endswith(build_oracle.source_hash()), "liboracle.so is stale"
        _lib.orc_full_rows_cell.restype = C.c_int
        _lib.orc_lj_cell_f64.restype = C.c_int
        _lib.orc_lj_cell_f64_ex.restype = C.c_int
        _lib.orc_nl_build_fixed.restype = C.c_int
        _lib.orc_lj_nl.restype = C.c_float
        _lib.orc_langevin_run.restype = C.c_int
        _lib.orc_langevin_run_cell_omp.restype = C.c_int
    return _lib


def _box(box, periodic=True):
    if box is None or not periodic:
        return OrcBox(0, 0, 0, 0)
    b = np.asarray(box, dtype=np.float32)
    return OrcBox(1, float(b[0, 0]), float(b[1, 1]), float(b[2, 2]))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def nl_build_fixed(x, box, cutoff_and_skin, m, periodic=True):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    nl = np.zeros((n, m), np.uint32)
    mask = np.zeros((n, m), np.int32)
    nnbr = np.zeros(n, np.int32)
    lib().orc_nl_build_fixed(_p(x, C.c_float), n, m, _box(box, periodic), C.c_float(cutoff_and_skin),
                             _p(nl, C.c_uint32), _p(mask, C.c_int32), _p(nnbr, C.c_int32))
    return mask, nl, nnbr


def lj_nl(x, nl, mask, box, sigma, eps, cutoff, periodic=True):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, m = nl.shape
    f = np.zeros((n, 3), np.float32)
    e = lib().orc_lj_nl(_p(x, C.c_float), _p(np.ascontiguousarray(nl), C.c_uint32), _p(np.ascontiguousarray(mask), C.c_int32),
                        n, m, _box(box, periodic), C.c_float(sigma), C.c_float(eps), C.c_float(cutoff), _p(f, C.c_float))
    return float(e), f


def langevin_run(x, v, box, mass, sigma, eps, cutoff, skin, m0, dt, a, b, kT, n_steps, noise=None, key=(0, 0)):
    x = np.ascontiguousarray(x, dtype=np.float32).copy()
    v = np.ascontiguousarray(v, dtype=np.float32).copy()
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    stats = np.zeros(3, np.float64)
    nz = None if noise is None else _p(np.ascontiguousarray(noise, dtype=np.float32), C.c_float)
    rc = lib().orc_langevin_run(_p(x, C.c_float), _p(v, C.c_float), x.shape[0], _box(box), _p(mass, C.c_float),
                                C.c_float(sigma), C.c_float(eps), C.c_float(cutoff), C.c_float(skin), int(m0),
                                C.c_float(dt), C.c_float(a), C.c_float(b), C.c_float(kT), int(n_steps), nz,
                                C.c_uint32(key[0]), C.c_uint32(key[1]), _p(stats, C.c_double))
    if rc != 0:
        raise RuntimeError("orc_langevin_run failed")
    return x, v, dict(rebuilds=int(stats[0]), n_max=int(stats[1]), build_seconds=float(stats[2]))


def effective_cores() -> int:
    """CPU cores this process may really use: the affinity mask, capped by the cgroup CPU quota
    (the containers here run under a CFS quota far below the visible core count)."""
    n = len(os.sched_getaffinity(0))
    try:  # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
        return n
    except (OSError, ValueError):
        pass
    try:  # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            quota = int(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = int(fh.read())
        if quota > 0 and period > 0:
            n = max(1, min(n, int(quota / period + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def langevin_run_cell_omp(x, v, box, mass, sigma, eps, cutoff, skin, dt, a, b, kT, n_steps, noise=None, key=(0, 0),
                          threads=0, cap0=96):
    """Cell-list + OpenMP variant of langevin_run: same trajectory bits, O(N) builds, `threads` cores (0 = OpenMP default)."""
    x = np.ascontiguousarray(x, dtype=np.float32).copy()
    v = np.ascontiguousarray(v, dtype=np.float32).copy()
    mass = np.ascontiguousarray(mass, dtype=np.float32)
    stats = np.zeros(4, np.float64)
    nz = None if noise is None else _p(np.ascontiguousarray(noise, dtype=np.float32), C.c_float)
    rc = lib().orc_langevin_run_cell_omp(_p(x, C.c_float), _p(v, C.c_float), x.shape[0], _box(box), _p(mass, C.c_float),
                                         C.c_float(sigma), C.c_float(eps), C.c_float(cutoff), C.c_float(skin), int(cap0),
                                         C.c_float(dt), C.c_float(a), C.c_float(b), C.c_float(kT), int(n_steps), nz,
                                         C.c_uint32(key[0]), C.c_uint32(key[1]), int(threads), _p(stats, C.c_double))
    if rc != 0:
        raise RuntimeError("orc_langevin_run_cell_omp failed")
    return x, v, dict(rebuilds=int(stats[0]), row_capacity=int(stats[1]), build_seconds=float(stats[2]), threads=int(stats[3]))


def full_rows_cell(x, box, cutoff_and_skin, threads=0, cap0=96):
    """Every particle's FULL row (all partners with d < cutoff + skin in the reference's fp32 arithmetic, ascending id)
    through the oracle's own cell grid: (rows (n, cap) uint32, deg (n,) int32).  Half row i of the reference
    (chiron/neighbors.py:548-626) = the entries of full row i that are > i."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, cap = x.shape[0], int(cap0)
    while True:
        rows = np.zeros((n, cap), np.uint32)
        deg = np.zeros(n, np.int32)
        mx = lib().orc_full_rows_cell(_p(x, C.c_float), n, _box(box), C.c_float(cutoff_and_skin), cap, _p(rows, C.c_uint32),
                                      _p(deg, C.c_int32), int(threads))
        if mx <= cap:
            return rows, deg
        cap = mx + 16


def half_list_from_full_rows(rows, deg):
    """(n_neighbors (n,), flat ascending neighbor ids row after row) of the reference's half list."""
    n, cap = rows.shape
    valid = np.arange(cap)[None, :] < deg[:, None]
    half = valid & (rows > np.arange(n, dtype=np.uint32)[:, None])
    return half.sum(axis=1).astype(np.int32), rows[half]


def lj_cell_f64(x, box, sigma, eps, cutoff, threads=0, cap0=64):
    """fp64 LJ energy + forces on ALL particles (interacting set = the reference's fp32 `d < cutoff`, values fp64):
    (E, force64 (n,3), sum|e_ij|, max_i max_c sum_j|f_ij,c|, n_pairs)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, cap = x.shape[0], int(cap0)
    f = np.zeros((n, 3), np.float64)
    stats = np.zeros(4, np.float64)
    while True:
        rc = lib().orc_lj_cell_f64(_p(x, C.c_float), n, _box(box), C.c_float(sigma), C.c_float(eps), C.c_float(cutoff), cap,
                                   _p(f, C.c_double), _p(stats, C.c_double), int(threads))
        if rc == 0:
            return float(stats[0]), f, float(stats[1]), float(stats[2]), int(stats[3])
        if rc < 0:
            raise MemoryError("orc_lj_cell_f64")
        cap = rc + 16


def lj_cell_f64_ex(x, box, sigma, eps, cutoff, threads=0, cap0=64, geom32=True):
    """lj_cell_f64 plus per-particle noise scales:
    (E, force64 (n,3), sum|e_ij|, fabs (n,3) = sum_j |f_ij,c|, fterm (n,3) = sum_j (|repulsive| + |attractive|)_c, n_pairs).
    geom32: pair vectors in the reference's fp32 arithmetic (chiron/neighbors.py:69-79; what the kernels reproduce bit
    for bit), Lennard-Jones values and sums in fp64 -- see oracle.c for why the clustered fluid needs it."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, cap = x.shape[0], int(cap0)
    f = np.zeros((n, 3), np.float64)
    fa = np.zeros((n, 3), np.float64)
    ft = np.zeros((n, 3), np.float64)
    stats = np.zeros(4, np.float64)
    while True:
        rc = lib().orc_lj_cell_f64_ex(_p(x, C.c_float), n, _box(box), C.c_float(sigma), C.c_float(eps), C.c_float(cutoff), cap,
                                      _p(f, C.c_double), _p(fa, C.c_double), _p(ft, C.c_double), _p(stats, C.c_double),
                                      int(threads), int(bool(geom32)))
        if rc == 0:
            return float(stats[0]), f, float(stats[1]), fa, ft, int(stats[3])
        if rc < 0:
            raise MemoryError("orc_lj_cell_f64_ex")
        cap = rc + 16
