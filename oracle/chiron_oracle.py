"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.

A NumPy restatement of the arithmetic on chiron's Lennard-Jones / neighbor-list /
Langevin / Metropolis hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker (or as the timed CPU baseline); nothing under ``chiron_amd/`` imports it.

Every function cites the reference lines it follows (paths relative to the
reference checkout, ``chiron/...``).  All fp32 functions keep the reference's
operation order, one rounding per operation (NumPy never contracts a*b+c into
an FMA), so that neighbor indices and masks can be compared bit-for-bit with the
HIP kernels (compiled with -ffp-contract=off).

PINNING.  The reference is Python-on-JAX; ``jax``/``openmm`` are not installable
in this image, so the reference cannot be run here and no fixtures could be
generated from it.  This oracle is pinned instead by the literal known answers in
the reference's own tests (transcribed as data in ``tests/golden/``):
test_pairs.py:26-57,102-160,275-318,348-405,441-485 (bit-exact indices/masks/
distances), test_potential.py:33-55,177-230 and test_testsystems.py:144-165
(energies/forces).  Whether XLA-CPU contracts x*x+y*y into an FMA in
``jnp.linalg.norm`` cannot be checked here, so agreement with a live JAX run at
1-ulp cutoff-boundary pairs is UNPINNED (expected <~1 flip per 1e7 pairs).
The two literals that depend on JAX's threefry stream ARE pinned: with the stream
restated in ``oracle/jax_prng.py`` (Threefry-2x32 + jax.random's split / uniform /
normal), ``baoa_half`` / ``kick`` / ``initial_velocities`` reproduce the five
Langevin energies of test_mcmc.py:81-84 to the reference's own tolerance (rtol
1e-5; observed 3e-6, a constant-factor offset consistent with an element-mass or
gas-constant digit in the reference's unpinned openmm) and ``barostat_propose`` /
``barostat_log_ratio`` / ``metropolis_accept`` reproduce the 8-of-10 acceptance of
test_mcmc.py:451-452 (tests/test_oracle_golden.py).  What stays UNPINNED is the
product's own Philox stream (a different generator by design).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

f32 = np.float32

# physical constants exactly as openmm.unit ships them
KB = 1.380649e-23  # J/K
NA = 6.02214076e23  # 1/mol
R_KJ = KB * NA * 1e-3  # kJ/mol/K


def _box_lengths(box) -> np.ndarray:
    box = np.asarray(box, dtype=f32)
    if box.shape == (3, 3):
        return np.array([box[0, 0], box[1, 1], box[2, 2]], dtype=f32)
    return box.astype(f32)


# ---------------------------------------------------------------------------
# spaces  (chiron/neighbors.py:39-175)
# ---------------------------------------------------------------------------
def norm3(r: np.ndarray) -> np.ndarray:
    """jnp.linalg.norm(r, axis=-1) in fp32: sqrt((x*x + y*y) + z*z)."""
    rx, ry, rz = r[..., 0], r[..., 1], r[..., 2]
    return np.sqrt((rx * rx + ry * ry) + rz * rz)


def periodic_displacement(x1, x2, box) -> Tuple[np.ndarray, np.ndarray]:
    """OrthogonalPeriodicSpace.displacement, chiron/neighbors.py:45-83."""
    if box is None:
        raise ValueError("box_vectors must be provided for a periodic system")
    x1 = np.asarray(x1, dtype=f32)
    x2 = np.asarray(x2, dtype=f32)
    L = _box_lengths(box)
    r = x1 - x2  # :69
    r = np.mod(r + L * f32(0.5), L) - L * f32(0.5)  # :79 (jnp.mod == floor-mod)
    return r.astype(f32), norm3(r).astype(f32)


def periodic_wrap(x, box) -> np.ndarray:
    """OrthogonalPeriodicSpace.wrap, chiron/neighbors.py:85-112."""
    if box is None:
        raise ValueError("box_vectors must be provided for a periodic system")
    x = np.asarray(x, dtype=f32)
    L = _box_lengths(box)
    return (x - np.floor(x / L) * L).astype(f32)  # :110


def nonperiodic_displacement(x1, x2, box=None) -> Tuple[np.ndarray, np.ndarray]:
    """OrthogonalNonPeriodicSpace.displacement, chiron/neighbors.py:116-150."""
    x1 = np.asarray(x1, dtype=f32)
    x2 = np.asarray(x2, dtype=f32)
    r = x1 - x2
    return r, norm3(r)


def displacement(x1, x2, box, periodic: bool):
    return periodic_displacement(x1, x2, box) if periodic else nonperiodic_displacement(x1, x2)


# ---------------------------------------------------------------------------
# NeighborListNsqrd  (chiron/neighbors.py:378-907)
# ---------------------------------------------------------------------------
def neighborlist_build_fixed(x, box, cutoff_and_skin, n_max, periodic=True, row_block=256):
    """One pass of vmap(_build_neighborlist), chiron/neighbors.py:548-626 / :694-704.

    Returns (neighbor_mask[N,M] i32, neighbor_list[N,M] u32, n_neighbors[N] i32).
    """
    x = np.asarray(x, dtype=f32)
    n = x.shape[0]
    cs = f32(cutoff_and_skin)
    nl = np.zeros((n, n_max), dtype=np.uint32)
    n_nbr = np.zeros(n, dtype=np.int32)
    ids = np.arange(n)
    for i0 in range(0, n, row_block):
        i1 = min(n, i0 + row_block)
        _, d = displacement(x[i0:i1, None, :], x[None, :, :], box, periodic)  # :595
        red = ids[i0:i1, None] < ids[None, :]  # _pairs_mask :542
        mask = (d < cs) & red  # :600-602
        for r in range(i1 - i0):
            i = i0 + r
            idx = np.nonzero(mask[r])[0]
            fill = int(idx[0]) if idx.size else 0  # argmax of a bool vector :606
            if fill == i:  # :609
                fill += 1
            n_nbr[i] = idx.size  # :612
            row = np.full(n_max, fill, dtype=np.uint32)  # :617-620
            k = min(idx.size, n_max)
            row[:k] = idx[:k]
            nl[i] = row
    nl_mask = (np.arange(n_max)[None, :] < n_nbr[:, None]).astype(np.int32)  # :623
    return nl_mask, nl, n_nbr


def neighborlist_build(x, box, cutoff, skin, n_max, periodic=True):
    """NeighborListNsqrd.build including the growth loop, chiron/neighbors.py:628-729.

    ``cutoff + skin`` is formed in double precision by the unit layer and cast to
    fp32 when it enters the jitted function (:674, :702).
    Returns (neighbor_mask, neighbor_list, n_neighbors, n_max_neighbors).
    """
    cs = float(cutoff) + float(skin)
    nl_mask, nl, n_nbr = neighborlist_build_fixed(x, box, cs, n_max, periodic)
    while np.any(n_nbr == n_max):  # :709  (equality, as in the reference)
        n_max = int(n_nbr.max() + 10)  # :713
        nl_mask, nl, n_nbr = neighborlist_build_fixed(x, box, cs, n_max, periodic)
    return nl_mask, nl, n_nbr, n_max


def neighborlist_calculate(x, nl, nl_mask, box, cutoff, periodic=True):
    """NeighborListNsqrd.calculate, chiron/neighbors.py:731-826.

    Returns (n_pairs[N] i32, mask[N,M] i32, dist[N,M] f32, r_ij[N,M,3] f32).
    """
    x = np.asarray(x, dtype=f32)
    xi = x[:, None, :]
    xj = x[nl.astype(np.int64)]  # :778
    r, d = displacement(np.broadcast_to(xi, xj.shape), xj, box, periodic)
    mask = ((d < f32(cutoff)) & (nl_mask != 0)).astype(np.int32)  # :782
    return mask.sum(axis=1).astype(np.int32), mask, d, r


def neighborlist_check(x, ref, box, skin, periodic=True) -> bool:
    """NeighborListNsqrd.check, chiron/neighbors.py:828-907 (>= skin/2, :868)."""
    x = np.asarray(x, dtype=f32)
    ref = np.asarray(ref, dtype=f32)
    if ref.shape[0] != x.shape[0]:  # :890
        return True
    _, d = displacement(x, ref, box, periodic)
    return bool(np.any(d >= f32(skin) / f32(2.0)))


# ---------------------------------------------------------------------------
# PairListNsqrd  (chiron/neighbors.py:910-1289)
# ---------------------------------------------------------------------------
def pairlist_build(n: int):
    """_pairs_and_mask, chiron/neighbors.py:1018-1067: all j != i ascending, mask i<j."""
    ids = np.arange(n, dtype=np.uint32)
    all_pairs = np.empty((n, max(n - 1, 0)), dtype=np.uint32)
    for i in range(n):
        all_pairs[i] = np.concatenate([ids[:i], ids[i + 1 :]])
    reduction_mask = ids[:, None] < all_pairs
    return all_pairs, reduction_mask


def pairlist_calculate(x, all_pairs, reduction_mask, box, cutoff: Optional[float], periodic=True):
    """PairListNsqrd.calculate, chiron/neighbors.py:1106-1270."""
    x = np.asarray(x, dtype=f32)
    xj = x[all_pairs.astype(np.int64)]
    xi = np.broadcast_to(x[:, None, :], xj.shape)
    r, d = displacement(xi, xj, box, periodic)
    if cutoff is None:
        mask = reduction_mask.astype(np.int32)  # :1211
    else:
        mask = ((d < f32(cutoff)) & reduction_mask).astype(np.int32)  # :1156
    return mask.sum(axis=1).astype(np.int32), mask, d, r


def nopbc_pairlist(x, cutoff):
    """NeuralNetworkPotential.compute_pairlist, chiron/potential.py:26-63 (no PBC)."""
    x = np.asarray(x, dtype=f32)
    n = x.shape[0]
    pids = np.arange(n)
    p1, p2 = np.meshgrid(pids, pids)  # :50
    p1 = p1.flatten()
    p2 = p2.flatten()
    sel = np.where(p1 < p2)
    p1, p2 = p1[sel], p2[sel]
    r = x[p1] - x[p2]  # :57
    d = norm3(r)
    inter = np.where(d < f32(cutoff))  # :59
    return d[inter], r[inter], np.stack((p1[inter], p2[inter]), axis=0)


# ---------------------------------------------------------------------------
# LJ energy / force  (chiron/potential.py:130-332)
# ---------------------------------------------------------------------------
def ipow(x: np.ndarray, n: int) -> np.ndarray:
    """x**n for a Python int n the way XLA lowers lax.integer_pow (square & multiply)."""
    acc = None
    while n:
        if n & 1:
            acc = x if acc is None else acc * x
        n >>= 1
        if n:
            x = x * x
    return acc


def lj_pair_energy_f32(d, sigma, epsilon):
    """4*eps*((sigma/d)**12 - (sigma/d)**6), chiron/potential.py:208-212, fp32."""
    s = f32(sigma) / d
    return f32(4.0 * float(epsilon)) * (ipow(s, 12) - ipow(s, 6))


def lj_energy_masked_f32(dist, mask, sigma, epsilon):
    """vmap(_compute_energy_masked)(dist, mask).sum(), chiron/potential.py:193-213,276-279."""
    with np.errstate(divide="ignore", invalid="ignore"):
        e = mask.astype(f32) * lj_pair_energy_f32(dist.astype(f32), sigma, epsilon)
    return e.sum(axis=-1, dtype=f32).sum(dtype=f32)


def lj_energy_nbrlist_f32(x, nl, nl_mask, box, sigma, epsilon, cutoff, periodic=True):
    """LJPotential.compute_energy, neighbor-list branch, chiron/potential.py:260-279."""
    _, mask, d, _ = neighborlist_calculate(x, nl, nl_mask, box, cutoff, periodic)
    return lj_energy_masked_f32(d, mask, sigma, epsilon)


def lj_energy_nopbc_f32(x, sigma, epsilon, cutoff):
    """LJPotential.compute_energy, nbr_list=None branch, chiron/potential.py:235-258."""
    d, _, _ = nopbc_pairlist(x, cutoff)
    if d.shape[0] == 0:
        return f32(0.0)
    return lj_pair_energy_f32(d, sigma, epsilon).sum(dtype=f32)


def lj_pair_force_f32(d, r, sigma, epsilon):
    """Closed-form pair force, chiron/potential.py:322-326 (fp32, reference op order)."""
    s = f32(sigma) / d
    return (f32(24.0) * (f32(epsilon) / (d * d)) * (f32(2.0) * ipow(s, 12) - ipow(s, 6)))[..., None] * r


def lj_force_from_pairs_f32(n, pairs_i, pairs_j, d, r, sigma, epsilon):
    """Scatter +f to i, -f to j (chiron/potential.py:328-332): what -jax.grad produces."""
    f = lj_pair_force_f32(d, r, sigma, epsilon).astype(f32)
    out = np.zeros((n, 3), dtype=f32)
    np.add.at(out, pairs_i, f)
    np.add.at(out, pairs_j, -f)
    return out


def lj_force_nbrlist_f32(x, nl, nl_mask, box, sigma, epsilon, cutoff, periodic=True):
    """-grad of the neighbor-list energy == analytic pair forces over the masked list."""
    n_pairs, mask, d, r = neighborlist_calculate(x, nl, nl_mask, box, cutoff, periodic)
    ii, kk = np.nonzero(mask)
    jj = nl[ii, kk].astype(np.int64)
    return lj_force_from_pairs_f32(x.shape[0], ii, jj, d[ii, kk], r[ii, kk], sigma, epsilon)


def lj_force_nopbc_f32(x, sigma, epsilon, cutoff):
    d, r, pairs = nopbc_pairlist(x, cutoff)
    return lj_force_from_pairs_f32(np.asarray(x).shape[0], pairs[0], pairs[1], d, r, sigma, epsilon)


def lj_bruteforce_f64(x, box, sigma, epsilon, cutoff, periodic=True, row_block=512,
                      pair_filter=None):
    """fp64 all-pairs truncated LJ with minimum image (accuracy yardstick).

    Returns (energy, forces[N,3], sum_abs_pair_energy, max_i sum_j |f_ij|).
    ``pair_filter(i_idx, j_idx) -> bool array`` optionally restricts the pair set.
    """
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    L = None if not periodic else _box_lengths(box).astype(np.float64)
    energy = 0.0
    eabs = 0.0
    force = np.zeros((n, 3))
    fabs = np.zeros(n)
    for i0 in range(0, n, row_block):
        i1 = min(n, i0 + row_block)
        r = x[i0:i1, None, :] - x[None, :, :]
        if periodic:
            r = r - L * np.round(r / L)
        d2 = (r * r).sum(-1)
        ii = np.arange(i0, i1)[:, None]
        jj = np.arange(n)[None, :]
        sel = (ii != jj) & (d2 < float(cutoff) ** 2)
        if pair_filter is not None:
            sel &= pair_filter(ii, jj)
        d2s = np.where(sel, d2, 1.0)
        s2 = float(sigma) ** 2 / d2s
        s6 = s2 ** 3
        e = np.where(sel, 4.0 * float(epsilon) * (s6 * s6 - s6), 0.0)
        fs = np.where(sel, 24.0 * float(epsilon) * (2.0 * s6 * s6 - s6) / d2s, 0.0)
        energy += 0.5 * e.sum()
        eabs += 0.5 * np.abs(e).sum()
        force[i0:i1] += (fs[..., None] * r).sum(1)
        fabs[i0:i1] += (np.abs(fs) * np.sqrt(d2s)).sum(1)
    return energy, force, eabs, float(fabs.max()) if n else 0.0


# ---------------------------------------------------------------------------
# Philox4x32-10 + normals (the build's own device stream; see chiron_amd/rng.py)
# ---------------------------------------------------------------------------
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_SH = np.uint64(32)

STREAM_VELOCITY = 0
STREAM_LANGEVIN = 1
STREAM_MC_DISPLACE = 2
STREAM_UNIFORM = 3


def philox4x32(c0, c1, c2, c3, k0: int, k1: int):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _SH, p0 & _MASK
        hi1, lo1 = p1 >> _SH, p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u32_to_uniform(u: np.ndarray) -> np.ndarray:
    return ((u >> np.uint32(9)).astype(f32) + f32(0.5)) * f32(1.0 / 8388608.0)


def philox_normals3(n: int, step: int, stream: int, key) -> np.ndarray:
    """3 standard normals per particle index 0..n-1 (Box-Muller on one Philox block).

    (z0, z1) = r01*(cos, sin)(2*pi*u1) with r01 = sqrt(-2 ln u0); z2 = r23*cos(2*pi*u3).
    Same construction as csrc/chx_rng.h; fp32 throughout.
    """
    i = np.arange(n, dtype=np.uint64)
    o = philox4x32(i, np.uint64(step & 0xFFFFFFFF), np.uint64(0), np.uint64(stream), int(key[0]), int(key[1]))
    u = [u32_to_uniform(w) for w in o]
    two_pi = f32(6.283185307179586)
    r01 = np.sqrt(f32(-2.0) * np.log(u[0]))
    r23 = np.sqrt(f32(-2.0) * np.log(u[2]))
    t1 = two_pi * u[1]
    t3 = two_pi * u[3]
    z = np.stack([r01 * np.cos(t1), r01 * np.sin(t1), r23 * np.cos(t3)], axis=1)
    return z.astype(f32)


# ---------------------------------------------------------------------------
# Langevin BAOAB  (chiron/integrators.py:126-137, :179-195) and velocities
# ---------------------------------------------------------------------------
def langevin_constants(temperature_K, timestep_ps, collision_rate_per_ps):
    """kT, a, b as the reference forms them (chiron/integrators.py:127-137)."""
    kT = R_KJ * float(temperature_K)
    a = np.exp(f32(-collision_rate_per_ps * timestep_ps))
    b = np.sqrt(f32(1.0) - np.exp(f32(-2.0 * collision_rate_per_ps * timestep_ps)))
    return kT, f32(a), f32(b)


def baoa_half(x, v, F, mass, dt, a, b, kT, noise):
    """B, A, O, A of one step (chiron/integrators.py:181-188), fp32, reference op order."""
    hdt = f32(dt * 0.5)
    m = np.asarray(mass, dtype=f32)[:, None]
    sigma_v = np.sqrt(f32(kT) / m)  # :131
    v = v + hdt * F / m  # :181
    x = x + hdt * v  # :183
    v = (f32(a) * v) + (f32(b) * sigma_v * noise)  # :186
    x = x + hdt * v  # :188
    return x.astype(f32), v.astype(f32)


def kick(v, F, mass, dt):
    """v += (dt/2) F / m  (chiron/integrators.py:195)."""
    m = np.asarray(mass, dtype=f32)[:, None]
    return (v + f32(dt * 0.5) * F / m).astype(f32)


def initial_velocities(mass, kT, normals):
    """sigma_v * N(0,1), chiron/utils.py:134-142."""
    m = np.asarray(mass, dtype=f32)[:, None]
    return (np.sqrt(f32(kT) / m) * normals).astype(f32)


def langevin_run(x, v, box, mass, sigma, epsilon, cutoff, skin, n_max, dt, gamma, temperature,
                 n_steps, noise_fn, use_nbr_list=True):
    """LangevinIntegrator.run loop with a NeighborListNsqrd, chiron/integrators.py:168-195.

    ``noise_fn(step) -> (N,3) fp32`` supplies the O-step noise.  Returns
    (x, v, n_rebuilds, final list tuple).
    """
    kT, a, b = langevin_constants(temperature, dt, gamma)
    x = np.asarray(x, dtype=f32).copy()
    v = np.asarray(v, dtype=f32).copy()
    n = x.shape[0]
    rebuilds = 0
    if use_nbr_list:
        nl_mask, nl, n_nbr, n_max = neighborlist_build(x, box, cutoff, skin, n_max)
        ref = x.copy()
        F = lj_force_nbrlist_f32(x, nl, nl_mask, box, sigma, epsilon, cutoff)
    else:
        nl_mask = nl = None
        F = lj_force_nopbc_f32(x, sigma, epsilon, cutoff)
    for step in range(n_steps):
        x, v = baoa_half(x, v, F, mass, dt, a, b, kT, noise_fn(step))
        if use_nbr_list:
            x = periodic_wrap(x, box)  # :239
            if neighborlist_check(x, ref, box, skin):  # :241
                nl_mask, nl, n_nbr, n_max = neighborlist_build(x, box, cutoff, skin, n_max)
                ref = x.copy()
                rebuilds += 1
            F = lj_force_nbrlist_f32(x, nl, nl_mask, box, sigma, epsilon, cutoff)
        else:
            F = lj_force_nopbc_f32(x, sigma, epsilon, cutoff)
        v = kick(v, F, mass, dt)
    return x, v, rebuilds, (nl_mask, nl, n_max)


# ---------------------------------------------------------------------------
# Monte Carlo  (chiron/mcmc.py:531-548, :724-777, :950-1000; chiron/states.py:307-325)
# ---------------------------------------------------------------------------
def mc_displace(x, sigma_d, normals, box=None, subset_mask=None):
    """x' = wrap(x + sigma_d * xi [* mask]), chiron/mcmc.py:733-755."""
    x = np.asarray(x, dtype=f32)
    d = normals.astype(f32) * f32(sigma_d)
    if subset_mask is not None:
        d = d * np.asarray(subset_mask, dtype=f32)[:, None]
    xp = x + d
    if box is not None:
        xp = periodic_wrap(xp, box)
    return xp.astype(f32)


def metropolis_accept(log_proposal_ratio: float, u: float) -> bool:
    """chiron/mcmc.py:544-548 (NaN handled by the caller, :428-429)."""
    return (-log_proposal_ratio <= 0.0) or (u < math.exp(log_proposal_ratio))


def reduced_potential(energy_kj_mol: float, temperature_K: float,
                      pressure_atm: Optional[float] = None, volume_nm3: Optional[float] = None) -> float:
    """beta*(U/N_A + pV), chiron/states.py:307-325, in SI doubles."""
    beta = 1.0 / (KB * temperature_K)
    red = energy_kj_mol * 1e3 / NA
    if pressure_atm is not None:
        red += pressure_atm * 101325.0 * volume_nm3 * 1e-27
    return beta * red


def barostat_propose(x, box, volume_max_scale: float, u01: float):
    """MonteCarloBarostatMove._propose geometry, chiron/mcmc.py:956-983.

    ``u01`` in (0,1) is mapped to uniform(-1,1).  Returns (x', box', V, V').
    """
    L = _box_lengths(box).astype(np.float64)
    v0 = float(L[0] * L[1] * L[2])
    dv = (2.0 * u01 - 1.0) * volume_max_scale * v0
    v1 = v0 + dv
    s = (v1 / v0) ** (1.0 / 3.0)
    return (np.asarray(x, dtype=f32) * f32(s)).astype(f32), (np.asarray(box, dtype=f32) * f32(s)).astype(f32), v0, v1


def barostat_log_ratio(u_new: float, u_old: float, n_atoms: int, v_new: float, v_old: float) -> float:
    """chiron/mcmc.py:998-1000."""
    return -(u_new - u_old) + n_atoms * math.log(v_new / v_old)


# ---------------------------------------------------------------------------
# Monte Carlo CHAINS: the trial loop itself (chiron/mcmc.py:243-306 update, :357-463 _step,
# :531-548 _accept_or_reject, :670-787 displacement move, :902-1009 barostat move), restated
# on this file's own Philox (key split / uniform twins of the product's counter-based key
# chain, derived here from philox4x32 above, NOT imported from chiron_amd.rng).
# ---------------------------------------------------------------------------
STREAM_SPLIT = 0x5EED


def key_split(key):
    """(new_key, subkey): what SamplerState.new_PRNG_key does to the stored key (chiron/states.py:150-154),
    on the Philox block (0, 0, 0, STREAM_SPLIT)."""
    o = philox4x32(0, 0, 0, STREAM_SPLIT, int(key[0]), int(key[1]))
    return (int(o[0]), int(o[1])), (int(o[2]), int(o[3]))


def key_uniform(key, counter: int = 0) -> float:
    """One (0,1) uniform of `key`: first word of the Philox block (0, counter, 0, STREAM_UNIFORM)."""
    o = philox4x32(0, counter, 0, STREAM_UNIFORM, int(key[0]), int(key[1]))
    return float(u32_to_uniform(np.asarray(o[0], dtype=np.uint32)))


class McList:
    """The neighbor-list state a chain carries: list arrays + the positions/box it was built from."""

    def __init__(self, x, box, cutoff, skin, n_max):
        self.cutoff, self.skin = cutoff, skin
        self.box = np.asarray(box, dtype=f32).copy()
        self.mask, self.nl, self.n_nbr, self.n_max = neighborlist_build(x, self.box, cutoff, skin, n_max)
        self.ref = np.asarray(x, dtype=f32).copy()
        self.builds = 1

    def rebuilt(self, x, box):
        new = McList(x, box, self.cutoff, self.skin, self.n_max)
        new.builds = self.builds + 1
        return new

    def energy(self, x, sigma, epsilon):
        return float(lj_energy_nbrlist_f32(x, self.nl, self.mask, self.box, sigma, epsilon, self.cutoff))


def mc_displacement_update(x, box, lst: McList, key, sigma, epsilon, temperature_K, sigma_d, n_moves, *,
                           pressure_atm=None, subset_mask=None, autotune=False, autotune_interval=100,
                           counters=(0, 0, 0), normals_fn=None, energy_fn=None):
    """One MonteCarloDisplacementMove.update(): n_moves Metropolis trials (chiron/mcmc.py:243-306).

    counters = (n_accepted, n_proposed, number_of_attempts_made) carried between calls (:465-486, :281).
    normals_fn(subkey) -> (N,3) fp32 replaces this file's Box-Muller draw (tests inject the device's
    normals to make the proposals bit-identical); energy_fn(x, lst) replaces the fp32 list energy.
    Returns (x, lst, key, sigma_d, counters, log) with per-trial log entries
    dict(accepted, energy, energy_proposed, log_ratio, uniform, rebuilt_list).
    """
    x = np.asarray(x, dtype=f32).copy()
    n = x.shape[0]
    n_acc, n_prop, attempts = counters
    volume = float(np.prod(_box_lengths(box).astype(np.float64))) if pressure_atm is not None else None
    efn = energy_fn or (lambda xx, ll: ll.energy(xx, sigma, epsilon))
    u_cur = None  # :271 -- recomputed at the first trial of every update()
    e_cur = None
    log = []
    for _ in range(n_moves):
        if u_cur is None:  # :396-406
            e_cur = efn(x, lst)
            u_cur = reduced_potential(e_cur, temperature_K, pressure_atm, volume)
        # ---- _propose (:680-787)
        key, sub = key_split(key)  # :724 new_PRNG_key
        z = normals_fn(sub) if normals_fn is not None else philox_normals3(n, 0, STREAM_MC_DISPLACE, sub)
        xp = mc_displace(x, sigma_d, z, box, subset_mask)  # :733-755 (wrap because a list is present)
        # :759 (a PairListNsqrd -- lst None here -- only reports a changed particle count, neighbors.py:1272-1289)
        rebuilt = lst is not None and bool(neighborlist_check(xp, lst.ref, lst.box, lst.skin))
        lst_p = lst.rebuilt(xp, box) if rebuilt else lst  # :761-770
        e_p = efn(xp, lst_p)
        u_p = reduced_potential(e_p, temperature_K, pressure_atm, volume)  # :772-774
        log_ratio = -u_p + u_cur  # :776
        # ---- _step (:428-461)
        key_p = key  # the proposed state is a deep copy taken AFTER the key advanced (:739)
        uni = None
        if math.isnan(u_p):
            decision = False
        else:
            key_p, sub2 = key_split(key_p)  # proposed_sampler_state.new_PRNG_key (:434)
            uni = key_uniform(sub2)
            decision = metropolis_accept(log_ratio, uni)
        n_acc += int(decision)  # :465-471
        n_prop += 1
        if decision:
            x, lst, u_cur, e_cur = xp, lst_p, u_p, e_p
        key = key_p  # accepted: the proposed state's key; rejected: copied over (:459-461)
        attempts += 1  # :281
        log.append(dict(accepted=bool(decision), energy=e_cur, energy_proposed=e_p, log_ratio=log_ratio, uniform=uni,
                        rebuilt_list=rebuilt, sigma_d=sigma_d))
        if autotune and attempts % autotune_interval == 0 and attempts > 0:  # :296-302, :670-678
            ratio = n_acc / n_prop
            if ratio > 0.6:
                sigma_d = sigma_d * 1.1
            elif ratio < 0.4:
                sigma_d = sigma_d / 1.1
    return x, lst, key, sigma_d, (n_acc, n_prop, attempts), log


def mc_barostat_update(x, box, key, temperature_K, pressure_atm, volume_max_scale, n_moves, energy_fn, *,
                       autotune=False, autotune_interval=100, counters=(0, 0, 0)):
    """One MonteCarloBarostatMove.update() (chiron/mcmc.py:243-306 with _propose :913-1009).

    energy_fn(x, box) -> potential energy in kJ/mol at the (scaled) configuration: the reference rebuilds a
    deep-copied list for every proposal (:985-990), so the energy is always over a fresh list.
    Returns (x, box, key, volume_max_scale, counters, log); log entries dict(accepted, volume, log_ratio, uniform).
    """
    x = np.asarray(x, dtype=f32).copy()
    box = np.asarray(box, dtype=f32).copy()
    n = x.shape[0]
    n_acc, n_prop, attempts = counters
    u_cur = None
    log = []
    for _ in range(n_moves):
        v0 = float(box[0, 0]) * float(box[1, 1]) * float(box[2, 2])  # :950-954 on the fp32 box entries
        if u_cur is None:
            u_cur = reduced_potential(energy_fn(x, box), temperature_K, pressure_atm, v0)
        key, sub = key_split(key)  # :943
        dv = (2.0 * key_uniform(sub) - 1.0) * (volume_max_scale * v0)  # :957-961
        v1 = v0 + dv  # :963
        s = (v1 / v0) ** (1.0 / 3.0)  # :966
        xp = (x * f32(s)).astype(f32)  # :971-973
        boxp = (box * f32(s)).astype(f32)  # :975-977
        vp = float(boxp[0, 0]) * float(boxp[1, 1]) * float(boxp[2, 2])  # states.py:315-319 reads the scaled box
        u_p = reduced_potential(energy_fn(xp, boxp), temperature_K, pressure_atm, vp)
        log_ratio = -(u_p - u_cur) + n * math.log(v1 / v0)  # :998-1000
        key_p = key
        uni = None
        if math.isnan(u_p):
            decision = False
        else:
            key_p, sub2 = key_split(key_p)
            uni = key_uniform(sub2)
            decision = metropolis_accept(log_ratio, uni)
        n_acc += int(decision)
        n_prop += 1
        if decision:
            x, box, u_cur = xp, boxp, u_p
        key = key_p
        attempts += 1
        log.append(dict(accepted=bool(decision), log_ratio=log_ratio, uniform=uni, max_volume_scale=volume_max_scale,
                        volume=float(box[0, 0]) * float(box[1, 1]) * float(box[2, 2])))
        if autotune and attempts % autotune_interval == 0 and attempts > 0:  # :902-911
            ratio = n_acc / n_prop
            if ratio < 0.25:
                volume_max_scale = volume_max_scale / 1.1
            elif ratio > 0.75:
                volume_max_scale = min(volume_max_scale * 1.1, 0.3)
    return x, box, key, volume_max_scale, (n_acc, n_prop, attempts), log


# ---------------------------------------------------------------------------
# synthetic LJ boxes -- shared by tests and bench
# ---------------------------------------------------------------------------
def lj_lattice(n: int, reduced_density: float, sigma: float, seed: int = 1234, jitter: float = 0.1):
    """Jittered simple-cubic lattice in a cubic box of edge (n sigma^3/rho*)^(1/3)."""
    L = (n * sigma ** 3 / reduced_density) ** (1.0 / 3.0)
    m = int(math.ceil(n ** (1.0 / 3.0) - 1e-9))
    a = L / m
    idx = np.arange(n)
    ix, iy, iz = idx % m, (idx // m) % m, idx // (m * m)
    rng = np.random.default_rng(seed)
    pos = (np.stack([ix, iy, iz], axis=1) + 0.5) * a + rng.uniform(-jitter, jitter, size=(n, 3)) * a
    pos = pos - np.floor(pos / L) * L
    box = np.diag([L, L, L]).astype(f32)
    return pos.astype(f32), box
