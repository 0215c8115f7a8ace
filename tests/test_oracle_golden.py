"""Pin the CPU oracle against every known answer the reference's tests hold."""
import math
import os

import numpy as np
import pytest

from oracle import chiron_oracle as O
from oracle import jax_prng as JX
from conftest import cube_coordinates

f32 = np.float32


def test_periodic_displacement_and_wrap(golden):
    g = golden["displacement_periodic"]
    r, d = O.periodic_displacement(g["p1"], g["p2"], g["box"])
    assert np.array_equal(r, np.array(g["r_ij"], dtype=f32))
    assert np.array_equal(d, np.array(g["dist"], dtype=f32))
    w = golden["wrap_periodic"]
    for case in w["cases"]:
        assert np.array_equal(O.periodic_wrap(case["x"], w["box"]), np.array(case["wrapped"], dtype=f32))
    with pytest.raises(ValueError):
        O.periodic_displacement(g["p1"], g["p2"], None)


def test_nonperiodic_displacement(golden):
    g = golden["displacement_nonperiodic"]
    r, d = O.nonperiodic_displacement(g["p1"], g["p2"])
    assert np.array_equal(r, np.array(g["r_ij"], dtype=f32))
    assert np.array_equal(d, np.array(g["dist"], dtype=f32))


def test_neighborlist_pair(golden):
    g = golden["neighborlist_pair"]
    x = np.array(g["coordinates"], dtype=f32)
    nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, g["box"], g["cutoff"], g["skin"], g["n_max_neighbors"])
    assert n_max == g["n_max_neighbors"]
    assert nl.dtype == np.uint32
    assert np.array_equal(nl, np.array(g["neighbor_list"]))
    assert np.array_equal(n_nbr, np.array(g["n_neighbors"]))
    assert np.array_equal(nl_mask, np.array(g["neighbor_mask"]))
    n_pairs, mask, d, r = O.neighborlist_calculate(x, nl, nl_mask, g["box"], g["cutoff"])
    assert np.array_equal(n_pairs, np.array(g["calc_n_neighbors"]))
    assert np.array_equal(mask, np.array(g["padding_mask"]))
    assert np.array_equal(d, np.array(g["dist"], dtype=f32))
    assert np.array_equal(r[0], np.tile(np.array(g["r_ij_row0"], dtype=f32), (5, 1)))
    assert np.array_equal(r[1], np.tile(np.array(g["r_ij_row1"], dtype=f32), (5, 1)))
    assert O.neighborlist_check(x, x, g["box"], g["skin"]) is g["check_same"]
    assert O.neighborlist_check(x + f32(g["check_shift"]), x, g["box"], g["skin"]) is g["check_shifted"]
    assert O.neighborlist_check(np.array(g["check_changed_n"], dtype=f32), x, g["box"], g["skin"]) is True


def test_neighborlist_cube(golden):
    g = golden["neighborlist_cube"]
    x = cube_coordinates(g["n_xyz"], g["scale_factor"])
    c = g["case_all"]
    nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, g["box"], c["cutoff"], c["skin"], c["n_max_neighbors"])
    assert np.array_equal(n_nbr, np.array(c["n_neighbors"]))
    n_int, _, _, _ = O.neighborlist_calculate(x, nl, nl_mask, g["box"], c["cutoff"])
    assert np.array_equal(n_int, np.array(c["n_interacting"]))
    c = g["case_subset"]
    nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, g["box"], c["cutoff"], c["skin"], c["n_max_neighbors"])
    assert n_max == c["grown_n_max"]  # growth 5 -> 7 + 10
    assert np.array_equal(n_nbr, np.array(c["n_neighbors"]))
    assert np.array_equal(nl, np.array(c["neighbor_list"]))
    n_int, _, _, _ = O.neighborlist_calculate(x, nl, nl_mask, g["box"], c["cutoff"])
    assert np.array_equal(n_int, np.array(c["n_interacting"]))


def test_pairlist_pair(golden):
    g = golden["pairlist_pair"]
    x = np.array(g["coordinates"], dtype=f32)
    all_pairs, red = O.pairlist_build(2)
    assert np.array_equal(all_pairs, np.array(g["all_pairs"]))
    assert np.array_equal(red, np.array(g["reduction_mask"]))
    for cutoff in (g["cutoff"], None):
        n_pairs, mask, d, r = O.pairlist_calculate(x, all_pairs, red, g["box"], cutoff)
        assert np.array_equal(n_pairs, np.array(g["n_pairs"]))
        assert np.array_equal(mask, np.array(g["mask"]))
        assert np.array_equal(d, np.array(g["dist"], dtype=f32))
        assert np.array_equal(r, np.array(g["displacement"], dtype=f32))
    _, mask, _, _ = O.pairlist_calculate(x, all_pairs, red, g["box"], g["short_cutoff"])
    assert np.array_equal(mask, np.array(g["short_cutoff_mask"]))
    _, mask, _, _ = O.pairlist_calculate(x, all_pairs, red, g["box"], None)
    assert np.array_equal(mask, np.array(g["no_cutoff_mask"]))


def test_pairlist_cube(golden):
    g = golden["pairlist_cube"]
    box = golden["neighborlist_cube"]["box"]
    x = cube_coordinates()
    all_pairs, red = O.pairlist_build(8)
    assert np.array_equal(all_pairs, np.array(g["all_pairs"]))
    n_int, mask, d, r = O.pairlist_calculate(x, all_pairs, red, box, g["cutoff"])
    assert np.array_equal(n_int, np.array(g["n_interacting"]))
    assert np.array_equal(d, np.array(g["dist"], dtype=f32))
    # mask-weighted distance sum equals the neighbor list's (test_pairs.py:471)
    nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, box, g["cutoff"], g["skin"], 20)
    _, mask1, d1, _ = O.neighborlist_calculate(x, nl, nl_mask, box, g["cutoff"])
    assert np.where(mask, d, 0).sum(dtype=f32) == np.where(mask1, d1, 0).sum(dtype=f32)


def test_nopbc_pairlist(golden):
    g = golden["nopbc_pairlist"]
    for case in g["two_particles_cases"]:
        d, r, pairs = O.nopbc_pairlist(np.array(g["two_particles_nm"]), case["cutoff"])
        assert pairs[0].size == case["n_pairs"] and pairs[1].size == case["n_pairs"]
    d, r, pairs = O.nopbc_pairlist(np.array(g["ethanol_nm"]), g["ethanol_cutoff"])
    assert pairs[0].size == g["ethanol_n_pairs"]
    assert np.all(pairs[0] < pairs[1])


def test_lennard_jones_two_particles(golden):
    g = golden["lennard_jones_two_particles"]
    sigma, eps, cutoff, skin = g["sigma"], g["epsilon"], g["cutoff"], g["skin"]
    for i in range(1, 11):
        x = np.array([[0, 0, 0], [i * 0.25 * 2 ** (1 / 6), 0, 0]], dtype=f32)
        nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, g["box"], cutoff, skin, g["n_max_neighbors"])
        e_pair = O.lj_energy_nopbc_f32(x, sigma, eps, cutoff)
        e_nbr = O.lj_energy_nbrlist_f32(x, nl, nl_mask, g["box"], sigma, eps, cutoff)
        disp = x[0] - x[1]
        dist = np.linalg.norm(disp)
        e_ref = 4.0 * eps * ((sigma / dist) ** 12 - (sigma / dist) ** 6)
        assert np.isclose(e_pair, e_ref, rtol=g["energy_rtol"], atol=g["energy_atol"])
        assert np.isclose(e_nbr, e_ref, rtol=g["energy_rtol"], atol=g["energy_atol"])
        if i == 4:
            assert np.isclose(e_nbr, g["energy_at_i4"], rtol=1e-5)
        f = (24 * (eps / (dist * dist)) * (2 * (sigma / dist) ** 12 - (sigma / dist) ** 6)) * disp
        f_ref = np.array([f, -f])
        assert np.allclose(O.lj_force_nopbc_f32(x, sigma, eps, cutoff), f_ref, atol=g["force_atol"])
        assert np.allclose(
            O.lj_force_nbrlist_f32(x, nl, nl_mask, g["box"], sigma, eps, cutoff), f_ref, atol=g["force_atol"]
        )
        e64, f64, _, _ = O.lj_bruteforce_f64(x, g["box"], sigma, eps, cutoff)
        assert np.isclose(e64, e_ref, rtol=1e-5) and np.allclose(f64, f_ref, atol=g["force_atol"], rtol=1e-5)


def test_lj_default_two_particles(golden):
    g = golden["lj_two_particle_default"]
    # two particles closer than the 1.0 nm default cutoff (test_testsystems.py:138-156)
    x = np.array([[0.0, 0.0, 0.0], [0.31, 0.12, -0.05]], dtype=f32)
    d = np.linalg.norm(x[0].astype(np.float64) - x[1])
    e_ref = 4 * g["epsilon_kj_mol"] * ((g["sigma_nm"] / d) ** 12 - (g["sigma_nm"] / d) ** 6)
    assert np.isclose(O.lj_energy_nopbc_f32(x, g["sigma_nm"], g["epsilon_kj_mol"], g["cutoff_nm"]), e_ref, rtol=1e-5)


def test_minimized_pair(golden):
    g = golden["minimized_pair"]
    x = np.array([[0, 0, 0], [g["separation"], 0, 0]], dtype=f32)
    assert np.isclose(O.lj_energy_nopbc_f32(x, 1.0, 1.0, 3.0), g["energy"], rtol=1e-6)
    assert np.allclose(O.lj_force_nopbc_f32(x, 1.0, 1.0, 3.0), 0.0, atol=1e-5)


def test_methane_fixture_crosscheck(golden, methane_coords):
    g = golden["methane"]
    L = g["box_edge"]
    box = np.diag([L, L, L])
    eps = g["epsilon_kcal_mol"] * 4.184
    e64, f64, eabs, fmax = O.lj_bruteforce_f64(methane_coords, box, g["sigma"], eps, g["cutoff"])
    assert abs(e64 - g["build_crosscheck_energy_kj_mol"]) < 1e-3
    # fp32 neighbor-list restatement (reference op order) agrees with fp64 within the stated tolerance
    x = methane_coords.astype(f32)
    nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, box, g["cutoff"], 0.4, 200)
    e32 = O.lj_energy_nbrlist_f32(x, nl, nl_mask, box, g["sigma"], eps, g["cutoff"])
    assert abs(float(e32) - e64) <= 1e-5 * eabs
    f32_ = O.lj_force_nbrlist_f32(x, nl, nl_mask, box, g["sigma"], eps, g["cutoff"])
    assert np.abs(f32_ - f64).max() <= 1e-5 * max(1.0, fmax)
    n_pairs, _, _, _ = O.neighborlist_calculate(x, nl, nl_mask, box, g["cutoff"])
    assert abs(int(n_pairs.sum()) - g["build_crosscheck_pairs_within_cutoff"]) <= 2


def test_ipow_matches_repeated_multiplication():
    x = np.linspace(0.3, 1.7, 101, dtype=f32)
    assert np.array_equal(O.ipow(x, 6), (x * x) * ((x * x) * (x * x)))
    x2 = x * x; x4 = x2 * x2; x8 = x4 * x4
    assert np.array_equal(O.ipow(x, 12), x4 * x8)


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    out = O.philox4x32(0, 0, 0, 0, 0, 0)
    assert [int(v) for v in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    out = O.philox4x32(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(v) for v in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    out = O.philox4x32(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
    assert [int(v) for v in out] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_philox_normals_statistics():
    z = O.philox_normals3(200000, 7, O.STREAM_LANGEVIN, (1234, 0))
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 1e-2


def test_metropolis_and_reduced_potential():
    assert O.metropolis_accept(0.5, 0.999) is True  # downhill always accepted
    assert O.metropolis_accept(-1.0, 0.3) is True and O.metropolis_accept(-1.0, 0.4) is False
    # ideal gas: u = beta p V (test_mcmc.py:440-448)
    u = O.reduced_potential(0.0, 300.0, 1.0, 1000.0)
    assert np.isclose(u, 101325.0 * 1000e-27 / (O.KB * 300.0))
    x, box, v0, v1 = O.barostat_propose(np.ones((4, 3)), np.diag([10.0, 10.0, 10.0]), 0.1, 0.75)
    assert np.isclose(v1 / v0, 1.05) and np.allclose(box[0, 0] ** 3, v1, rtol=1e-5)


# ---------------------------------------------------------------------------
# the C restatement (cpu_baseline) agrees with the NumPy oracle
# ---------------------------------------------------------------------------
def test_c_oracle_matches_numpy_oracle():
    from oracle import c_oracle as CO

    x, box = O.lj_lattice(700, 0.5, 0.34, seed=3)
    mask, nl, nnbr = CO.nl_build_fixed(x, box, 1.52, 150)
    mask0, nl0, nnbr0 = O.neighborlist_build_fixed(x, box, 1.52, 150)
    assert np.array_equal(nl, nl0) and np.array_equal(mask, mask0) and np.array_equal(nnbr, nnbr0)
    e, f = CO.lj_nl(x, nl, mask, box, 0.34, 0.995792, 1.02)
    e0 = O.lj_energy_nbrlist_f32(x, nl0, mask0, box, 0.34, 0.995792, 1.02)
    f0 = O.lj_force_nbrlist_f32(x, nl0, mask0, box, 0.34, 0.995792, 1.02)
    assert abs(e - float(e0)) <= 1e-5 * abs(float(e0)) and np.abs(f - f0).max() <= 1e-4 * max(1.0, np.abs(f0).max())
    # trajectory with injected noise, including rebuilds
    rng = np.random.default_rng(0)
    n, steps, skin = 300, 25, 0.02
    x, box = O.lj_lattice(n, 0.3, 0.34, seed=5)
    noise = rng.normal(size=(steps, n, 3)).astype(f32)
    mass = np.full(n, 39.948, f32)
    kT, a, b = O.langevin_constants(300.0, 0.004, 1.0)
    v0 = O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))
    xc, vc, st = CO.langevin_run(x, v0, box, mass, 0.34, 0.995792, 1.02, skin, 250, 0.004, a, b, kT, steps, noise)
    xo, vo, rebuilds, (_, _, n_max) = O.langevin_run(x, v0, box, mass, 0.34, 0.995792, 1.02, skin, 250, 0.004, 1.0,
                                                    300.0, steps, lambda s: noise[s])
    assert st["rebuilds"] == rebuilds and rebuilds >= 1 and st["n_max"] == n_max
    assert np.abs(xc - xo).max() < 1e-5 and np.abs(vc - vo).max() < 1e-3
    # internal Philox stream == NumPy stream
    kT, a, b = O.langevin_constants(300.0, 0.001, 1.0)
    xc, vc, _ = CO.langevin_run(x, v0, box, mass, 0.34, 0.995792, 1.02, 0.5, 250, 0.001, a, b, kT, 3, None, (7, 9))
    xo, vo, _, _ = O.langevin_run(x, v0, box, mass, 0.34, 0.995792, 1.02, 0.5, 250, 0.001, 1.0, 300.0, 3,
                                  lambda s: O.philox_normals3(n, s, O.STREAM_LANGEVIN, (7, 9)))
    assert np.abs(xc - xo).max() < 2e-6


def jax_stream_langevin_ho(g, steps=5):
    """The reference's HO Langevin test (chiron/tests/test_mcmc.py:12-84) on the oracle's BAOAB
    functions, with the draws of chiron/utils.py:27-38,142, chiron/states.py:150-154 and
    chiron/integrators.py:124,179,185 taken from the restated JAX stream.
    Returns (energies, x0, v0, noise[steps,1,3], k_md)."""
    f32 = np.float32
    _, state_key = JX.split(JX.PRNGKey(g["seed"]))  # PRNG.get_random_key()
    _, key = JX.split(state_key)                     # sampler_state.new_PRNG_key
    kT, a, b = O.langevin_constants(g["temperature_K"], g["timestep_fs"] * 1e-3, g["collision_rate_per_ps"])
    mass = np.array([g["mass_amu"]], f32)
    k_md = f32(g["k_kcal_mol_A2"] * 4.184 / 0.01)    # kJ/mol/nm^2
    x = np.zeros((1, 3), f32)
    v0 = O.initial_velocities(mass, kT, JX.normal(key, (1, 3)))  # refresh_velocities=True: the run key itself
    v = v0.copy()
    force = lambda xx: (-k_md * xx).astype(f32)      # -grad of (k/2) sum (x - x0)^2, chiron/potential.py:413-418
    F = force(x)
    energies, noise = [], []
    dt = g["timestep_fs"] * 1e-3
    for _ in range(steps):
        key, sub = JX.split(key)
        z = JX.normal(sub, (1, 3))
        noise.append(z)
        x, v = O.baoa_half(x, v, F, mass, dt, a, b, kT, z)
        F = force(x)
        v = O.kick(v, F, mass, dt)
        energies.append(float(f32(0.5) * k_md * np.sum((x * x).astype(f32), dtype=f32)))
    return np.array(energies), np.zeros((1, 3), f32), v0, np.stack(noise), float(k_md)


def test_langevin_literal_that_depends_on_the_jax_stream(golden):
    """chiron/tests/test_mcmc.py:81-84: the five Harmonic-oscillator energies, to the reference's own
    tolerance -- pins the oracle's BAOAB / velocity-initialisation arithmetic and the key handling
    against a real output of the reference."""
    g = golden["langevin_ho_jax_prng"]
    e, *_ = jax_stream_langevin_ho(g)
    assert np.allclose(e, np.array(g["potential_energy"]), rtol=g["rtol"], atol=1e-8)


def jax_stream_barostat_decisions(g, split=JX.split, uniform=JX.uniform):
    """The reference's ideal-gas barostat test (chiron/tests/test_mcmc.py:340-452) on the oracle's
    propose / log-ratio / accept functions with the restated JAX draws (chiron/mcmc.py:953,967,437)."""
    _, state_key = split(JX.PRNGKey(g["seed"]))
    L = g["box_nm"]
    x = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1]], np.float32)
    box = np.diag([L, L, L]).astype(np.float32)
    vol = lambda b: float(b[0, 0]) * float(b[1, 1]) * float(b[2, 2])
    u_cur = O.reduced_potential(0.0, g["temperature_K"], g["pressure_atm"], vol(box))
    decisions = []
    for _ in range(g["n_proposed"]):
        state_key, key = split(state_key)
        u01 = (float(uniform(key, (), -1.0, 1.0)) + 1.0) / 2.0  # barostat_propose maps u01 -> uniform(-1, 1)
        x_new, box_new, v0, v1 = O.barostat_propose(x, box, g["volume_max_scale"], u01)
        u_new = O.reduced_potential(0.0, g["temperature_K"], g["pressure_atm"], v1)
        lpr = O.barostat_log_ratio(u_new, u_cur, g["n_particles"], v1, v0)
        state_key, akey = split(state_key)
        ok = O.metropolis_accept(lpr, float(uniform(akey)))
        decisions.append(ok)
        if ok:
            x, box, u_cur = x_new, box_new, u_new
    return decisions, vol(box)


def test_barostat_literal_that_depends_on_the_jax_stream(golden):
    """chiron/tests/test_mcmc.py:451-452: 8 of the 10 volume moves are accepted."""
    g = golden["barostat_ideal_gas_jax_prng"]
    decisions, _ = jax_stream_barostat_decisions(g)
    assert len(decisions) == g["n_proposed"] and sum(decisions) == g["n_accepted"]


def test_threefry_known_answers():
    """Random123 known-answer vectors for Threefry-2x32-20 (kat_vectors: zero, all-ones, pi digits)."""
    u = np.uint32
    x0, x1 = JX.threefry2x32((0, 0), np.array([0], u), np.array([0], u))
    assert (int(x0[0]), int(x1[0])) == (0x6B200159, 0x99BA4EFE)
    x0, x1 = JX.threefry2x32((0xFFFFFFFF, 0xFFFFFFFF), np.array([0xFFFFFFFF], u), np.array([0xFFFFFFFF], u))
    assert (int(x0[0]), int(x1[0])) == (0x1CB996FC, 0xBB002BE7)
    x0, x1 = JX.threefry2x32((0x13198A2E, 0x03707344), np.array([0x243F6A88], u), np.array([0x85A308D3], u))
    assert (int(x0[0]), int(x1[0])) == (0xC4923A9C, 0x483DF7A0)


# ---------------------------------------------------------------------------
# Monte Carlo chains of the oracle (chiron/mcmc.py:243-306, :357-463, :680-787, :902-1009)
# ---------------------------------------------------------------------------
def test_oracle_mc_chain_bookkeeping():
    """Key-chain and counter rules of the restated trial loop: two key splits per trial (one when the proposed
    reduced potential is NaN: :428-429 skips new_PRNG_key), U = 0 accepts everything (-log ratio <= 0, :544),
    cached energy reused after a rejection, autotune thresholds (:670-678)."""
    from oracle import chiron_oracle as O

    x, box = O.lj_lattice(64, 0.1, 0.34, seed=3)
    key = (11, 22)
    zero = lambda *_: 0.0  # noqa: E731
    xs, _, k1, sig, c, log = O.mc_displacement_update(x, box, None, key, 0.34, 1.0, 300.0, 0.05, 7, energy_fn=zero,
                                                      autotune=True, autotune_interval=5)
    k = key
    for _ in range(14):
        k = O.key_split(k)[0]
    assert k1 == k and c == (7, 7, 7) and all(l["accepted"] for l in log)
    assert np.isclose(sig, 0.05 * 1.1) and log[4]["sigma_d"] == 0.05 and log[5]["sigma_d"] == sig
    nan = lambda *_: float("nan")  # noqa: E731
    xs2, _, k2, _, c2, log2 = O.mc_displacement_update(x, box, None, key, 0.34, 1.0, 300.0, 0.05, 3, energy_fn=nan)
    k = key
    for _ in range(3):
        k = O.key_split(k)[0]
    assert k2 == k and c2 == (0, 3, 3) and np.array_equal(xs2, x) and log2[0]["uniform"] is None
    # a real LJ chain: rejected trials keep the configuration and the cached energy
    lst = O.McList(x, box, 1.02, 0.3, 40)
    xs3, lst3, _, _, c3, log3 = O.mc_displacement_update(x, box, lst, key, 0.34, 0.9958, 300.0, 0.15, 12)
    assert 0 < c3[0] < 12
    e = [lst.energy(x, 0.34, 0.9958)] + [l["energy"] for l in log3]
    for i, l in enumerate(log3):
        assert (e[i + 1] == l["energy_proposed"]) if l["accepted"] else (e[i + 1] == e[i])
        assert l["accepted"] == O.metropolis_accept(l["log_ratio"], l["uniform"])
    assert lst3.builds - 1 == sum(l["rebuilt_list"] and l["accepted"] for l in log3) or lst3.builds >= 1


def test_oracle_c1_ideal_gas_volume_statistics():
    """The oracle's own run of config C1 (Examples/Idealgas.py:16-104) meets the example's acceptance test
    (:141-150): <V> within 5 % and sigma_V within 10 % of the ideal-gas values after 10 and after 100 iterations."""
    from oracle import chiron_oracle as O

    n, T, P = 216, 298.0, 1.0
    kT_over_p = O.KB * T / 101325.0 * 1e27  # nm^3
    L = (n * kT_over_p) ** (1.0 / 3.0)
    x = np.random.default_rng(1234).uniform(0.0, L, size=(n, 3)).astype(np.float32)
    box = np.diag([L, L, L]).astype(np.float32)
    zero = lambda *_: 0.0  # noqa: E731
    key, sig, vs, cd, cb, vols = (1234, 0), 0.1, 0.2, (0, 0, 0), (0, 0, 0), []
    for it in range(100):
        x, _, key, sig, cd, _ = O.mc_displacement_update(x, box, None, key, 0.34, 1.0, T, sig, 100, pressure_atm=P,
                                                         autotune=True, autotune_interval=100, counters=cd, energy_fn=zero)
        x, box, key, vs, cb, lg = O.mc_barostat_update(x, box, key, T, P, vs, 10, zero, autotune=True, autotune_interval=100,
                                                       counters=cb)
        vols += [l["volume"] for l in lg]
        if it + 1 in (10, 100):
            v = np.array(vols)
            assert abs(v.mean() - (n + 1) * kT_over_p) / ((n + 1) * kT_over_p) < 0.05
            assert abs(v.std() - math.sqrt(n + 1) * kT_over_p) / (math.sqrt(n + 1) * kT_over_p) < 0.1
    assert cb[1] == 1000 and 0.25 <= cb[0] / cb[1] <= 0.75


def test_c_oracle_cell_omp_variant_is_bit_identical_to_the_faithful_port():
    """cpu_baseline variant (ii) (cell-list build + OpenMP gather forces, oracle/oracle.c) reproduces the faithful
    1-thread port bit for bit -- trajectory, rebuild steps -- for any thread count, periodic with and without a
    usable cell grid, and the faithful port itself follows the NumPy oracle (injected noise)."""
    from oracle import c_oracle as CO
    from oracle import chiron_oracle as O

    eps = 0.238 * 4.184
    for n, rho, skin, steps in ((1500, 0.3, 0.1, 40), (200, 0.5, 0.08, 25)):  # second box: < 3 cells per axis
        x, box = O.lj_lattice(n, rho, 0.34, seed=5)
        mass = np.full(n, 39.948, np.float32)
        kT, a, b = O.langevin_constants(300.0, 0.002, 1.0)
        v0 = (3.0 * O.initial_velocities(mass, kT, O.philox_normals3(n, 0, 0, (1, 2)))).astype(np.float32)
        x1, v1, s1 = CO.langevin_run(x, v0, box, mass, 0.34, eps, 1.02, skin, 180, 0.002, a, b, kT, steps, None, (9, 9))
        assert s1["rebuilds"] >= 2
        for threads in (1, 3):
            x2, v2, s2 = CO.langevin_run_cell_omp(x, v0, box, mass, 0.34, eps, 1.02, skin, 0.002, a, b, kT, steps, None, (9, 9),
                                                  threads=threads, cap0=8)  # cap0 = 8 forces the capacity growth loop
            assert s2["rebuilds"] == s1["rebuilds"] and s2["threads"] == threads
            assert np.array_equal(x1, x2) and np.array_equal(v1, v2)
    assert CO.effective_cores() >= 1


def test_c_oracle_full_size_checkers_against_the_numpy_oracle():
    """The two routines the full-size GPU tests lean on (every list row, every particle's fp64 force at N = 32 768 /
    262 144) pinned at sizes the NumPy oracle reaches: full rows through the C cell grid give the O(N^2) half list
    bit for bit; the fp64 cell-list energy / forces equal the fp64 brute force over all pairs."""
    from oracle import c_oracle as CO

    sigma, eps, cutoff, skin = 0.34, 0.238 * 4.184, 1.02, 0.5
    for n, rho, seed in ((1500, 0.5, 11), (4096, 0.1, 12), (900, 0.8, 13)):
        x, box = O.lj_lattice(n, rho, sigma, seed=seed)
        rng = np.random.default_rng(seed)
        x = O.periodic_wrap((x + rng.normal(0, 0.03, x.shape)).astype(np.float32), box)
        rows, deg = CO.full_rows_cell(x, box, cutoff + skin, threads=2, cap0=8)  # (cap0 small: the growth path runs)
        nn, flat = CO.half_list_from_full_rows(rows, deg)
        m = int(nn.max()) + 1
        mask, nl, nnbr = CO.nl_build_fixed(x, box, cutoff + skin, m)
        real = np.arange(m)[None, :] < nnbr[:, None]
        assert np.array_equal(nn, nnbr) and np.array_equal(flat, nl[real])
        # ... and the NumPy oracle's row for a few particles
        for i in (0, n // 2, n - 2):
            _, d = O.periodic_displacement(np.broadcast_to(x[i], (n, 3)), x, box)
            want = np.nonzero((d < np.float32(cutoff + skin)) & (np.arange(n) > i))[0]
            assert np.array_equal(rows[i, :deg[i]][rows[i, :deg[i]] > i], want)
        e, f, eabs, fabs, pairs = CO.lj_cell_f64(x, box, sigma, eps, cutoff, threads=2, cap0=4)
        # (the C routine takes fp32 parameters, like the kernels: hand the brute force the same rounded values)
        e64, f64, eabs64, fmax64 = O.lj_bruteforce_f64(x, box, float(f32(sigma)), float(f32(eps)), float(f32(cutoff)))
        assert abs(e - e64) <= 1e-12 * eabs64 and abs(eabs - eabs64) <= 1e-12 * eabs64
        assert np.abs(f - f64).max() <= 1e-11 * max(1.0, fmax64)
        assert pairs > n  # a fluid, not a gas of isolated particles
        # the variant the production-state GPU tests use: same routine with the per-particle noise scale ...
        e_x, f_x, eabs_x, fa_x, _, pairs_x = CO.lj_cell_f64_ex(x, box, sigma, eps, cutoff, threads=2, cap0=4, geom32=False)
        assert e_x == e and eabs_x == eabs and pairs_x == pairs and np.array_equal(f_x, f) and fa_x.max() == fabs
        # ... and with the REFERENCE's fp32 pair vectors (chiron/neighbors.py:69-79) under fp64 Lennard-Jones values:
        # restated here with the NumPy oracle's displacement over all pairs
        e_g, f_g, eabs_g, fa_g, ft_g, pairs_g = CO.lj_cell_f64_ex(x, box, sigma, eps, cutoff, threads=2, cap0=4, geom32=True)
        assert pairs_g == pairs
        ii, jj = np.triu_indices(n, k=1)
        r32, d32 = O.periodic_displacement(x[ii], x[jj], box)
        sel = d32 < np.float32(cutoff)
        assert int(sel.sum()) == pairs
        r = r32[sel].astype(np.float64)
        d2 = (r * r).sum(axis=1)
        s6 = (float(f32(sigma)) ** 2 / d2) ** 3
        fs = 24.0 * float(f32(eps)) * (2.0 * s6 * s6 - s6) / d2
        want_f = np.zeros((n, 3))
        want_a = np.zeros((n, 3))
        want_t = np.zeros((n, 3))
        ts = 24.0 * float(f32(eps)) * (2.0 * s6 * s6 + s6) / d2
        np.add.at(want_t, ii[sel], ts[:, None] * np.abs(r))
        np.add.at(want_t, jj[sel], ts[:, None] * np.abs(r))
        np.add.at(want_f, ii[sel], fs[:, None] * r)
        np.add.at(want_f, jj[sel], -fs[:, None] * r)
        np.add.at(want_a, ii[sel], np.abs(fs[:, None] * r))
        np.add.at(want_a, jj[sel], np.abs(fs[:, None] * r))
        assert np.abs(f_g - want_f).max() <= 1e-11 * max(1.0, want_a.max())
        assert np.abs(fa_g - want_a).max() <= 1e-11 * max(1.0, want_a.max())
        assert np.abs(ft_g - want_t).max() <= 1e-11 * max(1.0, want_t.max()) and (ft_g >= fa_g * (1 - 1e-12)).all()
        assert abs(e_g - (4.0 * float(f32(eps)) * (s6 * s6 - s6)).sum()) <= 1e-11 * eabs_g
        # the two geometries differ by the fp32 minimum image's own rounding, far below any tolerance at this scale
        assert np.abs(f_g - f).max() <= 1e-4 * max(1.0, fabs)


def test_c_oracle_binary_cannot_go_stale(tmp_path, monkeypatch):
    """oracle/_build/liboracle.so carries the hash of oracle.c + flags; build() recompiles on any difference and
    c_oracle.lib() always goes through build() (round-2 review: an existing binary used to load unchecked)."""
    import shutil

    from oracle import build_oracle

    assert build_oracle.binary_hash(build_oracle.build()) == build_oracle.source_hash()
    src = tmp_path / "oracle.c"
    shutil.copy(build_oracle.SRC, src)
    out = tmp_path / "liboracle.so"
    monkeypatch.setattr(build_oracle, "SRC", str(src))
    monkeypatch.setattr(build_oracle, "OUT", str(out))
    monkeypatch.setattr(build_oracle, "OUT_DIR", str(tmp_path))
    build_oracle.build()
    first = build_oracle.binary_hash(str(out))
    assert first == build_oracle.source_hash()
    stamp = out.stat().st_mtime_ns
    build_oracle.build()
    assert out.stat().st_mtime_ns == stamp  # current: not rebuilt
    src.write_text(src.read_text() + "\n/* edited */\n")
    os.utime(src, ns=(1, 1))  # an OLD mtime: file times are not what decides
    assert build_oracle.binary_hash(str(out)) != build_oracle.source_hash()
    build_oracle.build()
    assert build_oracle.binary_hash(str(out)) == build_oracle.source_hash() != first
