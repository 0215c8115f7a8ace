"""GPU parity tests: HIP kernels (through the C ABI / the reference-shaped classes)
against the CPU oracle, the reference's golden literals, and fp64 brute force.

Bars: neighbor indices / masks / counts / dist / r_ij bit-exact;
energies |E32 - E64| <= 1e-5 * sum|e_ij|; forces max|F32 - F64| <= 1e-5 * max(1, max_i sum_j |f_ij|).
"""
import copy
import math

import numpy as np
import pytest
import torch

from conftest import cube_coordinates

pytestmark = pytest.mark.gpu

f32 = np.float32
SIGMA, EPS_KCAL = 0.34, 0.238
EPS = EPS_KCAL * 4.184
CUTOFF, SKIN = 3 * SIGMA, 0.5


@pytest.fixture(scope="module")
def O():
    from oracle import chiron_oracle

    return chiron_oracle


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import chiron_amd
    from chiron_amd import _lib, integrators, mcmc, neighbors, ops, potential, states, testsystems, unit, utils

    _lib.load()
    return type("NS", (), dict(unit=unit, neighbors=neighbors, potential=potential, states=states, ops=ops,
                               utils=utils, integrators=integrators, mcmc=mcmc, testsystems=testsystems))


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def nm(x, ca):
    return ca.unit.Quantity(x, ca.unit.nanometer)


def make_state(ca, x, box):
    ca.utils.PRNG.set_seed(1234)
    return ca.states.SamplerState(nm(np.asarray(x, dtype=f32), ca), ca.utils.PRNG.get_random_key(),
                                  box_vectors=None if box is None else nm(np.asarray(box, dtype=f32), ca))


def nbr_class(ca, name):
    """'NeighborListNsqrd' here means the O(N^2) kernel explicitly (the class's default 'auto'
    algorithm would take the cell kernel on most boxes); 'NeighborListAuto' is the plain default."""
    import functools

    if name == "NeighborListNsqrd":
        return functools.partial(ca.neighbors.NeighborListNsqrd, algorithm="nsq")
    if name == "NeighborListAuto":
        return ca.neighbors.NeighborListNsqrd
    return getattr(ca.neighbors, name)


def lj_pot(ca, n, sigma=SIGMA, eps_kj=EPS, cutoff=CUTOFF):
    u = ca.unit
    from chiron_amd.topology import Topology

    return ca.potential.LJPotential(Topology.homogeneous(n), sigma=sigma * u.nanometer,
                                    epsilon=eps_kj * u.kilojoule_per_mole, cutoff=cutoff * u.nanometer)


# ---------------------------------------------------------------------------
# spaces
# ---------------------------------------------------------------------------
def test_spaces_golden(ca, golden):
    g = golden["displacement_periodic"]
    sp = ca.neighbors.OrthogonalPeriodicSpace()
    r, d = sp.displacement(np.array(g["p1"], f32), np.array(g["p2"], f32), np.array(g["box"], f32))
    assert np.array_equal(r.cpu().numpy(), np.array(g["r_ij"], f32))
    assert np.array_equal(d.cpu().numpy(), np.array(g["dist"], f32))
    w = golden["wrap_periodic"]
    for case in w["cases"]:
        out = sp.wrap(np.array([case["x"]], f32), np.array(w["box"], f32))
        assert np.array_equal(out.cpu().numpy()[0], np.array(case["wrapped"], f32))
    with pytest.raises(ValueError):
        sp.displacement(np.array(g["p1"], f32), np.array(g["p2"], f32), None)
    g = golden["displacement_nonperiodic"]
    sp = ca.neighbors.OrthogonalNonPeriodicSpace()
    r, d = sp.displacement(np.array(g["p1"], f32), np.array(g["p2"], f32), None)
    assert np.array_equal(r.cpu().numpy(), np.array(g["r_ij"], f32))
    assert np.array_equal(d.cpu().numpy(), np.array(g["dist"], f32))
    assert np.array_equal(sp.wrap(np.array([g["wrap_x"]], f32)).cpu().numpy()[0], np.array(g["wrap_expected"], f32))


def test_spaces_random_bitexact(ca, O):
    rng = np.random.default_rng(7)
    box = np.diag([3.7, 4.1, 5.3]).astype(f32)
    a = (rng.uniform(-12, 12, size=(5000, 3))).astype(f32)  # deliberately unwrapped, several box lengths away
    b = (rng.uniform(-12, 12, size=(5000, 3))).astype(f32)
    sp = ca.neighbors.OrthogonalPeriodicSpace()
    r, d = sp.displacement(a, b, box)
    r0, d0 = O.periodic_displacement(a, b, box)
    assert np.array_equal(r.cpu().numpy(), r0) and np.array_equal(d.cpu().numpy(), d0)
    assert np.array_equal(sp.wrap(a, box).cpu().numpy(), O.periodic_wrap(a, box))


# ---------------------------------------------------------------------------
# neighbor list: golden literals
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("cls_name", ["NeighborListNsqrd", "NeighborListCell"])
def test_neighborlist_pair_golden(ca, golden, cls_name):
    g = golden["neighborlist_pair"]
    u = ca.unit
    cls = nbr_class(ca, cls_name)
    x = np.array(g["coordinates"], f32)
    state = make_state(ca, x, g["box"])
    nbr = cls(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=g["cutoff"] * u.nanometer, skin=g["skin"] * u.nanometer,
              n_max_neighbors=g["n_max_neighbors"])
    assert nbr.cutoff == g["cutoff"] * u.nanometer and nbr.skin == g["skin"] * u.nanometer
    nbr.build_from_state(state)
    assert nbr.is_built and nbr.n_max_neighbors == 5
    assert np.array_equal(nbr.ref_positions.cpu().numpy(), x)
    assert np.array_equal(u32(nbr.neighbor_list), np.array(g["neighbor_list"]))
    assert np.array_equal(nbr.n_neighbors.cpu().numpy(), np.array(g["n_neighbors"]))
    assert np.array_equal(nbr.neighbor_mask.cpu().numpy(), np.array(g["neighbor_mask"]))
    n_nb, nl, mask, dist, rij = nbr.calculate(x)
    assert np.array_equal(n_nb.cpu().numpy(), np.array(g["calc_n_neighbors"]))
    assert nl.shape == (2, 5)
    assert np.array_equal(mask.cpu().numpy(), np.array(g["padding_mask"]))
    assert np.array_equal(dist.cpu().numpy(), np.array(g["dist"], f32))
    assert np.array_equal(rij.cpu().numpy()[0], np.tile(np.array(g["r_ij_row0"], f32), (5, 1)))
    assert np.array_equal(rij.cpu().numpy()[1], np.tile(np.array(g["r_ij_row1"], f32), (5, 1)))
    assert nbr.check(x) is False
    assert nbr.check(x + f32(0.1)) is True
    assert nbr.check(np.array(g["check_changed_n"], f32)) is True


@pytest.mark.parametrize("cls_name", ["NeighborListNsqrd", "NeighborListCell"])
def test_neighborlist_cube_golden(ca, golden, cls_name):
    g = golden["neighborlist_cube"]
    u = ca.unit
    cls = nbr_class(ca, cls_name)
    x = cube_coordinates(g["n_xyz"], g["scale_factor"])
    state = make_state(ca, x, g["box"])
    c = g["case_all"]
    nbr = cls(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=c["cutoff"] * u.nanometer, skin=c["skin"] * u.nanometer,
              n_max_neighbors=c["n_max_neighbors"])
    nbr.build_from_state(state)
    assert np.array_equal(nbr.n_neighbors.cpu().numpy(), np.array(c["n_neighbors"]))
    assert np.array_equal(nbr.calculate(x)[0].cpu().numpy(), np.array(c["n_interacting"]))
    c = g["case_subset"]
    nbr = cls(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=c["cutoff"] * u.nanometer, skin=c["skin"] * u.nanometer,
              n_max_neighbors=c["n_max_neighbors"])
    nbr.build_from_state(state)
    assert nbr.n_max_neighbors == c["grown_n_max"]
    n_int, nl, mask, dist, rij = nbr.calculate(x)
    assert np.array_equal(n_int.cpu().numpy(), np.array(c["n_interacting"]))
    assert nl.shape == (8, 17)
    assert np.array_equal(u32(nl), np.array(c["neighbor_list"]))
    nbr.build(state.positions, state.box_vectors)
    assert np.array_equal(nbr.n_neighbors.cpu().numpy(), np.array(c["n_neighbors"]))


def test_neighborlist_input_errors(ca):
    u = ca.unit
    N = ca.neighbors
    space = N.OrthogonalPeriodicSpace()
    nbr = N.NeighborListNsqrd(space, cutoff=2.1 * u.nanometer, skin=0.1 * u.nanometer, n_max_neighbors=5)
    with pytest.raises(TypeError):
        nbr.build_from_state(123)
    x = np.array([[1, 2, 3], [0, 0, 0]], f32)
    with pytest.raises(ValueError):
        nbr.build_from_state(make_state(ca, x, None))
    with pytest.raises(ValueError):
        nbr.build(x, np.zeros((4, 3), f32))
    box = np.diag([10.0, 10.0, 10.0]).astype(f32)
    with pytest.raises(ValueError):
        nbr.build(u.Quantity(x, u.radians), box)
    with pytest.raises(ValueError):
        nbr.build(u.Quantity(x, u.nanometers), u.Quantity(box, u.radians))
    with pytest.raises(TypeError):
        N.NeighborListNsqrd(123, cutoff=2.1 * u.nanometer, skin=0.1 * u.nanometer)
    with pytest.raises(ValueError):
        N.NeighborListNsqrd(space, cutoff=u.Quantity(123, u.radian), skin=u.Quantity(123, u.nanometer))
    with pytest.raises(ValueError):
        N.NeighborListNsqrd(space, cutoff=u.Quantity(123, u.nanometer), skin=u.Quantity(123, u.radian))


# ---------------------------------------------------------------------------
# neighbor list: seeded random systems, bit-exact vs the oracle
# ---------------------------------------------------------------------------
def _systems(O):
    out = []
    for n, rho, seed in [(1500, 0.1, 1), (1200, 0.8, 2), (777, 0.02, 3)]:
        x, box = O.lj_lattice(n, rho, SIGMA, seed=seed)
        out.append((f"n{n}_rho{rho}", x, box))
    # unwrapped coordinates: whole box lengths added at random
    x, box = O.lj_lattice(1000, 0.3, SIGMA, seed=4)
    rng = np.random.default_rng(5)
    x = (x + rng.integers(-2, 3, size=x.shape) * box[0, 0]).astype(f32)
    out.append(("unwrapped", x, box))
    # non-cubic box
    rng = np.random.default_rng(6)
    box = np.diag([5.0, 6.5, 8.1]).astype(f32)
    x = (rng.uniform(0, 1, size=(1400, 3)) * np.diag(box)).astype(f32)
    out.append(("noncubic_random", x, box))
    return out


@pytest.mark.parametrize("cls_name", ["NeighborListNsqrd", "NeighborListCell"])
def test_neighborlist_random_bitexact(ca, O, cls_name):
    u = ca.unit
    cls = nbr_class(ca, cls_name)
    for name, x, box in _systems(O):
        for m0 in (180, 12):  # 12 forces the growth loop
            nbr = cls(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer, skin=SKIN * u.nanometer,
                      n_max_neighbors=m0)
            nbr.build(x, box)
            nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, box, CUTOFF, SKIN, m0)
            assert nbr.n_max_neighbors == n_max, name
            assert np.array_equal(nbr.n_neighbors.cpu().numpy(), n_nbr), name
            assert np.array_equal(u32(nbr.neighbor_list), nl), name
            assert np.array_equal(nbr.neighbor_mask.cpu().numpy(), nl_mask), name
        # calculate() at displaced positions: every output bit-exact
        rng = np.random.default_rng(11)
        x2 = (x + rng.normal(0, 0.05, size=x.shape)).astype(f32)
        n_p, _, mask, dist, rij = nbr.calculate(x2)
        n_p0, mask0, d0, r0 = O.neighborlist_calculate(x2, nl, nl_mask, box, CUTOFF)
        assert np.array_equal(n_p.cpu().numpy(), n_p0), name
        assert np.array_equal(mask.cpu().numpy(), mask0), name
        assert np.array_equal(dist.cpu().numpy(), d0), name
        assert np.array_equal(rij.cpu().numpy(), r0), name
        assert nbr.check(x2) == O.neighborlist_check(x2, x, box, SKIN), name
        assert nbr.check(x) is False


def test_neighborlist_nonperiodic_and_truncation(ca, O):
    u = ca.unit
    x, box = O.lj_lattice(900, 0.6, SIGMA, seed=9)
    nbr = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalNonPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=SKIN * u.nanometer, n_max_neighbors=400)
    nbr.build(x, None)
    nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, None, CUTOFF, SKIN, 400, periodic=False)
    assert np.array_equal(u32(nbr.neighbor_list), nl) and np.array_equal(nbr.n_neighbors.cpu().numpy(), n_nbr)
    # fixed-width pass with rows longer than m: first m ascending ids, mask all ones (reference quirk)
    nlk, maskk, n_nbrk = ca.ops.nl_build_nsq(dev(x), ca.ops.make_box(box, True), CUTOFF + SKIN, 7)
    mask0, nl0, n0 = O.neighborlist_build_fixed(x, box, CUTOFF + SKIN, 7)
    assert np.array_equal(u32(nlk), nl0) and np.array_equal(maskk.cpu().numpy(), mask0)
    assert np.array_equal(n_nbrk.cpu().numpy(), n0)


def test_cell_equals_nsq_medium(ca, O):
    """Cell build == O(N^2) build on the GPU itself at a size the NumPy oracle would crawl through."""
    u = ca.unit
    x, box = O.lj_lattice(32768, 0.1, SIGMA, seed=1234)
    a = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                       skin=SKIN * u.nanometer, n_max_neighbors=180, algorithm="nsq")
    b = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                      skin=SKIN * u.nanometer, n_max_neighbors=180)
    a.build(x, box)
    b.build(x, box)
    assert b._perm is not None, "cell path was not taken"
    assert a._perm is None, "the O(N^2) kernel was not taken"
    c = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                       skin=SKIN * u.nanometer, n_max_neighbors=180)  # default: auto -> cell kernel
    c.build(x, box)
    assert c._perm is not None and torch.equal(c.neighbor_list, a.neighbor_list)
    with pytest.raises(ValueError):
        ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), algorithm="octree")
    assert torch.equal(a.neighbor_list, b.neighbor_list)
    assert torch.equal(a.n_neighbors, b.n_neighbors) and torch.equal(a.neighbor_mask, b.neighbor_mask)
    assert 4.5e5 < int(a.n_neighbors.sum()) < 7.5e5  # ~6e5 list pairs at rho*=0.1


# ---------------------------------------------------------------------------
# pair list
# ---------------------------------------------------------------------------
def test_pairlist_golden(ca, golden):
    u = ca.unit
    g = golden["pairlist_pair"]
    x = np.array(g["coordinates"], f32)
    state = make_state(ca, x, g["box"])
    space = ca.neighbors.OrthogonalPeriodicSpace()
    pl = ca.neighbors.PairListNsqrd(space, cutoff=g["cutoff"] * u.nanometer)
    assert pl.cutoff == g["cutoff"] * u.nanometer
    pl.build_from_state(state)
    assert np.array_equal(u32(pl.all_pairs), np.array(g["all_pairs"]))
    assert np.array_equal(pl.reduction_mask.cpu().numpy(), np.array(g["reduction_mask"]))
    for cutoff in (g["cutoff"] * u.nanometer, None):
        pl.cutoff = cutoff
        n_pairs, all_pairs, mask, dist, disp = pl.calculate(state.positions)
        assert np.array_equal(n_pairs.cpu().numpy(), np.array(g["n_pairs"]))
        assert all_pairs.shape == (2, 1)
        assert np.array_equal(mask.cpu().numpy(), np.array(g["mask"]))
        assert np.array_equal(dist.cpu().numpy(), np.array(g["dist"], f32))
        assert disp.shape == (2, 1, 3)
        assert np.array_equal(disp.cpu().numpy(), np.array(g["displacement"], f32))
    assert pl.check(x) is False
    assert pl.check(np.zeros((3, 3), f32)) is True
    with pytest.raises(ValueError):
        pl.calculate(np.zeros((3, 3), f32))
    pl.cutoff = g["short_cutoff"] * u.nanometer
    assert np.array_equal(pl.calculate(state.positions)[2].cpu().numpy(), np.array(g["short_cutoff_mask"]))
    pl.cutoff = None
    assert np.array_equal(pl.calculate(state.positions)[2].cpu().numpy(), np.array(g["no_cutoff_mask"]))

    g = golden["pairlist_cube"]
    x = cube_coordinates()
    box = golden["neighborlist_cube"]["box"]
    pl = ca.neighbors.PairListNsqrd(space, cutoff=g["cutoff"] * u.nanometer)
    pl.build(x, np.array(box, f32))
    n_int, all_pairs, mask, dist, rij = pl.calculate(x)
    assert np.array_equal(n_int.cpu().numpy(), np.array(g["n_interacting"]))
    assert np.array_equal(u32(all_pairs), np.array(g["all_pairs"]))
    assert np.array_equal(dist.cpu().numpy(), np.array(g["dist"], f32))
    nbr = ca.neighbors.NeighborListNsqrd(space, cutoff=g["cutoff"] * u.nanometer, skin=g["skin"] * u.nanometer,
                                         n_max_neighbors=20)
    nbr.build(x, np.array(box, f32))
    _, _, mask1, dist1, _ = nbr.calculate(x)
    s0 = np.where(mask.cpu().numpy(), dist.cpu().numpy(), 0).sum(dtype=f32)
    s1 = np.where(mask1.cpu().numpy(), dist1.cpu().numpy(), 0).sum(dtype=f32)
    assert s0 == s1


def test_pairlist_random_bitexact(ca, O):
    u = ca.unit
    x, box = O.lj_lattice(300, 0.5, SIGMA, seed=21)
    pl = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer)
    pl.build(x, box)
    ap, red = O.pairlist_build(300)
    assert np.array_equal(u32(pl.all_pairs), ap) and np.array_equal(pl.reduction_mask.cpu().numpy(), red)
    for cutoff in (CUTOFF, None):
        pl.cutoff = None if cutoff is None else cutoff * u.nanometer
        n_p, _, mask, dist, rij = pl.calculate(x)
        n0, m0, d0, r0 = O.pairlist_calculate(x, ap, red, box, cutoff)
        assert np.array_equal(n_p.cpu().numpy(), n0) and np.array_equal(mask.cpu().numpy(), m0)
        assert np.array_equal(dist.cpu().numpy(), d0) and np.array_equal(rij.cpu().numpy(), r0)


def test_nopbc_pairlist_golden(ca, golden):
    g = golden["nopbc_pairlist"]
    pot = ca.potential.NeuralNetworkPotential(model=None)
    for case in g["two_particles_cases"]:
        d, r, pairs = pot.compute_pairlist(np.array(g["two_particles_nm"], f32), case["cutoff"])
        assert pairs[0].numel() == case["n_pairs"] and pairs[1].numel() == case["n_pairs"]
    d, r, pairs = pot.compute_pairlist(np.array(g["ethanol_nm"], f32), g["ethanol_cutoff"])
    assert pairs[0].numel() == g["ethanol_n_pairs"]


# ---------------------------------------------------------------------------
# LJ energy / force
# ---------------------------------------------------------------------------
def test_lj_input_errors(ca):
    u = ca.unit
    LJ = ca.potential.LJPotential
    with pytest.raises(TypeError):
        LJ(1)
    with pytest.raises(TypeError):
        LJ(None, sigma=1.0)
    with pytest.raises(TypeError):
        LJ(None, epsilon=1.0)
    with pytest.raises(TypeError):
        LJ(None, cutoff=1.0)
    with pytest.raises(ValueError):
        LJ(None, sigma=1.0 * u.kilocalories_per_mole)
    with pytest.raises(ValueError):
        LJ(None, epsilon=1.0 * u.nanometer)
    with pytest.raises(ValueError):
        LJ(None, cutoff=1.0 * u.kilocalories_per_mole)
    x = np.array([[0, 0, 0], [1, 0, 0]], f32)
    box = np.diag([10.0, 10.0, 10.0]).astype(f32)
    lj = LJ(None, sigma=1.0 * u.nanometer)
    nbr = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=2.0 * u.nanometer)
    with pytest.raises(ValueError):
        lj.compute_energy(x, nbr)  # not built
    nbr.build(x, box)
    with pytest.raises(ValueError):
        lj.compute_energy(x, nbr)  # cutoff mismatch


def test_lennard_jones_two_particles_golden(ca, golden):
    g = golden["lennard_jones_two_particles"]
    u = ca.unit
    sigma, eps, cutoff, skin = g["sigma"], g["epsilon"], g["cutoff"], g["skin"]
    pot = ca.potential.LJPotential(None, u.Quantity(sigma, u.nanometer), u.Quantity(eps, u.kilojoules_per_mole),
                                   u.Quantity(cutoff, u.nanometer))
    space = ca.neighbors.OrthogonalPeriodicSpace()
    for i in range(1, 11):
        x = np.array([[0, 0, 0], [i * 0.25 * 2 ** (1 / 6), 0, 0]], f32)
        state = make_state(ca, x, g["box"])
        nbr = ca.neighbors.NeighborListNsqrd(space, cutoff=u.Quantity(cutoff, u.nanometer),
                                             skin=u.Quantity(skin, u.nanometer), n_max_neighbors=5)
        nbr.build_from_state(state)
        e = float(pot.compute_energy(state.positions))
        e_nbr = float(pot.compute_energy(state.positions, nbr))
        disp = x[0] - x[1]
        dist = np.linalg.norm(disp)
        e_ref = 4.0 * eps * ((sigma / dist) ** 12 - (sigma / dist) ** 6)
        assert np.isclose(e, e_ref, rtol=g["energy_rtol"], atol=g["energy_atol"])
        assert np.isclose(e_nbr, e_ref, rtol=g["energy_rtol"], atol=g["energy_atol"])
        f = (24 * (eps / (dist * dist)) * (2 * (sigma / dist) ** 12 - (sigma / dist) ** 6)) * disp
        f_ref = np.array([f, -f])
        # the reference's atol=1e-5 sits on top of jnp.allclose's default rtol=1e-5
        for got in (pot.compute_force(state.positions), pot.compute_force(state.positions, nbr),
                    pot.compute_force_analytical(state.positions)):
            assert np.allclose(got.cpu().numpy(), f_ref, atol=g["force_atol"], rtol=1e-5), i


def _check_vs_f64(O, e, f, x, box, sigma, eps, cutoff, periodic=True, pair_filter=None, tol=1e-5):
    e64, f64, eabs, fmax = O.lj_bruteforce_f64(x, box, sigma, eps, cutoff, periodic=periodic, pair_filter=pair_filter)
    assert abs(float(e) - e64) <= tol * max(eabs, 1e-12), (float(e), e64, eabs)
    assert np.abs(f - f64).max() <= tol * max(1.0, fmax), (np.abs(f - f64).max(), fmax)


@pytest.mark.parametrize("cls_name", ["NeighborListNsqrd", "NeighborListCell"])
def test_lj_nbrlist_energy_force(ca, O, cls_name):
    u = ca.unit
    cls = nbr_class(ca, cls_name)
    for name, x, box in _systems(O):
        n = x.shape[0]
        pot = lj_pot(ca, n)
        nbr = cls(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer, skin=SKIN * u.nanometer,
                  n_max_neighbors=180)
        nbr.build(x, box)
        # evaluate at displaced (still inside the skin guarantee) positions
        rng = np.random.default_rng(3)
        x2 = (x + rng.normal(0, 0.03, size=x.shape)).astype(f32)
        e = pot.compute_energy(x2, nbr)
        f = pot.compute_force(x2, nbr).cpu().numpy()
        # vs the fp32 restatement of the reference
        nl_mask, nl, n_nbr, n_max = O.neighborlist_build(x, box, CUTOFF, SKIN, 180)
        e32 = O.lj_energy_nbrlist_f32(x2, nl, nl_mask, box, SIGMA, EPS, CUTOFF)
        f32o = O.lj_force_nbrlist_f32(x2, nl, nl_mask, box, SIGMA, EPS, CUTOFF)
        e64, f64, eabs, fmax = O.lj_bruteforce_f64(x2, box, SIGMA, EPS, CUTOFF)
        assert abs(float(e) - float(e32)) <= 2e-5 * eabs, name
        assert np.abs(f - f32o).max() <= 2e-5 * max(1.0, fmax), name
        _check_vs_f64(O, e, f, x2, box, SIGMA, EPS, CUTOFF)
        e2, f2 = pot.compute_energy_and_force(x2, nbr)
        assert float(e2) == float(e) and np.array_equal(f2.cpu().numpy(), f)


def test_lj_stale_list_uses_list_pairs_only(ca, O):
    """Pairs that drift inside the cutoff without being in the list do not interact (reference semantics)."""
    u = ca.unit
    x, box = O.lj_lattice(800, 0.2, SIGMA, seed=31)
    nbr = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=0.05 * u.nanometer, n_max_neighbors=300)
    nbr.build(x, box)
    rng = np.random.default_rng(8)
    x2 = (x + rng.normal(0, 0.04, size=x.shape)).astype(f32)  # beyond skin/2 = 0.025 nm for most particles
    assert nbr.check(x2) is True
    pot = lj_pot(ca, 800)
    e = pot.compute_energy(x2, nbr)
    f = pot.compute_force(x2, nbr).cpu().numpy()
    nl_mask, nl, n_nbr, _ = O.neighborlist_build(x, box, CUTOFF, 0.05, 300)
    listed = np.zeros((800, 800), bool)
    for i in range(800):
        listed[i, nl[i, : n_nbr[i]]] = True
    listed |= listed.T
    # randomly displaced particles make close contacts (|f| ~ 1e2): fp32 conditioning, not the pair set, sets the error
    _check_vs_f64(O, e, f, x2, box, SIGMA, EPS, CUTOFF, pair_filter=lambda ii, jj: listed[ii, jj], tol=3e-5)


def test_lj_pairlist_and_nolist(ca, O, methane_coords, golden):
    u = ca.unit
    x, box = O.lj_lattice(512, 0.1, SIGMA, seed=1234)  # config C2
    pot = lj_pot(ca, 512)
    pl = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer)
    pl.build(x, box)
    e = pot.compute_energy(x, pl)
    f = pot.compute_force(x, pl).cpu().numpy()
    _check_vs_f64(O, e, f, x, box, SIGMA, EPS, CUTOFF)
    # no list: no PBC
    e = pot.compute_energy(x)
    f = pot.compute_force(x).cpu().numpy()
    _check_vs_f64(O, e, f, x, None, SIGMA, EPS, CUTOFF, periodic=False)
    assert np.isclose(float(e), float(O.lj_energy_nopbc_f32(x, SIGMA, EPS, CUTOFF)), rtol=2e-5)
    # methane fixture (Examples/LJ_MCMC.py): dense liquid
    g = golden["methane"]
    L = g["box_edge"]
    box = np.diag([L, L, L]).astype(f32)
    eps = g["epsilon_kcal_mol"] * 4.184
    xm = methane_coords.astype(f32)
    pot = lj_pot(ca, 1100, g["sigma"], eps, g["cutoff"])
    nbr = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=g["cutoff"] * u.nanometer,
                                        skin=0.4 * u.nanometer, n_max_neighbors=200)
    nbr.build(xm, box)
    e = pot.compute_energy(xm, nbr)
    f = pot.compute_force(xm, nbr).cpu().numpy()
    _check_vs_f64(O, e, f, xm, box, g["sigma"], eps, g["cutoff"])
    assert abs(float(e) - g["build_crosscheck_energy_kj_mol"]) < 0.2


def test_lj_nl_direct_half_list_kernel(ca, O):
    """chx_lj_nl_energy_force: energy/force straight from (neighbor_list, n_neighbors) -- the arrays the
    reference itself would hand over -- agree with the fp64 brute force and with the force-list path."""
    u = ca.unit
    for seed, n, rho in ((5, 700, 0.3), (6, 1500, 0.05)):
        x, box = O.lj_lattice(n, rho, SIGMA, seed=seed)
        for cls_name in ("NeighborListNsqrd", "NeighborListCell"):
            nbr = nbr_class(ca, cls_name)(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                                  skin=SKIN * u.nanometer, n_max_neighbors=180)
            nbr.build(x, box)
            x2 = (x + np.random.default_rng(seed).normal(0, 0.03, size=x.shape)).astype(f32)
            f, e = ca.ops.lj_nl_energy_force(dev(x2), nbr.neighbor_list, nbr.n_neighbors, nbr._box_h, SIGMA, EPS, CUTOFF)
            _check_vs_f64(O, e[0], f.cpu().numpy(), x2, box, SIGMA, EPS, CUTOFF)
            pot = lj_pot(ca, n)
            e_fl, f_fl = pot.compute_energy_and_force(x2, nbr)
            assert np.isclose(float(e[0]), float(e_fl), rtol=2e-6)
            assert np.abs(f.cpu().numpy() - f_fl.cpu().numpy()).max() < 1e-4 * max(1.0, np.abs(f_fl.cpu().numpy()).max())
            f_only, none = ca.ops.lj_nl_energy_force(dev(x2), nbr.neighbor_list, nbr.n_neighbors, nbr._box_h, SIGMA, EPS,
                                                     CUTOFF, want_energy=False)
            assert none is None and np.allclose(f_only.cpu().numpy(), f.cpu().numpy(), atol=1e-3)


def test_baoab_step_entry_point_bitexact(ca, O):
    """chx_baoab_step on user-order arrays: B-A-O-A, wrap and the skin flag, bit for bit against the fp32
    restatement of chiron/integrators.py:181-188, :239-241 with injected noise."""
    from chiron_amd.rng import PRNGKey

    n = 777
    x, box = O.lj_lattice(n, 0.4, SIGMA, seed=12)
    rng = np.random.default_rng(8)
    mass = rng.uniform(1.0, 40.0, n).astype(f32)
    kT, a, b = O.langevin_constants(300.0, 0.002, 1.0)
    v = O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))
    F = rng.normal(0, 300.0, size=(n, 3)).astype(f32)
    noise = rng.normal(size=(n, 3)).astype(f32)
    xo, vo = O.baoa_half(x, v, F, mass, 0.002, a, b, kT, noise)
    xo = O.periodic_wrap(xo, box)
    bx = ca.ops.make_box(box, True)
    key = PRNGKey(3, 4)
    for skin in (10.0, 1e-4):
        xd, vd = dev(x).clone(), dev(v).clone()
        flag = ca.ops.baoab_step(xd, vd, dev(F), dev(mass), 0.001, float(a), float(b), float(np.float32(kT)), bx, True,
                                 dev(x), skin, key, 0, dev(noise))
        assert np.array_equal(xd.cpu().numpy(), xo) and np.array_equal(vd.cpu().numpy(), vo)
        assert bool(flag.item()) == O.neighborlist_check(xo, x, box, skin)
    # device noise: same stream as the integrator (Philox stream 1 at counter (particle, step))
    xd, vd = dev(x).clone(), dev(v).clone()
    ca.ops.baoab_step(xd, vd, dev(F), dev(mass), 0.001, float(a), float(b), float(np.float32(kT)), bx, True, None, 1.0, key, 7)
    z = O.philox_normals3(n, 7, O.STREAM_LANGEVIN, key)
    x1, v1 = O.baoa_half(x, v, F, mass, 0.002, a, b, kT, z)
    assert np.allclose(vd.cpu().numpy(), v1, rtol=1e-4, atol=1e-5)
    assert np.allclose(xd.cpu().numpy(), O.periodic_wrap(x1, box), atol=1e-5)


def test_lj_empty_and_tiny(ca):
    u = ca.unit
    pot = lj_pot(ca, 2)
    x = np.array([[0, 0, 0], [5, 0, 0]], f32)  # beyond cutoff: energy 0 (potential.py:249-250)
    assert float(pot.compute_energy(x)) == 0.0
    assert np.array_equal(pot.compute_force(x).cpu().numpy(), np.zeros((2, 3), f32))
    pot1 = lj_pot(ca, 1)
    assert float(pot1.compute_energy(np.zeros((1, 3), f32))) == 0.0


# ---------------------------------------------------------------------------
# RNG stream, velocities
# ---------------------------------------------------------------------------
def test_velocity_init_matches_oracle_stream(ca, O):
    from chiron_amd.rng import PRNGKey

    key = PRNGKey(0xDEADBEEF, 0x1234)
    n = 4096
    mass = np.full(n, 39.948, f32)
    kT = O.R_KJ * 300.0
    v = ca.ops.init_velocities(dev(mass), kT, key).cpu().numpy()
    z = O.philox_normals3(n, 0, O.STREAM_VELOCITY, key)
    v0 = O.initial_velocities(mass, kT, z)
    assert np.allclose(v, v0, rtol=2e-5, atol=2e-6)
    # injected normals: bit-exact
    v = ca.ops.init_velocities(dev(mass), kT, key, dev(z)).cpu().numpy()
    assert np.array_equal(v, v0)
    # Maxwell-Boltzmann: <m v^2> = kT per degree of freedom
    assert abs((39.948 * v ** 2).mean() / kT - 1.0) < 0.03


# ---------------------------------------------------------------------------
# Langevin
# ---------------------------------------------------------------------------
def _langevin_setup(ca, O, n, rho, seed, skin, dt_fs, nbr_cls="NeighborListCell", m=180):
    u = ca.unit
    x, box = O.lj_lattice(n, rho, SIGMA, seed=seed)
    pot = lj_pot(ca, n)
    state = make_state(ca, x, box)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    nbr = nbr_class(ca, nbr_cls)(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=skin * u.nanometer, n_max_neighbors=m)
    integ = ca.integrators.LangevinIntegrator(timestep=dt_fs * u.femtoseconds, collision_rate=1.0 / u.picoseconds,
                                              chunk_size=7)
    return x, box, pot, state, thermo, nbr, integ


def test_langevin_injected_noise_matches_oracle(ca, O):
    n, steps = 600, 12
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.6, 41, SKIN, 2.0)
    rng = np.random.default_rng(99)
    noise = rng.normal(size=(steps, n, 3)).astype(f32)
    mass = np.full(n, 39.948, f32)
    kT = O.R_KJ * 300.0
    v0 = O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))
    state.velocities = v0
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr, noise=noise)
    xo, vo, rebuilds, _ = O.langevin_run(x, v0, box, mass, SIGMA, EPS, CUTOFF, SKIN, 180, 0.002, 1.0, 300.0, steps,
                                         lambda s: noise[s])
    assert integ.n_rebuilds == rebuilds
    assert np.abs(out.positions.cpu().numpy() - xo).max() < 2e-6
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 2e-4 * max(1.0, np.abs(vo).max())


def test_langevin_box_crossings_and_unwrapped_input_match_oracle(ca, O):
    """The step kernel's epilogue runs a branch-free form for 0 <= x < L, |x - ref| < L and redoes a
    particle through the general wrap / minimum-image formulas otherwise: start from UNWRAPPED
    coordinates (images up to +-3 boxes away, so |x - ref| >= L after the first wrap) with a hot, fast
    system whose particles cross the box faces, and follow the oracle through both branches."""
    n, steps, skin = 700, 25, 0.3
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.5, 45, skin, 4.0)
    rng = np.random.default_rng(17)
    L = np.diag(box).astype(f32)
    x = (x + rng.integers(-3, 4, size=(n, 3)).astype(f32) * L).astype(f32)
    state = make_state(ca, x, box)
    noise = rng.normal(size=(steps, n, 3)).astype(f32)
    mass = np.full(n, 39.948, f32)
    kT = O.R_KJ * 300.0
    v0 = (3.0 * O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))).astype(f32)
    state.velocities = v0
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr, noise=noise)
    xo, vo, rebuilds, _ = O.langevin_run(x, v0, box, mass, SIGMA, EPS, CUTOFF, skin, 180, 0.004, 1.0, 300.0, steps,
                                         lambda s: noise[s])
    assert integ.n_rebuilds == rebuilds
    got = out.positions.cpu().numpy()
    assert (got >= 0).all() and (got <= L).all()
    # particles that left through a face came back in through the opposite one
    x_w = x - np.floor(x / L) * L
    assert (np.abs(xo - x_w) > 0.5 * L).any(), "test needs box crossings to be meaningful"
    assert np.abs(got - xo).max() < 5e-6
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 5e-4 * max(1.0, np.abs(vo).max())


@pytest.mark.parametrize("inner,every", [(0.03, 4), (0.002, 8), (0.01, 1000)], ids=["pruned", "mostly-stale", "never-refreshed"])
def test_langevin_pruned_inner_list_keeps_the_reference_pairs(ca, O, monkeypatch, inner, every):
    """The dynamically pruned inner list is a speed device only: with it on -- including when it
    goes stale and the kernel falls back to the full list -- the trajectory follows the oracle
    (injected noise) and the list rebuilds happen at the reference's steps."""
    n, steps, skin = 2500, 40, 0.05
    monkeypatch.setenv("CHX_INNER_SKIN", str(inner))
    monkeypatch.setenv("CHX_PRUNE_EVERY", str(every))
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    rng = np.random.default_rng(5)
    noise = rng.normal(size=(steps, n, 3)).astype(f32)
    mass = np.full(n, 39.948, f32)
    kT = O.R_KJ * 300.0
    v0 = O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))
    state.velocities = v0
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr, noise=noise)
    xo, vo, rebuilds, _ = O.langevin_run(x, v0, box, mass, SIGMA, EPS, CUTOFF, skin, 180, 0.004, 1.0, 300.0, steps,
                                         lambda s_: noise[s_])
    assert integ.n_rebuilds == rebuilds and rebuilds >= 1
    assert integ.prune_passes >= 1
    if inner < 0.005 or every > steps:
        assert integ.full_list_launches > 0  # the stale-list fallback really ran
    else:
        assert integ.full_list_launches <= steps // 4  # the pruned table carried (nearly) every launch
    assert np.abs(out.positions.cpu().numpy() - xo).max() < 5e-6
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 5e-4 * max(1.0, np.abs(vo).max())
    # and the run is still independent of how the steps are batched into launches
    x2, box2, pot2, state2, thermo2, nbr2, integ2 = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    integ2.chunk_size = 32
    state2.velocities = v0
    out2, _ = integ2.run(state2, thermo2, number_of_steps=steps, nbr_list=nbr2, noise=noise)
    assert torch.equal(out.positions, out2.positions) and torch.equal(out.velocities, out2.velocities)
    assert (integ2.prune_passes, integ2.full_list_launches) == (integ.prune_passes, integ.full_list_launches)


@pytest.mark.parametrize("sabotage", [False, True], ids=["accepted", "rejected"])
def test_langevin_speculative_rebuild(ca, O, monkeypatch, sabotage):
    """Rebuilds inside run() do not wait for the build's status block (the reference's sync,
    chiron/neighbors.py:709): steps are enqueued on the new list at once and the block is read a
    chunk later.  With the block clean the trajectory is the oracle's; when a capacity overflowed
    (here: the first rebuild is handed a four-entry force list) the speculative list and everything run
    on it are thrown away, the list is rebuilt the synchronous way and the trajectory is still the
    oracle's."""
    import os

    if os.environ.get("CHX_SYNC_REBUILD") == "1":
        pytest.skip("speculative rebuilds are switched off (CHX_SYNC_REBUILD=1)")
    n, steps, skin = 2500, 40, 0.05
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    rng = np.random.default_rng(5)
    noise = rng.normal(size=(steps, n, 3)).astype(f32)
    mass = np.full(n, 39.948, f32)
    v0 = O.initial_velocities(mass, O.R_KJ * 300.0, rng.normal(size=(n, 3)).astype(f32))
    state.velocities = v0
    calls = {"spec": 0, "rejected": 0}
    cls = type(nbr)
    orig_spec, orig_confirm, orig_geo = cls._build_speculative, cls._confirm_build, cls._fl_geometry

    def spec(self, positions, **kw):
        calls["spec"] += 1
        if sabotage and calls["spec"] == 1:
            monkeypatch.setattr(cls, "_fl_geometry", lambda s, n_: (1, 1))  # room for four entries per row
            try:
                return orig_spec(self, positions, **kw)
            finally:
                monkeypatch.setattr(cls, "_fl_geometry", orig_geo)
        return orig_spec(self, positions, **kw)

    def confirm(self, **kw):
        pending = getattr(self, "_spec", None) is not None
        ok = orig_confirm(self, **kw)
        if pending and not ok:
            calls["rejected"] += 1
        return ok

    monkeypatch.setattr(cls, "_build_speculative", spec)
    monkeypatch.setattr(cls, "_confirm_build", confirm)
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr, noise=noise)
    xo, vo, rebuilds, (nl_mask, nl, n_max) = O.langevin_run(x, v0, box, mass, SIGMA, EPS, CUTOFF, skin, 180, 0.004, 1.0,
                                                            300.0, steps, lambda s_: noise[s_])
    assert rebuilds >= 1 and integ.n_rebuilds == rebuilds
    assert calls["spec"] == rebuilds and calls["rejected"] == (1 if sabotage else 0)
    assert np.abs(out.positions.cpu().numpy() - xo).max() < 5e-6
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 5e-4 * max(1.0, np.abs(vo).max())
    assert nbr.n_max_neighbors == n_max
    assert np.array_equal(u32(nbr.neighbor_list)[:, :nl.shape[1]], nl)


def test_langevin_rebuild_step_matches_oracle(ca, O):
    """Tiny skin + large timestep: the device flag must stop the chunk at the reference's rebuild step."""
    n, steps = 400, 30
    skin = 0.02
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0, "NeighborListNsqrd", 250)
    rng = np.random.default_rng(5)
    noise = rng.normal(size=(steps, n, 3)).astype(f32)
    mass = np.full(n, 39.948, f32)
    kT = O.R_KJ * 300.0
    v0 = O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))
    state.velocities = v0
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr, noise=noise)
    xo, vo, rebuilds, (nl_mask, nl, n_max) = O.langevin_run(x, v0, box, mass, SIGMA, EPS, CUTOFF, skin, 250, 0.004,
                                                            1.0, 300.0, steps, lambda s: noise[s])
    assert rebuilds >= 2, "test needs rebuilds to be meaningful"
    assert integ.n_rebuilds == rebuilds
    assert np.abs(out.positions.cpu().numpy() - xo).max() < 1e-5
    assert nbr.n_max_neighbors == n_max
    assert np.array_equal(u32(nbr.neighbor_list), nl)  # same list => same rebuild steps and same ref positions


def test_langevin_device_rng_matches_oracle_stream(ca, O):
    n, steps = 300, 5
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.5, 47, SKIN, 1.0)
    mass = np.full(n, 39.948, f32)
    state.velocities = np.zeros((n, 3), f32)
    import chiron_amd.rng as R

    key_before = state.current_PRNG_key
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    _, run_key = R.split(key_before)  # what new_PRNG_key handed to run()
    xo, vo, _, _ = O.langevin_run(x, np.zeros((n, 3), f32), box, mass, SIGMA, EPS, CUTOFF, SKIN, 180, 0.001, 1.0, 300.0,
                                  steps, lambda s: O.philox_normals3(n, s, O.STREAM_LANGEVIN, run_key))
    assert np.abs(out.positions.cpu().numpy() - xo).max() < 2e-6
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 1e-4


@pytest.mark.parametrize("n,rho", [(200, 0.5), (2100, 0.5), (512, 0.1)],
                         ids=["wave-per-particle", "16-lanes-per-particle", "c2-n512-rho0.1"])
def test_langevin_nolist_and_pairlist(ca, O, n, rho):
    """The paths without a Verlet list run ONE launch per step too (k_ap_step: all-pairs force + BAOAB;
    a wavefront per particle up to N = 2048, 16 lanes per particle beyond): nbr_list=None is the
    reference's no-PBC branch (chiron/potential.py:235-258), PairListNsqrd the PBC all-pairs list with
    a wrap every step and no rebuilds (chiron/neighbors.py:1272-1289).  Both follow the oracle with
    injected noise; the PBC trajectory is the neighbor-list trajectory (same pairs interact).  The third case is
    BASELINE.json's C2 itself: N = 512 at rho* = 0.1 (box edge 5.8601 nm, Examples/LJ_langevin.py)."""
    u = ca.unit
    steps = 6
    x, box = O.lj_lattice(n, rho, SIGMA, seed=51)
    if (n, rho) == (512, 0.1):
        assert abs(float(box[0, 0]) - 5.8601) < 1e-3
    pot = lj_pot(ca, n)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    rng = np.random.default_rng(1)
    noise = rng.normal(size=(steps, n, 3)).astype(f32)
    mass = np.full(n, 39.948, f32)
    integ = ca.integrators.LangevinIntegrator(timestep=1.0 * u.femtoseconds, chunk_size=4)
    state = make_state(ca, x, box)
    state.velocities = np.zeros((n, 3), f32)
    out, _ = integ.run(state, thermo, number_of_steps=steps, nbr_list=None, noise=noise)
    xo, vo, _, _ = O.langevin_run(x, np.zeros((n, 3), f32), box, mass, SIGMA, EPS, CUTOFF, SKIN, 180, 0.001, 1.0, 300.0,
                                  steps, lambda s: noise[s], use_nbr_list=False)
    assert np.abs(out.positions.cpu().numpy() - xo).max() < 2e-6
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 1e-4
    # PairListNsqrd: PBC all-pairs + wrap each step
    pl = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer)
    state = make_state(ca, x, box)
    state.velocities = np.zeros((n, 3), f32)
    out, pl = integ.run(state, thermo, number_of_steps=steps, nbr_list=pl, noise=noise)
    xw = out.positions.cpu().numpy()
    assert np.isfinite(xw).all() and (xw >= 0).all() and (xw <= box[0, 0]).all()
    xo, vo, _, _ = O.langevin_run(x, np.zeros((n, 3), f32), box, mass, SIGMA, EPS, CUTOFF, SKIN, 400, 0.001, 1.0, 300.0,
                                  steps, lambda s: noise[s], use_nbr_list=True)
    assert np.abs(xw - xo).max() < 3e-6
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 2e-4
    # an odd number of steps leaves the state in the other position buffer; a second run continues from it
    out2, pl = integ.run(out, thermo, number_of_steps=3, nbr_list=pl, noise=noise[:3])
    assert np.isfinite(out2.positions.cpu().numpy()).all()


def test_langevin_ideal_gas_external_force_path(ca):
    u = ca.unit
    ig = ca.testsystems.IdealGas(64)
    pot = ca.potential.IdealGasPotential(ig.topology)
    thermo = ca.states.ThermodynamicState(pot, temperature=298 * u.kelvin)
    ca.utils.PRNG.set_seed(1234)
    state = ca.states.SamplerState(ig.positions, ca.utils.PRNG.get_random_key(), box_vectors=ig.box_vectors)
    out, _ = ca.integrators.LangevinIntegrator().run(state, thermo, number_of_steps=10)
    assert torch.isfinite(out.positions).all() and out.velocities.shape == (64, 3)


def test_langevin_kinetic_temperature(ca, O):
    """Statistical pin of BAOAB with the device PRNG: <m v^2> = kT per dof after equilibration."""
    n = 4096
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.1, 1234, SKIN, 2.0)
    integ = ca.integrators.LangevinIntegrator(timestep=2.0 * ca.unit.femtoseconds,
                                              collision_rate=5.0 / ca.unit.picoseconds)
    out, nbr = integ.run(state, thermo, number_of_steps=1500, nbr_list=nbr)
    kT = O.R_KJ * 300.0
    v = out.velocities.cpu().numpy()
    assert abs((39.948 * v ** 2).mean() / kT - 1.0) < 0.04
    assert integ.n_rebuilds >= 1


def test_reference_workflow_langevin_lj_fluid(ca, O, tmp_path):
    """chiron/tests/test_convergence_tests.py:128-184 (skipped upstream as too slow for JAX-CPU) in the
    reference's own shape: LennardJonesFluid(reduced_density=0.1, nparticles=5000), NeighborListNsqrd
    (cutoff 3 sigma, skin 0.5 nm, M = 180) built from the state, LangevinIntegrator with a
    LangevinDynamicsReporter every 100 steps, 1000 steps with the progress bar on.  Checked here:
    the reporter's frames, the box, the thermostat, that the input state is not advanced (only its PRNG
    key, as in the reference), and that the same key reproduces the run bit for bit."""
    from chiron_amd.reporters import BaseReporter, LangevinDynamicsReporter
    from chiron_amd.testsystems import LennardJonesFluid

    u = ca.unit
    fluid = LennardJonesFluid(reduced_density=0.1, nparticles=5000)
    cutoff, skin = 3 * 0.34 * u.nanometer, 0.5 * u.nanometer
    pot = ca.potential.LJPotential(fluid.topology, sigma=0.34 * u.nanometer, cutoff=cutoff)
    thermo = ca.states.ThermodynamicState(potential=pot, temperature=300 * u.kelvin)
    BaseReporter.set_directory(str(tmp_path))

    def run_once():
        ca.utils.PRNG.set_seed(1234)
        state = ca.states.SamplerState(positions=fluid.positions, box_vectors=fluid.system.getDefaultPeriodicBoxVectors(),
                                       current_PRNG_key=ca.utils.PRNG.get_random_key())
        nbr = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=cutoff, skin=skin,
                                             n_max_neighbors=180)
        nbr.build_from_state(state)
        reporter = LangevinDynamicsReporter("lj_fluid_workflow")
        integ = ca.integrators.LangevinIntegrator(reporter=reporter, report_interval=100)
        x_before, key_before = state.positions.clone(), state.current_PRNG_key
        out, nbr_out = integ.run(state, thermo, number_of_steps=1000, nbr_list=nbr, progress_bar=True)
        assert torch.equal(state.positions, x_before) and state.current_PRNG_key != key_before
        reporter.flush_buffer()
        return out, nbr_out, reporter, integ

    out, nbr, reporter, integ = run_once()
    steps = np.asarray(reporter.get_property("step")).ravel()
    assert steps.tolist() == list(range(0, 1000, 100))
    energy = np.asarray(reporter.get_property("potential_energy"), dtype=np.float64).ravel()
    frames = np.asarray(reporter.get_property("positions"))
    assert energy.shape == (10,) and np.isfinite(energy).all() and frames.shape == (10, 5000, 3)
    L = float(out.box_vectors[0, 0])
    x = out.positions.cpu().numpy()
    assert (x >= 0).all() and (x <= L).all()
    # the reported energy of a frame is the potential's energy of that frame (fresh list, fp64 oracle)
    nbr9 = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=cutoff, skin=skin, n_max_neighbors=180)
    nbr9.build(frames[-1], out.box_vectors)
    e_again = float(pot.compute_energy(torch.as_tensor(frames[-1], device=out.positions.device), nbr9))
    assert abs(e_again - energy[-1]) <= 2e-5 * abs(energy[-1]) + 1e-3
    assert pot.epsilon == pytest.approx(4.184)  # the reference's default: 1 kcal/mol (chiron/potential.py:136)
    e64, _, abs_e, _ = O.lj_bruteforce_f64(frames[-1], out.box_vectors.cpu().numpy(), SIGMA, pot.epsilon, CUTOFF)
    assert abs(energy[-1] - e64) <= 1e-5 * max(abs_e, 1.0)
    # eps = 1 kcal/mol puts 300 K below the LJ critical point (T* = 0.6): the lattice gas condenses, the
    # potential energy falls and the 1/ps thermostat has not yet carried the heat away after 1 ps (the
    # equilibrium <m v^2> = kT is pinned by test_langevin_kinetic_temperature)
    v = out.velocities.cpu().numpy()
    assert 0.9 < (39.948 * v ** 2).mean() / (O.R_KJ * 300.0) < 1.6
    assert energy[-1] < energy[0]
    assert nbr.is_built and integ.n_rebuilds >= 2
    # the list handed back is the reference's list of its last rebuild: ascending ids, padding rule, mask
    nl_mask, nl, n_nbr, n_max = O.neighborlist_build(nbr.ref_positions.cpu().numpy(), out.box_vectors.cpu().numpy(),
                                                     1.02, 0.5, nbr.n_max_neighbors)
    assert n_max == nbr.n_max_neighbors and np.array_equal(u32(nbr.neighbor_list), nl)
    assert np.array_equal(nbr.neighbor_mask.cpu().numpy(), nl_mask)
    out2, _, reporter2, _ = run_once()
    assert torch.equal(out.positions, out2.positions) and torch.equal(out.velocities, out2.velocities)
    assert np.array_equal(np.asarray(reporter2.get_property("potential_energy")), np.asarray(reporter.get_property("potential_energy")))


# ---------------------------------------------------------------------------
# multistate (chiron/multistate.py) on the real kernels
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("swap", [None, "neighbors"], ids=["reference-no-swap", "neighbor-swaps"])
def test_multistate_sampler_on_lj_replicas(ca, O, swap):
    """chiron/tests/test_multistate.py:211-251 in shape (MultiStateSampler over K thermodynamic states,
    one MCMCSampler with a LangevinDynamicsMove per state, energy matrix every iteration), with LJ
    replicas on the HIP kernels instead of harmonic oscillators: after every iteration
    u_kn[k, r] = beta_k U(x_r) for the replicas' CURRENT positions (fp64 brute-force oracle), each
    replica keeps its own neighbor list, and with swaps off the state assignment never changes
    (the reference's placeholder, multistate.py:447-460)."""
    from chiron_amd.mcmc import LangevinDynamicsMove, MCMCSampler, MoveSchedule
    from chiron_amd.multistate import MultiStateSampler

    u = ca.unit
    K, n = 3, 1500
    x, box = O.lj_lattice(n, 0.3, SIGMA, seed=7)
    pot = lj_pot(ca, n)
    temps = [300.0, 330.0, 360.0]
    thermo = [ca.states.ThermodynamicState(pot, temperature=t * u.kelvin) for t in temps]
    states, lists = [], []
    for k in range(K):
        from chiron_amd.rng import PRNGKey

        st = make_state(ca, x, box)
        st.current_PRNG_key = PRNGKey(1234, k)
        states.append(st)
        lists.append(ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                                    skin=SKIN * u.nanometer, n_max_neighbors=180))
    move = LangevinDynamicsMove(timestep=2.0 * u.femtoseconds, number_of_steps=60)
    ms = MultiStateSampler(MCMCSampler(MoveSchedule([("LangevinDynamicsMove", move)])), swap=swap)
    ms.create(thermo, states, lists)
    ms.run(n_iterations=3, n_mcmc_iterations=1)
    assert ms.iteration == 3
    u_kn = ms._reporter.get_property("u_kn")
    assert u_kn.shape == (4, K, K)
    finals = ms.sampler_states
    assign = list(ms._replica_thermodynamic_states)
    assert sorted(assign) == list(range(K))
    if swap is None:
        assert assign == list(range(K))
    xs = [st.positions.cpu().numpy() for st in finals]
    assert not np.array_equal(xs[0], xs[1])  # the replicas really are separate trajectories
    for r in range(K):
        e64, _, eabs, _ = O.lj_bruteforce_f64(xs[r], box, SIGMA, EPS, CUTOFF)
        for k in range(K):
            want = O.reduced_potential(e64, temps[k])
            assert abs(u_kn[-1][k, r] - want) <= 2e-5 * O.reduced_potential(eabs, temps[k]) + 1e-6
        # every replica's list is its own, built from its own positions
        nb = ms._nbr_lists[r]
        d = nb.ref_positions.cpu().numpy() - xs[r]
        d -= np.diag(box) * np.round(d / np.diag(box))
        assert nb.is_built and np.sqrt((d ** 2).sum(1)).max() < 0.26  # within skin / 2 (+ a step) of its own replica
    # hotter states hold more kinetic energy (velocities follow the state a replica is assigned to)
    ke = [float((st.velocities ** 2).mean()) for st in finals]
    assert all(k_ > 0 for k_ in ke)


@pytest.mark.parametrize("with_mc", [False, True], ids=["langevin", "langevin+mc"])
def test_multistate_concurrent_replicas_give_the_serial_results(ca, O, with_mc):
    """``MultiStateSampler(concurrent_replicas=2)`` propagates two replicas side by side (two threads, two HIP streams:
    a step of a medium-size system leaves the card mostly idle).  Replicas are independent between mixing steps, so
    positions, velocities, PRNG keys, neighbor lists, state assignment and the whole u_kn history must be the serial
    loop's bits -- five replicas (an odd number: the lanes get unequal shares), swaps on, thin skin (rebuilds inside
    every segment, so both threads sit in the C launch loop and in the rebuild path at the same time)."""
    from chiron_amd.mcmc import LangevinDynamicsMove, MCMCSampler, MoveSchedule
    from chiron_amd.multistate import MultiStateSampler
    from chiron_amd.rng import PRNGKey

    u = ca.unit
    K, n = 5, 2500
    x, box = O.lj_lattice(n, 0.3, SIGMA, seed=7)
    pot = lj_pot(ca, n)
    temps = [300.0 * 1.08 ** k for k in range(K)]

    def run(workers):
        thermo = [ca.states.ThermodynamicState(pot, temperature=t * u.kelvin) for t in temps]
        states, lists = [], []
        for k in range(K):
            st = make_state(ca, x, box)
            st.current_PRNG_key = PRNGKey(4321, k)
            states.append(st)
            lists.append(ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                                       skin=0.06 * u.nanometer, n_max_neighbors=180))
        move = LangevinDynamicsMove(timestep=4.0 * u.femtoseconds, number_of_steps=120)
        schedule = [("LangevinDynamicsMove", move)]
        if with_mc:  # (the device-resident Monte Carlo loop in both lanes at once: its key chain is made on the host)
            schedule.append(("MC", ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=0.004 * u.nanometer, number_of_moves=40)))
        # (batch_replicas=False: this test is about the threads-and-streams path; the batched path has tests/test_gpu_batch.py)
        ms = MultiStateSampler(MCMCSampler(MoveSchedule(schedule)), swap="neighbors", concurrent_replicas=workers,
                               batch_replicas=False)
        ms.create(thermo, states, lists)
        ms.run(n_iterations=4, n_mcmc_iterations=1)
        torch.cuda.synchronize()
        return ms

    a, b = run(1), run(2)
    assert np.array_equal(a._reporter.get_property("u_kn"), b._reporter.get_property("u_kn"))
    assert list(a._replica_thermodynamic_states) == list(b._replica_thermodynamic_states)
    for sa, sb, la, lb in zip(a._sampler_states, b._sampler_states, a._nbr_lists, b._nbr_lists):
        assert torch.equal(sa.positions, sb.positions) and torch.equal(sa.velocities, sb.velocities)
        assert tuple(sa.current_PRNG_key) == tuple(sb.current_PRNG_key)
        assert torch.equal(la.ref_positions, lb.ref_positions) and torch.equal(la.neighbor_list, lb.neighbor_list)
        assert la.n_builds == lb.n_builds and la.n_builds > 5  # rebuilds happened inside the segments


# ---------------------------------------------------------------------------
# Monte Carlo
# ---------------------------------------------------------------------------
def test_mc_displace_matches_oracle(ca, O):
    from chiron_amd.rng import PRNGKey

    key = PRNGKey(77, 1)
    x, box = O.lj_lattice(500, 0.2, SIGMA, seed=61)
    z = O.philox_normals3(500, 0, O.STREAM_MC_DISPLACE, key)
    bx = ca.ops.make_box(box, True)
    got = ca.ops.mc_displace(dev(x), 0.05, bx, key, 0, None, dev(z)).cpu().numpy()
    assert np.array_equal(got, O.mc_displace(x, 0.05, z, box))
    mask = (np.arange(500) % 3 == 0).astype(f32)
    got = ca.ops.mc_displace(dev(x), 0.05, bx, key, 0, dev(mask), dev(z)).cpu().numpy()
    assert np.array_equal(got, O.mc_displace(x, 0.05, z, box, mask))
    got = ca.ops.mc_displace(dev(x), 0.05, bx, key, 0).cpu().numpy()  # device Philox
    assert np.allclose(got, O.mc_displace(x, 0.05, z, box), atol=2e-6)
    got = ca.ops.mc_displace(dev(x), 0.05, ca.ops.make_box(None, False), key, 0, None, dev(z)).cpu().numpy()
    assert np.array_equal(got, O.mc_displace(x, 0.05, z, None))


def test_mc_displacement_move_lj(ca, O):
    u = ca.unit
    n = 500
    x, box = O.lj_lattice(n, 0.1, SIGMA, seed=71)
    pot = lj_pot(ca, n)
    state = make_state(ca, x, box)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    nbr = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                        skin=SKIN * u.nanometer, n_max_neighbors=180)
    nbr.build_from_state(state)
    rep = ca_reporter(ca)
    move = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=0.002 * u.nanometer, number_of_moves=40,
                                              reporter=rep, report_interval=1, autotune=True, autotune_interval=20)
    s2, t2, n2 = move.update(state, thermo, nbr)
    st = move.statistics
    assert st["n_proposed"] == 40 and 0 < st["n_accepted"] <= 40
    # Metropolis bookkeeping: the cached reduced potential equals a fresh evaluation of the final state
    assert np.isclose(float(move._current_reduced_potential), float(thermo.get_reduced_potential(s2, n2)), rtol=1e-5)
    assert rep.get_property("potential_energy").shape == (40,)
    # every decision is consistent with the oracle rule on the recorded numbers
    sampler = ca.mcmc.MCMCSampler(ca.mcmc.MoveSchedule([("mc", move)]))
    sampler.run(state, thermo, n_iterations=1, nbr_list=nbr)
    with pytest.raises(ValueError):
        ca.mcmc.MoveSchedule([("bad", 123)])


def _mc_run(ca, O, host_loop, monkeypatch, *, n=500, skin=SKIN, sigma_d=0.002, moves=60, subset=None, autotune=False,
            nbr_cls="NeighborListCell", iterations=1, pressure=None):
    u = ca.unit
    monkeypatch.setenv("CHX_MC_HOST_LOOP", "1" if host_loop else "0")
    x, box = O.lj_lattice(n, 0.1, SIGMA, seed=71)
    pot = lj_pot(ca, n)
    state = make_state(ca, x, box)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin, pressure=pressure)
    nbr = nbr_class(ca, nbr_cls)(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=skin * u.nanometer, n_max_neighbors=180)
    nbr.build_from_state(state)
    rep = ca_reporter(ca)
    move = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=sigma_d * u.nanometer, number_of_moves=moves,
                                              atom_subset=subset, reporter=rep, report_interval=3,
                                              autotune=autotune, autotune_interval=16)
    assert move._device_loop_applies(state, thermo, nbr) == (not host_loop)
    for _ in range(iterations):
        state, thermo, nbr = move.update(state, thermo, nbr)
    return dict(
        x=state.positions.cpu().numpy(), key=tuple(state._current_PRNG_key), stats=move.statistics,
        attempts=move.number_of_attemps_made, u=float(move._current_reduced_potential),
        sigma=move.displacement_sigma.value_in_unit_system(u.md_unit_system),
        ref=nbr.ref_positions.cpu().numpy(), n_nbr=nbr.n_neighbors.cpu().numpy(),
        refreshes=getattr(move, "inner_list_refreshes", 0),
        reports={k: rep.get_property(k) for k in ("step", "iteration", "number_of_attempts_made", "potential_energy",
                                                  "displacement_sigma", "acceptance_probability")},
    )


@pytest.mark.parametrize(
    "kw",
    [
        dict(sigma_d=0.03, moves=30),                             # mixed accept/reject on one list
        dict(skin=0.05, sigma_d=0.01, moves=80),                  # several skin violations -> rebuild branch
        dict(skin=0.05, sigma_d=0.01, moves=50, autotune=True, iterations=2),
        dict(subset=[0, 3, 17, 499], sigma_d=0.05, moves=40),
        dict(nbr_cls="NeighborListNsqrd", sigma_d=0.02, moves=30, pressure=1.0),
    ],
    ids=["plain", "rebuilds", "autotune", "subset", "nsq-npt"],
)
def test_mc_device_loop_equals_host_loop(ca, O, monkeypatch, kw):
    """The chunked device-resident trial loop reproduces the reference-shaped host loop exactly:
    same proposals, energies, decisions, statistics, reports, PRNG chain and list rebuilds."""
    if "pressure" in kw and kw["pressure"] is not None:
        kw = dict(kw, pressure=kw["pressure"] * ca.unit.atmosphere)
    host = _mc_run(ca, O, True, monkeypatch, **kw)
    devr = _mc_run(ca, O, False, monkeypatch, **kw)
    assert devr["stats"] == host["stats"] and devr["attempts"] == host["attempts"]
    assert 0 < host["stats"]["n_accepted"] < host["stats"]["n_proposed"] or kw.get("subset")
    assert devr["key"] == host["key"]
    assert np.array_equal(devr["x"], host["x"])
    assert np.array_equal(devr["ref"], host["ref"]) and np.array_equal(devr["n_nbr"], host["n_nbr"])
    assert devr["sigma"] == host["sigma"]
    assert np.isclose(devr["u"], host["u"], rtol=1e-12)
    for k, v in host["reports"].items():
        assert np.array_equal(np.asarray(devr["reports"][k], dtype=np.float64), np.asarray(v, dtype=np.float64)), k
    if kw.get("skin") == 0.05:
        assert not np.array_equal(host["ref"], O.lj_lattice(500, 0.1, SIGMA, seed=71)[0])  # the list was rebuilt


@pytest.mark.parametrize("kw", [dict(sigma_d=0.003, moves=400), dict(sigma_d=0.03, moves=150),
                                dict(skin=0.05, sigma_d=0.008, moves=300), dict(subset=[0, 3, 17, 499], sigma_d=0.02, moves=200)],
                         ids=["small-moves", "mixed-decisions", "rebuilds", "subset"])
def test_mc_inner_list_changes_no_bit(ca, O, monkeypatch, kw):
    """The trial launches of the device-resident displacement loop walk an inner list (include/chx.h: chx_mc_t::fli --
    the force-list entries within cutoff + inner skin of the proposal it was taken at, compacted per lane and per
    accumulator) and renew it when a particle has strayed inner skin / 2.  Energies, decisions, reports, key chain and
    rebuild trials must be those of the loop without it, and of the host loop, for a wide skin (renewed rarely), a
    narrow one (every few trials) and none."""
    runs = {}
    for tag, s_in in (("off", 0.0), ("wide", 0.1), ("narrow", 0.02)):
        monkeypatch.setattr(ca.mcmc, "_MC_INNER_SKIN", s_in)
        runs[tag] = _mc_run(ca, O, False, monkeypatch, **kw)
    host = _mc_run(ca, O, True, monkeypatch, **kw)
    trials = host["attempts"]
    for tag, r in runs.items():
        assert r["stats"] == host["stats"] and r["key"] == host["key"], tag
        assert np.array_equal(r["x"], host["x"]) and np.array_equal(r["ref"], host["ref"]), tag
        for k, v in host["reports"].items():
            assert np.array_equal(np.asarray(r["reports"][k], dtype=np.float64), np.asarray(v, dtype=np.float64)), (tag, k)
    assert runs["off"]["refreshes"] == 0
    # (every context of the device loop -- one per chunk and per trial that ran through the host's rebuild branch --
    # opens with a refresh, so a run that rebuilds all the time counts about one per trial whatever the inner skin)
    assert 1 <= runs["wide"]["refreshes"] <= 2 * trials and 1 <= runs["narrow"]["refreshes"] <= 2 * trials
    print("inner-list refreshes (wide, narrow, trials):", runs["wide"]["refreshes"], runs["narrow"]["refreshes"], trials)
    if kw["sigma_d"] == 0.003:
        assert runs["wide"]["refreshes"] < trials // 8 < runs["narrow"]["refreshes"]  # rarely / every few trials
    assert host["stats"]["n_accepted"] > 0
    if kw["sigma_d"] == 0.03:
        assert host["stats"]["n_accepted"] < host["stats"]["n_proposed"]


def _decisions_from_reports(rep):
    """Per-trial accept decisions from the reported running acceptance probability (report_interval = 1)."""
    p = np.asarray(rep.get_property("acceptance_probability"), dtype=np.float64)
    acc = np.rint(p * np.arange(1, len(p) + 1)).astype(np.int64)
    return np.diff(np.concatenate([[0], acc])).astype(bool)


@pytest.mark.parametrize("host_loop", [False, True], ids=["device-loop", "host-loop"])
@pytest.mark.parametrize(
    "kw",
    [
        dict(skin=SKIN, sigma_d=0.03, moves=40, iterations=1),
        dict(skin=0.05, sigma_d=0.01, moves=60, iterations=2, autotune=True),
        dict(skin=0.05, sigma_d=0.02, moves=50, iterations=1, subset=[0, 3, 17, 499]),
    ],
    ids=["plain", "rebuilds-autotune", "subset"],
)
def test_mc_displacement_chain_matches_oracle_chain(ca, O, monkeypatch, kw, host_loop):
    """MonteCarloDisplacementMove.update() -- BOTH the device-resident trial loop (csrc/chx_mc.hip) and the
    reference-shaped host loop -- against the oracle's independent restatement of chiron/mcmc.py:243-306,
    :357-463, :680-787 (oracle.mc_displacement_update: own Philox, own key splits, NumPy energies and lists):
    accept/reject sequence, per-trial energies, rebuild trials, autotuned sigma, PRNG key chain and final
    positions.  Pass 1 lets the oracle draw its own normals (libm Box-Muller: proposals agree to 2e-6);
    pass 2 feeds it the device's normals so that every proposal, hence the whole chain, is bit-identical."""
    from chiron_amd.rng import PRNGKey

    u = ca.unit
    n, T = 500, 300.0
    monkeypatch.setenv("CHX_MC_HOST_LOOP", "1" if host_loop else "0")
    x0, box = O.lj_lattice(n, 0.1, SIGMA, seed=71)
    pot = lj_pot(ca, n)
    state = make_state(ca, x0, box)
    key0 = tuple(state._current_PRNG_key)
    thermo = ca.states.ThermodynamicState(pot, temperature=T * u.kelvin)
    nbr = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                        skin=kw["skin"] * u.nanometer, n_max_neighbors=180)
    nbr.build_from_state(state)
    rep = ca_reporter(ca)
    move = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=kw["sigma_d"] * u.nanometer, number_of_moves=kw["moves"],
                                              atom_subset=kw.get("subset"), reporter=rep, report_interval=1,
                                              autotune=kw.get("autotune", False), autotune_interval=16)
    assert move._device_loop_applies(state, thermo, nbr) == (not host_loop)
    st, nb = state, nbr
    for _ in range(kw["iterations"]):
        st, thermo, nb = move.update(st, thermo, nb)
    got_x = st.positions.cpu().numpy()
    got_dec = _decisions_from_reports(rep)
    got_e = np.asarray(rep.get_property("potential_energy"), dtype=np.float64)
    got_sigma = np.asarray(rep.get_property("displacement_sigma"), dtype=np.float64)
    total = kw["moves"] * kw["iterations"]
    assert len(got_dec) == total

    mask = None
    if kw.get("subset"):
        mask = np.zeros(n, f32)
        mask[kw["subset"]] = 1.0
    open_box = ca.ops.make_box(None, False)
    zeros = torch.zeros((n, 3), dtype=torch.float32, device="cuda")

    def device_normals(sub):  # x' = 0 + 1.0 * z, no wrap: the kernel's own Box-Muller output, exactly
        return ca.ops.mc_displace(zeros, 1.0, open_box, PRNGKey(*sub), 0).cpu().numpy()

    e64, _, eabs, _ = O.lj_bruteforce_f64(x0, box, SIGMA, EPS, CUTOFF)
    for normals_fn in (None, device_normals):
        x, key, sig = x0.copy(), key0, kw["sigma_d"]
        lst = O.McList(x0, box, CUTOFF, kw["skin"], 180)
        counters, log = (0, 0, 0), []
        for _ in range(kw["iterations"]):
            x, lst, key, sig, counters, lg = O.mc_displacement_update(
                x, box, lst, key, SIGMA, EPS, T, sig, kw["moves"], subset_mask=mask, autotune=kw.get("autotune", False),
                autotune_interval=16, counters=counters, normals_fn=normals_fn)
            log += lg
        want_dec = np.array([l["accepted"] for l in log])
        # a decision is only comparable when it is not a numerical coin toss: |log u - log ratio| well above
        # the energy tolerance (1e-5 sum|e| kJ/mol times beta)
        beta = 1.0 / (O.R_KJ * T)
        margins = np.array([abs(math.log(l["uniform"]) - l["log_ratio"]) if l["log_ratio"] < 0 else np.inf for l in log])
        assert margins.min() > 10 * beta * 1e-5 * eabs, "pick another seed: a trial sits on the acceptance threshold"
        assert np.array_equal(got_dec, want_dec), (np.nonzero(got_dec != want_dec)[0][:5], normals_fn)
        assert 0 < want_dec.sum() < total or kw.get("subset")
        assert move.statistics == dict(n_accepted=counters[0], n_proposed=counters[1]) and move.number_of_attemps_made == counters[2]
        assert tuple(st._current_PRNG_key) == tuple(key), "PRNG key chain differs"
        want_e = np.array([l["energy"] for l in log])
        assert np.all(np.abs(got_e - want_e) <= 1e-5 * eabs), np.abs(got_e - want_e).max()
        assert np.allclose(got_sigma, [l["sigma_d"] for l in log], rtol=1e-12)
        assert np.isclose(move.displacement_sigma.value_in_unit_system(u.md_unit_system), sig, rtol=1e-12)
        # list rebuilds happened in the same trials: same reference positions, same list
        if normals_fn is None:
            d = got_x - x
            d -= np.rint(d / np.diag(box)) * np.diag(box)  # a coordinate may wrap on one side only
            assert np.abs(d).max() <= 2e-6 * max(1, want_dec.sum()), np.abs(d).max()
            assert np.abs(nb.ref_positions.cpu().numpy() - lst.ref).max() <= 2e-6 * max(1, want_dec.sum()) or np.array_equal(
                nb.n_neighbors.cpu().numpy(), lst.n_nbr)
        else:
            assert np.array_equal(got_x, x), "final positions differ from the oracle chain on identical normals"
            assert np.array_equal(nb.ref_positions.cpu().numpy(), lst.ref)
            assert np.array_equal(nb.n_neighbors.cpu().numpy(), lst.n_nbr)
            if nb.n_max_neighbors == lst.n_max:
                assert np.array_equal(u32(nb.neighbor_list), lst.nl)
    if kw["skin"] == 0.05:
        assert lst.builds > 1, "this case is meant to exercise the rebuild branch"


def ca_reporter(ca):
    from chiron_amd.reporters import MCReporter

    return MCReporter("mc_test", 1)


def test_ideal_gas_barostat(ca):
    """Config C1 plumbing: NPT ideal gas, PE == 0, reduced potential == beta p V (test_mcmc.py:340-452)."""
    u = ca.unit
    from chiron_amd.topology import Topology

    positions = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1]], f32)
    box = np.diag([10.0, 10.0, 10.0]).astype(f32)
    pot = ca.potential.IdealGasPotential(Topology.homogeneous(8))
    ca.utils.PRNG.set_seed(1234)
    state = ca.states.SamplerState(nm(positions, ca), ca.utils.PRNG.get_random_key(), box_vectors=nm(box, ca))
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin, pressure=1.0 * u.atmosphere)
    rep = ca_reporter(ca)
    move = ca.mcmc.MonteCarloBarostatMove(volume_max_scale=0.1, number_of_moves=10, reporter=rep, report_interval=1)
    nbr = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=0 * u.nanometer)
    state, thermo, nbr = move.update(state, thermo, nbr)
    pe = rep.get_property("potential_energy")
    vol = rep.get_property("volume")
    assert pe[0] == 0 and pe[-1] == 0
    expect = thermo.pressure * thermo.beta * (vol[-1] * u.nanometer ** 3)
    assert np.isclose(thermo.get_reduced_potential(state), expect, rtol=1e-3)
    assert move.statistics["n_proposed"] == 10
    assert 5 <= move.statistics["n_accepted"] <= 10  # the reference's 8/10 depends on the JAX stream
    # positions scaled with the box
    assert np.isclose(float(state.box_vectors[0][0]) ** 3, vol[-1], rtol=1e-4)


def test_c1_ideal_gas_mcmc_workload(ca, O):
    """BASELINE config C1 at its own size: the schedule of Examples/Idealgas.py:16-104 (N = 216 ideal gas at
    298 K / 1 atm; per iteration 100 displacement moves sigma = 0.1 nm, then 10 barostat moves
    volume_max_scale = 0.2, both autotuned every 100 attempts; PairListNsqrd with cutoff 0) with the
    example's own acceptance test (:141-150: <V> within 5 %, sigma_V within 10 % of the ideal-gas
    expectation) after the example's 10 iterations and again after 100, and the whole chain -- every
    barostat decision, every reported volume, the autotuned scale, the final box and PRNG key -- against
    the oracle's restatement of the two moves (oracle.mc_displacement_update / mc_barostat_update)."""
    u = ca.unit
    n, T, P = 216, 298.0, 1.0
    gas = ca.testsystems.IdealGas(nparticles=n, temperature=T * u.kelvin, pressure=P * u.atmosphere)
    pot = ca.potential.IdealGasPotential(gas.topology)
    thermo = ca.states.ThermodynamicState(potential=pot, temperature=T * u.kelvin, pressure=P * u.atmosphere)
    ca.utils.PRNG.set_seed(1234)
    state = ca.states.SamplerState(positions=gas.positions, current_PRNG_key=ca.utils.PRNG.get_random_key(),
                                   box_vectors=gas.system.getDefaultPeriodicBoxVectors())
    key = tuple(state._current_PRNG_key)
    pairs = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=0.0 * u.nanometer)
    pairs.build_from_state(state)
    rep = ca_reporter(ca)
    baro = ca.mcmc.MonteCarloBarostatMove(volume_max_scale=0.2, number_of_moves=10, reporter=rep, autotune=True,
                                          autotune_interval=100)
    disp = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=0.1 * u.nanometer, number_of_moves=100, autotune=True,
                                              autotune_interval=100)
    sampler = ca.mcmc.MCMCSampler(ca.mcmc.MoveSchedule([("MonteCarloDisplacementMove", disp),
                                                        ("MonteCarloBarostatMove", baro)]))
    ideal_v = gas.get_volume_expectation(thermo).value_in_unit(u.nanometer ** 3)
    ideal_sd = gas.get_volume_standard_deviation(thermo).value_in_unit(u.nanometer ** 3)
    assert np.isclose(ideal_v, (n + 1) * O.KB * T / 101325.0 * 1e27, rtol=1e-9)

    # the oracle chain from the same start
    x = gas.positions.value_in_unit(u.nanometer).astype(f32)
    box = gas.box_vectors.value_in_unit(u.nanometer).astype(f32)
    zero = lambda *_: 0.0  # noqa: E731  (IdealGasPotential.compute_energy: 0 for any configuration)
    sig, vs, cd, cb, vols, dec = 0.1, 0.2, (0, 0, 0), (0, 0, 0), [], []

    st, th, nb = state, thermo, pairs
    done = 0
    for upto in (10, 100):
        st, th, nb = sampler.run(st, th, n_iterations=upto - done, nbr_list=nb)
        for _ in range(upto - done):
            x, _, key, sig, cd, _ = O.mc_displacement_update(x, box, None, key, SIGMA, EPS, T, sig, 100, pressure_atm=P,
                                                             autotune=True, autotune_interval=100, counters=cd, energy_fn=zero)
            x, box, key, vs, cb, lg = O.mc_barostat_update(x, box, key, T, P, vs, 10, zero, autotune=True,
                                                           autotune_interval=100, counters=cb)
            vols += [l["volume"] for l in lg]
            dec += [l["accepted"] for l in lg]
        done = upto
        volume = np.asarray(rep.get_property("volume"), dtype=np.float64)
        assert volume.shape == (10 * upto,)
        # Examples/Idealgas.py:141-150
        assert abs(ideal_v - volume.mean()) / ideal_v < 0.05, (upto, volume.mean(), ideal_v)
        assert abs(ideal_sd - volume.std()) / ideal_sd < 0.1, (upto, volume.std(), ideal_sd)
        # chain parity with the oracle
        assert np.array_equal(_decisions_from_reports(rep), np.array(dec))
        assert np.allclose(volume, vols, rtol=1e-12)
        assert baro.statistics == dict(n_accepted=cb[0], n_proposed=cb[1]) and baro.number_of_attemps_made == cb[2]
        assert disp.statistics == dict(n_accepted=cd[0], n_proposed=cd[1]) and cd[0] == cd[1] == 100 * upto  # U = 0: all accepted
        assert np.isclose(baro.volume_max_scale, vs, rtol=1e-12)
        assert np.isclose(disp.displacement_sigma.value_in_unit_system(u.md_unit_system), sig, rtol=1e-12)
        assert tuple(st._current_PRNG_key) == tuple(key)
        assert np.array_equal(st.box_vectors.cpu().numpy(), box)
        assert float(pot.compute_energy(st.positions, nb)) == 0.0
        got = st.positions.cpu().numpy()
        L = np.diag(box)
        assert np.all((got >= 0) & (got <= L))
        if upto == 10:
            # the displacement proposals themselves: device Box-Muller vs the oracle's libm one agree to ~2e-6 of
            # sigma_d per move, and coordinates near 20 nm carry an ulp of 2e-6.  (Not compared after 100
            # iterations: every ideal-gas displacement is accepted, so the reference's autotune multiplies sigma_d
            # by 1.1 per iteration -- 0.1 nm x 1.1^100 = 1.4 um -- and fp32 positions lose all absolute meaning.)
            d = got - x
            d -= np.rint(d / L) * L
            assert np.abs(d).max() < 2e-4, np.abs(d).max()

def _methane_system(ca, golden, methane_coords, nbr_kind):
    """Examples/LJ_MCMC.py:10-93: united-atom methane, 1100 beads, NPT state at 140 K / 13 atm."""
    u = ca.unit
    from chiron_amd.topology import Element, Topology

    g = golden["methane"]
    top = Topology()
    element = Element(1000, "CH4", "CH4", u.Quantity(16.04, u.gram / u.mole))
    chain = top.addChain()
    for _ in range(1100):
        residue = top.addResidue("CH4", chain)
        top.addAtom("CH4", element, residue)
    pot = ca.potential.LJPotential(top, sigma=g["sigma"] * u.nanometer,
                                   epsilon=g["epsilon_kcal_mol"] * u.kilocalories_per_mole,
                                   cutoff=g["cutoff"] * u.nanometer)
    thermo = ca.states.ThermodynamicState(potential=pot, temperature=140 * u.kelvin, pressure=13.00765 * u.atmosphere)
    L = g["box_edge"]
    ca.utils.PRNG.set_seed(1234)
    state = ca.states.SamplerState(positions=nm(methane_coords.astype(f32), ca), current_PRNG_key=ca.utils.PRNG.get_random_key(),
                                   box_vectors=nm(np.diag([L, L, L]).astype(f32), ca))
    space = ca.neighbors.OrthogonalPeriodicSpace()
    if nbr_kind == "pairlist":
        nbr = ca.neighbors.PairListNsqrd(space, cutoff=g["cutoff"] * u.nanometer)
    else:
        nbr = ca.neighbors.NeighborListCell(space, cutoff=g["cutoff"] * u.nanometer, skin=0.3 * u.nanometer, n_max_neighbors=250)
    nbr.build_from_state(state)
    return pot, thermo, state, nbr, g


def test_lj_barostat_proposal_matches_oracle(ca, O, golden, methane_coords):
    """One MonteCarloBarostatMove proposal on the LJ_MCMC system: scaled coordinates/box, forced list
    rebuild and log_proposal_ratio = -(u' - u) + N ln(V'/V) (chiron/mcmc.py:913-1009) vs the fp64 oracle."""
    u = ca.unit
    from chiron_amd.rng import split, uniform

    pot, thermo, state, nbr, g = _methane_system(ca, golden, methane_coords, "cell")
    move = ca.mcmc.MonteCarloBarostatMove(volume_max_scale=0.02, number_of_moves=1)
    key_before = state._current_PRNG_key
    u_cur = thermo.get_reduced_potential(state, nbr)
    prop, _, u_prop, lpr, nbr_prop = move._propose(state, thermo, u_cur, nbr)
    # the proposal the host logic must have made from the key chain
    _, sub = split(key_before)
    L = g["box_edge"]
    v0 = float(np.float32(L)) ** 3
    dv = (2.0 * uniform(sub) - 1.0) * 0.02 * v0
    scale = ((v0 + dv) / v0) ** (1.0 / 3.0)
    x0 = methane_coords.astype(f32)
    assert np.array_equal(prop.positions.cpu().numpy(), x0 * np.float32(scale))
    assert np.allclose(prop.box_vectors.cpu().numpy(), np.diag([L, L, L]).astype(f32) * np.float32(scale), rtol=1e-7)
    assert nbr_prop is not nbr and np.array_equal(nbr_prop.ref_positions.cpu().numpy(), prop.positions.cpu().numpy())
    # energies against the fp64 brute force at both volumes
    eps = g["epsilon_kcal_mol"] * 4.184
    e0 = O.lj_bruteforce_f64(x0, np.diag([L, L, L]).astype(f32), g["sigma"], eps, g["cutoff"])[0]
    bx1 = prop.box_vectors.cpu().numpy()
    e1 = O.lj_bruteforce_f64(prop.positions.cpu().numpy(), bx1, g["sigma"], eps, g["cutoff"])[0]
    v1 = float(np.float64(bx1[0, 0]) * np.float64(bx1[1, 1]) * np.float64(bx1[2, 2]))
    u0 = O.reduced_potential(e0, 140.0, 13.00765, v0)
    u1 = O.reduced_potential(e1, 140.0, 13.00765, v1)
    assert np.isclose(float(u_cur), u0, rtol=2e-5) and np.isclose(float(u_prop), u1, rtol=2e-5)
    expect = -(u1 - u0) + 1100 * np.log(v1 / v0)
    assert abs(float(lpr) - expect) < 2e-4 * max(1.0, abs(u0))


@pytest.mark.parametrize("nbr_kind", ["pairlist", "cell"])
def test_lj_mcmc_three_move_schedule(ca, O, golden, methane_coords, nbr_kind):
    """Examples/LJ_MCMC.py:95-170 (shortened): Langevin + displacement + barostat moves on NPT methane."""
    u = ca.unit
    from chiron_amd.reporters import LangevinDynamicsReporter, MCReporter

    pot, thermo, state, nbr, g = _methane_system(ca, golden, methane_coords, nbr_kind)
    rep_d, rep_b, rep_l = MCReporter("disp", 10), MCReporter("baro", 1), LangevinDynamicsReporter("lang", 10)
    disp = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=0.001 * u.nanometer, number_of_moves=30, reporter=rep_d,
                                              report_interval=10, autotune=True, autotune_interval=100)
    baro = ca.mcmc.MonteCarloBarostatMove(volume_max_scale=0.01, number_of_moves=6, reporter=rep_b, report_interval=1,
                                          autotune=True, autotune_interval=50)
    lang = ca.mcmc.LangevinDynamicsMove(timestep=1.0 * u.femtoseconds, collision_rate=1.0 / u.picoseconds,
                                        number_of_steps=40, reporter=rep_l, report_interval=10)
    sampler = ca.mcmc.MCMCSampler(ca.mcmc.MoveSchedule([("LangevinDynamicsMove", lang),
                                                        ("MonteCarloDisplacementMove", disp),
                                                        ("MonteCarloBarostatMove", baro)]))
    s2, t2, n2 = sampler.run(state, thermo, n_iterations=2, nbr_list=nbr)
    assert disp.statistics["n_proposed"] == 60 and baro.statistics["n_proposed"] == 12
    assert disp.statistics["n_accepted"] > 0
    vol = rep_b.get_property("volume")
    assert vol.shape == (12,) and np.all(np.isfinite(vol))
    bx = s2.box_vectors.cpu().numpy()
    assert np.isclose(bx[0, 0] * bx[1, 1] * bx[2, 2], vol[-1], rtol=1e-5)
    if baro.statistics["n_accepted"] > 0:
        assert not np.isclose(vol[-1], g["box_edge"] ** 3, rtol=1e-9)
    # the final configuration sits inside the final box and its energy is what the fp64 oracle says
    x = s2.positions.cpu().numpy()
    assert x.min() >= 0.0 and x.max() <= bx[0, 0] * (1 + 1e-6)
    e = float(pot.compute_energy(s2.positions, n2))
    e64, _, eabs, _ = O.lj_bruteforce_f64(x, bx, g["sigma"], g["epsilon_kcal_mol"] * 4.184, g["cutoff"])
    assert abs(e - e64) <= 1e-5 * eabs
    assert np.isclose(float(baro._current_reduced_potential), float(t2.get_reduced_potential(s2, n2)), rtol=1e-5)
    assert rep_l.get_property("potential_energy") is not None


def test_minimize_energy(ca, O, golden):
    """chiron/tests/test_minization.py: 0 iterations keep the energy, 100 iterations lower it on an
    LJ fluid (:30-80); two particles relax to the LJ minimum E = -1, r = 2^(1/6) (:83-128)."""
    u = ca.unit
    from chiron_amd.minimze import minimize_energy

    n = 600
    x, box = O.lj_lattice(n, 0.6, SIGMA, seed=3, jitter=0.25)
    pot = lj_pot(ca, n)
    nbr = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=SKIN * u.nanometer, n_max_neighbors=180)
    nbr.build(x, box)
    e0 = float(pot.compute_energy(x, nbr))
    r0 = minimize_energy(dev(x), pot.compute_energy, nbr, maxiter=0)
    assert np.isclose(float(pot.compute_energy(r0.params, nbr)), e0)
    r = minimize_energy(dev(x), pot.compute_energy, nbr, maxiter=100)
    e_min = float(pot.compute_energy(r.params, nbr))
    assert e_min < e0 and np.isfinite(e_min) and r.state.iter_num <= 100
    # two particles (sigma = 1 nm, eps = 1 kJ/mol, pair list, 10 nm box)
    g = golden["minimized_pair"]
    pot2 = ca.potential.LJPotential(None, sigma=1.0 * u.nanometer, epsilon=1.0 * u.kilojoules_per_mole, cutoff=3.0 * u.nanometer)
    coords = np.array([[0.0, 0.0, 0.0], [0.9, 0.0, 0.0]], f32)
    pl = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=3.0 * u.nanometer)
    pl.build(coords, np.diag([10.0, 10.0, 10.0]).astype(f32))
    res = minimize_energy(dev(coords), pot2.compute_energy, pl, maxiter=10_000)
    mx = res.params.cpu().numpy()
    assert np.isclose(float(pot2.compute_energy(res.params, pl)), g["energy"], atol=1e-3)
    assert np.isclose(np.linalg.norm(mx[1] - mx[0]), g["separation"], atol=1e-3)
    with pytest.raises(TypeError):
        minimize_energy(dev(coords), lambda x_: 0.0, None, maxiter=3)


def test_langevin_reference_literal_on_the_hip_integrator(ca, golden):
    """chiron/tests/test_mcmc.py:12-84 end to end on the product: a Harmonic oscillator propagated by
    LangevinIntegrator (BAOAB kernels, generic-potential branch) with the velocities and O-step
    noise of the reference's own JAX stream (restated in oracle/jax_prng.py) reports the five
    energies the reference asserts, to the reference's tolerance."""
    from test_oracle_golden import jax_stream_langevin_ho

    from chiron_amd.potential import HarmonicOscillatorPotential
    from chiron_amd.reporters import LangevinDynamicsReporter
    from chiron_amd.topology import Topology

    u = ca.unit
    g = golden["langevin_ho_jax_prng"]
    e_oracle, x0, v0, noise, k_md = jax_stream_langevin_ho(g)
    pot = HarmonicOscillatorPotential(Topology.homogeneous(1, "Ar"), g["k_kcal_mol_A2"] * u.kilocalories_per_mole / u.angstrom**2)
    assert pot.k == k_md
    thermo = ca.states.ThermodynamicState(potential=pot, temperature=g["temperature_K"] * u.kelvin)
    ca.utils.PRNG.set_seed(g["seed"])
    state = ca.states.SamplerState(positions=nm(x0, ca), current_PRNG_key=ca.utils.PRNG.get_random_key())
    state.velocities = v0
    reporter = LangevinDynamicsReporter()
    integ = ca.integrators.LangevinIntegrator(timestep=g["timestep_fs"] * u.femtosecond, reporter=reporter, report_interval=1,
                                              refresh_velocities=False)
    integ.run(state, thermo, number_of_steps=5, noise=noise)
    reporter.flush_buffer()
    energy = np.asarray(reporter.get_property("potential_energy"), dtype=np.float64).ravel()
    assert energy.shape == (5,)
    assert np.allclose(energy, np.array(g["potential_energy"]), rtol=g["rtol"], atol=1e-8)
    assert np.allclose(energy, e_oracle, rtol=2e-6)


def test_barostat_reference_literal_on_the_product_move(ca, golden, monkeypatch):
    """chiron/tests/test_mcmc.py:340-452 on the product's MonteCarloBarostatMove with its two random
    draws (key split, uniform) switched to the reference's JAX stream: 8 of 10 moves accepted, the
    same decisions as the oracle."""
    from test_oracle_golden import jax_stream_barostat_decisions

    from oracle import jax_prng as JX
    from chiron_amd import mcmc as mcmc_mod, states as states_mod
    from chiron_amd.rng import PRNGKey
    from chiron_amd.topology import Topology

    u = ca.unit
    g = golden["barostat_ideal_gas_jax_prng"]

    def jax_split(key):
        a, b = JX.split((key[0], key[1]))
        return PRNGKey(*a), PRNGKey(*b)

    monkeypatch.setattr(states_mod, "split", jax_split)
    monkeypatch.setattr(mcmc_mod, "host_uniform", lambda key, counter=0: float(JX.uniform((key[0], key[1]))))
    positions = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1]], f32)
    L = g["box_nm"]
    pot = ca.potential.IdealGasPotential(Topology.homogeneous(g["n_particles"]))
    _, first = JX.split(JX.PRNGKey(g["seed"]))  # PRNG.get_random_key() of the reference
    state = ca.states.SamplerState(nm(positions, ca), PRNGKey(*first), box_vectors=nm(np.diag([L, L, L]).astype(f32), ca))
    thermo = ca.states.ThermodynamicState(pot, temperature=g["temperature_K"] * u.kelvin, pressure=g["pressure_atm"] * u.atmosphere)
    rep = ca_reporter(ca)
    move = ca.mcmc.MonteCarloBarostatMove(volume_max_scale=g["volume_max_scale"], number_of_moves=g["n_proposed"],
                                          reporter=rep, report_interval=1)
    nbr = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=0 * u.nanometer)
    state, thermo, nbr = move.update(state, thermo, nbr)
    assert move.statistics == {"n_accepted": g["n_accepted"], "n_proposed": g["n_proposed"]}
    decisions, final_volume = jax_stream_barostat_decisions(g)
    vol = rep.get_property("volume")
    assert np.isclose(vol[-1], final_volume, rtol=1e-5)
    accepted = np.concatenate([[vol[0] != L ** 3], np.diff(vol) != 0])
    assert list(accepted) == decisions


def test_jax_normal_kernel_matches_restated_jax_stream(ca):
    """chx_jax_normal == oracle/jax_prng.normal (NumPy restatement pinned by the reference's literals) for even and
    odd element counts (jax pads the counter array to an even length and splits it in halves)."""
    from oracle import jax_prng as JX
    from chiron_amd.rng import JaxKey

    for key, shape in (((0, 1234), (1, 3)), ((603280156, 445306386), (7, 3)), ((5, 6), (500, 3)), ((9, 9), (64, 3))):
        got = ca.ops.jax_normal(JaxKey(*key), shape).cpu().numpy()
        want = JX.normal(key, shape)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 4e-6, (key, shape, np.abs(got - want).max())
    z = ca.ops.jax_normal(JaxKey(1, 2), (200000, 3)).cpu().numpy().astype(np.float64)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3
    with pytest.raises(ValueError):
        ca.ops.jax_normal(JaxKey(1, 2), (4, 3), out=torch.empty((5, 3), device="cuda"))


def test_reference_literals_through_the_public_api_with_the_threefry_stream(ca, golden):
    """SURVEY.md 8f-4 in the PRODUCT: with `PRNG.set_seed(1234, stream="threefry")` the two tests of the reference
    that depend on its JAX stream run as written in the reference -- no injected noise, no patched draws:
    chiron/tests/test_mcmc.py:12-84 (five Langevin energies of a harmonic oscillator, refresh_velocities=True) and
    :340-452 (8 of 10 barostat moves accepted)."""
    from chiron_amd.potential import HarmonicOscillatorPotential
    from chiron_amd.reporters import LangevinDynamicsReporter
    from chiron_amd.rng import JaxKey
    from chiron_amd.topology import Topology

    u = ca.unit
    try:
        g = golden["langevin_ho_jax_prng"]
        pot = HarmonicOscillatorPotential(Topology.homogeneous(1, "Ar"), g["k_kcal_mol_A2"] * u.kilocalories_per_mole / u.angstrom**2)
        thermo = ca.states.ThermodynamicState(potential=pot, temperature=g["temperature_K"] * u.kelvin)
        ca.utils.PRNG.set_seed(g["seed"], stream="threefry")
        state = ca.states.SamplerState(positions=nm(np.zeros((1, 3), f32), ca), current_PRNG_key=ca.utils.PRNG.get_random_key())
        assert isinstance(state.current_PRNG_key, JaxKey)
        reporter = LangevinDynamicsReporter()
        integ = ca.integrators.LangevinIntegrator(timestep=g["timestep_fs"] * u.femtosecond, reporter=reporter, report_interval=1,
                                                  refresh_velocities=True)
        out, _ = integ.run(state, thermo, number_of_steps=5)
        reporter.flush_buffer()
        energy = np.asarray(reporter.get_property("potential_energy"), dtype=np.float64).ravel()
        assert np.allclose(energy, np.array(g["potential_energy"]), rtol=g["rtol"], atol=1e-8), energy
        assert isinstance(out.current_PRNG_key, JaxKey)

        g = golden["barostat_ideal_gas_jax_prng"]
        positions = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1]], f32)
        L = g["box_nm"]
        pot = ca.potential.IdealGasPotential(Topology.homogeneous(g["n_particles"]))
        ca.utils.PRNG.set_seed(g["seed"], stream="threefry")
        state = ca.states.SamplerState(nm(positions, ca), ca.utils.PRNG.get_random_key(), box_vectors=nm(np.diag([L, L, L]).astype(f32), ca))
        thermo = ca.states.ThermodynamicState(pot, temperature=g["temperature_K"] * u.kelvin, pressure=g["pressure_atm"] * u.atmosphere)
        rep = ca_reporter(ca)
        move = ca.mcmc.MonteCarloBarostatMove(volume_max_scale=g["volume_max_scale"], number_of_moves=g["n_proposed"],
                                              reporter=rep, report_interval=1)
        nbr = ca.neighbors.PairListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=0 * u.nanometer)
        state, thermo, nbr = move.update(state, thermo, nbr)
        pe = rep.get_property("potential_energy")
        assert pe[0] == 0 and pe[-1] == 0
        assert move.statistics == {"n_accepted": g["n_accepted"], "n_proposed": g["n_proposed"]}  # 8 of 10
        # a displacement move on the same stream: normals from jrandom.normal(key, (N, 3)) (chiron/mcmc.py:731), host loop
        disp = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=0.1 * u.nanometer, number_of_moves=3)
        assert not disp._device_loop_applies(state, thermo, nbr)
        s2, _, _ = disp.update(state, thermo, nbr)
        assert isinstance(s2.current_PRNG_key, JaxKey) and disp.statistics["n_accepted"] == 3
    finally:
        ca.utils.PRNG.set_seed(1234)  # back to the product's default stream for the tests that follow


def test_metropolis_select_kernel(ca):
    from chiron_amd.rng import PRNGKey

    key = PRNGKey(5, 6)
    x_cur = dev(np.zeros((10, 3), f32))
    x_prop = dev(np.ones((10, 3), f32))
    acc = torch.zeros(1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
    u_cur = dev([5.0])
    u_prop = dev([4.0])
    ca.ops.metropolis_select(x_cur, x_prop, dev([1.0]), u_cur, u_prop, key, 0, acc, cnt)  # downhill: accept
    assert int(acc) == 1 and float(u_cur) == 4.0 and torch.equal(x_cur, x_prop)
    x_cur.zero_()
    ca.ops.metropolis_select(x_cur, x_prop, dev([-1.0]), u_cur, u_prop, key, 0, acc, cnt, uniform=0.9)
    assert int(acc) == 0 and float(x_cur.abs().sum()) == 0.0
    ca.ops.metropolis_select(x_cur, x_prop, dev([-1.0]), u_cur, u_prop, key, 0, acc, cnt, uniform=0.3)
    assert int(acc) == 1
    ca.ops.metropolis_select(x_cur, x_prop, dev([float("nan")]), u_cur, dev([float("nan")]), key, 0, acc, cnt)
    assert int(acc) == 0
    assert cnt.cpu().tolist() == [2, 4]
    # device uniform == host twin
    import chiron_amd.rng as R

    uu = R.uniform(key, 3)
    lp = float(np.log(uu))  # U < exp(lp) is false at equality, true just above
    ca.ops.metropolis_select(x_cur, x_prop, dev([lp * 0.999]), u_cur, u_prop, key, 3, acc, cnt)
    assert int(acc) == 1
    ca.ops.metropolis_select(x_cur, x_prop, dev([lp * 1.001]), u_cur, u_prop, key, 3, acc, cnt)
    assert int(acc) == 0


def test_cpu_tensors_are_refused(ca):
    from chiron_amd._lib import ChxError

    with pytest.raises(ChxError):
        ca.ops.wrap(torch.zeros(4, 3), ca.ops.make_box(np.eye(3, dtype=f32), True))


def test_direct_build_forgets_the_box_token_and_copies_the_box(ca, O):
    """(a) build() called directly says nothing about the sampler state the list will be used with: the
    "same box" token is dropped, so a later run() rebuilds from the STATE's box as the reference does
    (chiron/integrators.py:169-170) instead of speculating on the list's stale one.  (b) The list keeps its own
    copy of the box: mutating the caller's tensor in place (impossible with JAX arrays) cannot desynchronise
    box_vectors from the host-side copy the kernels use."""
    u = ca.unit
    n = 600
    x, box = O.lj_lattice(n, 0.3, SIGMA, seed=9)
    state = make_state(ca, x, box)
    nbr = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                        skin=SKIN * u.nanometer, n_max_neighbors=180)
    nbr.build_from_state(state)
    assert nbr._box_token is state._box_token
    other = (box * 1.25).astype(f32)
    nbr.build(nm((x * 1.25).astype(f32), ca), nm(other, ca))
    assert nbr._box_token is None
    assert np.allclose(nbr.box_vectors.cpu().numpy(), other)
    pot = lj_pot(ca, n)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    integ = ca.integrators.LangevinIntegrator(timestep=1.0 * u.femtoseconds, collision_rate=1.0 / u.picoseconds)
    out, nbr2 = integ.run(state, thermo, number_of_steps=3, nbr_list=nbr)
    assert np.allclose(nbr2.box_vectors.cpu().numpy(), box)  # rebuilt from the state's box
    L = float(box[0, 0])
    assert float(out.positions.max()) <= L * (1 + 1e-6)
    # (b)
    bt = torch.as_tensor(box).cuda()
    nbr3 = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=SKIN * u.nanometer, n_max_neighbors=180)
    nbr3.build(dev(x), bt)
    bt.mul_(2.0)
    assert np.allclose(nbr3.box_vectors.cpu().numpy(), box)
    assert nbr3._box_h.lx == f32(box[0, 0])
    nbr3.build(dev(x), bt)  # the same tensor object with new values: the list must notice
    assert np.allclose(nbr3.box_vectors.cpu().numpy(), 2 * box) and nbr3._box_h.lx == f32(2 * box[0, 0])


def test_mc_device_loop_yields_to_overridden_bookkeeping(ca, O):
    """A subclass that customises _report / _autotune / _update_statistics gets the host loop, which calls them
    (the device loop builds its reports itself)."""
    u = ca.unit
    n = 300
    x, box = O.lj_lattice(n, 0.1, SIGMA, seed=4)
    state = make_state(ca, x, box)
    thermo = ca.states.ThermodynamicState(lj_pot(ca, n), temperature=300 * u.kelvin)
    nbr = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                        skin=SKIN * u.nanometer, n_max_neighbors=180)
    nbr.build_from_state(state)
    base = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=0.01 * u.nanometer, number_of_moves=5)
    import os

    if os.environ.get("CHX_MC_HOST_LOOP") == "1":
        pytest.skip("the device loop is switched off (CHX_MC_HOST_LOOP=1)")
    assert base._device_loop_applies(state, thermo, nbr)
    seen = []

    class Custom(ca.mcmc.MonteCarloDisplacementMove):
        def _update_statistics(self, decision):
            seen.append(bool(decision))
            super()._update_statistics(decision)

    mv = Custom(displacement_sigma=0.01 * u.nanometer, number_of_moves=5)
    assert not mv._device_loop_applies(state, thermo, nbr)
    mv.update(state, thermo, nbr)
    assert len(seen) == 5 and mv.statistics["n_proposed"] == 5

    class Custom2(ca.mcmc.MonteCarloDisplacementMove):
        def _report(self, *a, **k):
            return super()._report(*a, **k)

    assert not Custom2(number_of_moves=1)._device_loop_applies(state, thermo, nbr)


def test_reference_mcmc_workflows_on_a_harmonic_oscillator(ca, O):
    """The MCMC-class workflows of chiron/tests/test_mcmc.py:87-286 on the product, with nbr_list=None (no wrap, no
    list: chiron/mcmc.py:756-770) and a generic (non-LJ) potential -- the host-shaped trial loop:
    (1) LangevinDynamicsMove inside MCMCSampler (:87-165); (2) MonteCarloDisplacementMove with atom_subset=[0] on
    one particle (:166-226); (3) the same move on a HarmonicOscillatorArray (:227-286).  (2) and (3) are followed
    trial by trial by the oracle's chain (same key splits, oracle normals, HO energy in NumPy)."""
    from chiron_amd.potential import HarmonicOscillatorPotential
    from chiron_amd.reporters import LangevinDynamicsReporter
    from chiron_amd.topology import Topology

    u = ca.unit
    k_q = 100.0 * u.kilocalories_per_mole / u.angstrom**2  # openmmtools HarmonicOscillator default K
    k_md = 100.0 * 4.184 / 0.01
    T = 300.0

    def ho_energy(x64):
        return float(f32(0.5 * k_md) * np.sum((x64 * x64).astype(f32), dtype=f32))

    # (1) Langevin dynamics as an MCMC move
    pot = HarmonicOscillatorPotential(Topology.homogeneous(1, "Ar"), k_q)
    thermo = ca.states.ThermodynamicState(pot, temperature=T * u.kelvin, volume=30 * u.angstrom**3)
    ca.utils.PRNG.set_seed(1234)
    state = ca.states.SamplerState(nm(np.zeros((1, 3), f32), ca), current_PRNG_key=ca.utils.PRNG.get_random_key())
    rep = LangevinDynamicsReporter()
    move = ca.mcmc.LangevinDynamicsMove(number_of_steps=10, refresh_velocities=True, reporter=rep, report_interval=1)
    sampler = ca.mcmc.MCMCSampler(ca.mcmc.MoveSchedule([("LangevinMove", move)]))
    s1, _, _ = sampler.run(state, thermo, n_iterations=2)
    rep.flush_buffer()
    assert np.asarray(rep.get_property("potential_energy")).shape[0] == 20 and np.isfinite(s1.positions.cpu().numpy()).all()

    # (2) + (3): displacement moves, chain vs oracle
    for n_part, subset in ((1, [0]), (5, None)):
        x0 = np.zeros((n_part, 3), f32)
        x0[:, 0] = np.arange(n_part, dtype=f32)  # HarmonicOscillatorArray: particles spaced along x, each bound to its site
        pot = HarmonicOscillatorPotential(Topology.homogeneous(n_part, "Ar"), k_q, x0=u.Quantity(x0.astype(np.float64), u.nanometer))
        thermo = ca.states.ThermodynamicState(pot, temperature=T * u.kelvin, volume=30 * u.angstrom**3)
        ca.utils.PRNG.set_seed(1234)
        state = ca.states.SamplerState(nm(x0, ca), current_PRNG_key=ca.utils.PRNG.get_random_key())
        key = tuple(state._current_PRNG_key)
        rep = ca_reporter(ca)
        mv = ca.mcmc.MonteCarloDisplacementMove(number_of_moves=10, displacement_sigma=0.1 * u.angstrom, atom_subset=subset,
                                                reporter=rep, report_interval=1)
        sampler = ca.mcmc.MCMCSampler(ca.mcmc.MoveSchedule([("MonteCarloDisplacementMove", mv)]))
        s2, _, n2 = sampler.run(state, thermo, n_iterations=2)
        assert n2 is None and mv.statistics["n_proposed"] == 20
        mask = None
        if subset is not None:
            mask = np.zeros(n_part, f32)
            mask[subset] = 1.0
        x, cnt, log = x0.copy(), (0, 0, 0), []
        for _ in range(2):
            x, _, key, _, cnt, lg = O.mc_displacement_update(x, None, None, key, 1.0, 1.0, T, 0.01, 10, subset_mask=mask,
                                                             counters=cnt, energy_fn=lambda xx, ll: ho_energy(xx - x0))
            log += lg
        assert np.array_equal(_decisions_from_reports(rep), np.array([l["accepted"] for l in log]))
        assert mv.statistics == dict(n_accepted=cnt[0], n_proposed=cnt[1])
        assert tuple(s2._current_PRNG_key) == tuple(key)
        got_e = np.asarray(rep.get_property("potential_energy"), dtype=np.float64)
        assert np.allclose(got_e, [l["energy"] for l in log], rtol=2e-4, atol=1e-6)
        assert np.abs(s2.positions.cpu().numpy() - x).max() < 2e-6 * 20


@pytest.mark.parametrize("window", [1, 16, 256])
@pytest.mark.parametrize("where", ["end-of-run", "end-of-chunk", "mid-chunk"])
def test_langevin_halt_at_chunk_and_run_boundaries_matches_oracle(ca, O, where, window):
    """The skin check firing in the LAST step of a run (the rebuild then happens after the loop's last A-O-A and
    before the trailing force + kick, chiron/integrators.py:188-195 with :239-242) and in the last step of a chunk:
    the cases in which the halt word reaches the host late -- with the trailing-kick launch or with the first
    launch of the next chunk (include/chx.h: chx_langevin_t::host_ctrl).  Rebuild count, the rebuilt list and the
    trajectory must be the oracle's in every case."""
    n, skin = 400, 0.02
    rng = np.random.default_rng(5)
    mass = np.full(n, 39.948, f32)
    kT = O.R_KJ * 300.0
    noise = rng.normal(size=(40, n, 3)).astype(f32)
    x, box = O.lj_lattice(n, 0.3, SIGMA, seed=43)
    v0 = O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))

    def oracle(steps):
        return O.langevin_run(x, v0, box, mass, SIGMA, EPS, CUTOFF, skin, 250, 0.004, 1.0, 300.0, steps, lambda s: noise[s])

    # first rebuild step of the reference trajectory: the smallest number of steps after which one rebuild has happened
    h = next(k for k in range(1, 40) if oracle(k)[2] >= 1)
    assert 2 <= h <= 30
    if where == "end-of-run":
        steps, chunk = h, 7 if h % 7 else 5   # the halt comes out of the run's last step, not at a chunk end
    elif where == "end-of-chunk":
        steps, chunk = h + 9, h               # ... out of the last launch of the first chunk
    else:
        steps, chunk = h + 9, h + 3           # (control: in the middle of a chunk)
    _, _, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0, "NeighborListNsqrd", 250)
    integ.chunk_size = chunk
    integ.launch_window = window  # launches kept queued ahead of the device (chx_langevin_fused_run); 256: the whole run
    state.velocities = v0
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr, noise=noise[:steps])
    xo, vo, rebuilds, (nl_mask, nl, n_max) = oracle(steps)
    assert rebuilds >= 1 and integ.n_rebuilds == rebuilds
    assert np.abs(out.positions.cpu().numpy() - xo).max() < 1e-5
    assert np.abs(out.velocities.cpu().numpy() - vo).max() < 5e-4 * max(1.0, np.abs(vo).max())
    assert nbr.n_max_neighbors == n_max and np.array_equal(u32(nbr.neighbor_list), nl)
    if where == "end-of-run":
        # rebuilt from the final positions: the list's reference positions ARE the returned positions
        assert np.array_equal(nbr.ref_positions.cpu().numpy(), out.positions.cpu().numpy())


@pytest.mark.parametrize("reports", [False, True], ids=["one-segment", "report-segments"])
def test_langevin_launch_window_does_not_change_the_trajectory(ca, O, reports):
    """The bounded launch queue (LangevinIntegrator.launch_window -> chx_langevin_fused_run) only changes how far the
    host runs ahead of the device: trajectory, rebuild count, pruning counters and the rebuilt list are the same
    bits for every window (256: everything is enqueued at once), with the Philox stream and
    with the run cut into report segments.  Thin skin + large time step: a rebuild every few steps."""
    n, skin, steps = 2500, 0.05, 150
    ref = None
    for window, chunk in ((256, 32), (1, 32), (2, 5), (16, 32), (64, 7), (16, 1)):
        x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
        integ.chunk_size, integ.launch_window = chunk, window
        if reports:
            integ.report_interval, integ.save_traj_in_memory = 11, True
        ca.utils.PRNG.set_seed(77)
        state.current_PRNG_key = ca.utils.PRNG.get_random_key()
        out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
        got = (out.positions.clone(), out.velocities.clone(), integ.n_rebuilds, integ.prune_passes, integ.full_list_launches,
               nbr.neighbor_list.clone(), [t.clone() for t in integ.traj])
        assert integ.n_rebuilds >= 5
        if ref is None:
            ref = got
            continue
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (window, chunk)
        assert got[2:5] == ref[2:5], (window, chunk, got[2:5], ref[2:5])
        assert torch.equal(got[5], ref[5])
        assert len(got[6]) == len(ref[6]) and all(torch.equal(a, b) for a, b in zip(got[6], ref[6]))


def test_both_cell_build_kernels_write_the_same_bits(tmp_path):
    """The one-pass cell build has two kernels (csrc/chx_cell_fl.hip): one wavefront per particle and one per cell
    (the cell's particles share the scan of their 27-cell neighbourhood; chosen when there are enough cells to fill
    the chip).  Padded list, mask, counts, the grown n_max_neighbors, force-list rows, degrees and the internal order
    must be identical bits -- dilute and dense cells, rows longer than the initial capacity, three cells per axis."""
    import os
    import subprocess
    import sys

    got = {}
    # (round 4: the per-particle kernel reads per-cell candidate index lists written by the binning -- chx_cell.h --
    # unless CHX_BUILD_LISTS=0, or a cell's neighbourhood is longer than a list holds: the "dense" system; a third dump
    # runs it on the older range search)
    # (round 5: the integrator's rebuilds -- half rows counted only -- take sixteen lanes per row, four rows per wavefront,
    # where the per-particle kernel would run with candidate lists; a fourth dump switches that off: CHX_BUILD_SUB=0)
    for kern, lists, sub in (("particle", "1", "1"), ("cell", "1", "1"), ("particle", "0", "1"), ("particle", "1", "0")):
        path = tmp_path / f"{kern}{lists}{sub}.npz"
        env = dict(os.environ, CHX_BUILD_KERNEL=kern, CHX_BUILD_LISTS=lists, CHX_BUILD_SUB=sub)
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_dump_cell_build.py"), str(path)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[kern + lists + ("" if sub == "1" else "_nosub")] = np.load(path)
    a = got["particle1"]
    assert len(a.files) >= 80
    if "CHX_FL_G" not in os.environ:  # (16 or 64 lanes per particle: a single chunk holds these rows)
        assert int(a["dilute_run_status_tight"][4]) > 0  # (rows beyond a one-chunk force list: counted)
    for other in ("cell1", "particle0", "particle1_nosub"):
        b = got[other]
        assert sorted(a.files) == sorted(b.files)
        for key in a.files:
            assert np.array_equal(a[key], b[key]), (other, key)
    assert int(a["grow_m"][0]) > 12  # the growth loop really ran (rows longer than the initial capacity)


def test_build_riders_and_split_entry_points_write_the_same_bits(ca, O):
    """Round 4's rebuild chain at the C ABI (include/chx.h): (a) chx_cell_bin_fl + chx_nl_rows_cell_fl == the one call
    chx_nl_build_cell_fl; (b) the riders -- a copy of the positions, the Langevin state packed into the new internal
    order, the control-block reset, the status block published to host-mapped memory -- write exactly what the separate
    launches they replace write (clone, chx_pack_langevin with its ctrl rider, chx_publish)."""
    import ctypes as C

    from chiron_amd import _lib, ops

    lib = _lib.load()
    n, m = 5000, 180
    x, box = O.lj_lattice(n, 0.1, SIGMA, seed=5)
    rng = np.random.default_rng(1)
    x = O.periodic_wrap((x + rng.normal(0, 0.05, x.shape)).astype(f32), box)
    xt, vt = dev(x), dev(rng.normal(size=(n, 3)))
    mass = dev(rng.uniform(10.0, 50.0, size=n))
    ref = dev(O.periodic_wrap((x + rng.normal(0, 0.01, x.shape)).astype(f32), box))
    bh = ops.make_box(dev(box), True)
    cs = CUTOFF + SKIN
    ncell, nc = ops.cell_grid(bh, cs, n)
    assert ncell > 0
    g, chunks = 4, 6
    plain = ops.nl_build_cell_fl(xt, bh, cs, m, ncell, nc, g, chunks, {})
    want_pack = ops.pack_langevin(xt, vt, mass, ref, plain[3], True)
    for ref3 in (ref, None):
        f32k = dict(dtype=torch.float32, device="cuda")
        outs = (torch.zeros((n, 4), **f32k), torch.zeros((n, 3), **f32k), torch.zeros((n,), **f32k), torch.zeros((n, 4), **f32k))
        ctrl = torch.full((_lib.CTRL_WORDS,), 77, dtype=torch.int32, device="cuda")
        host = torch.full((8,), -1, dtype=torch.int32).pin_memory()
        copy = torch.zeros_like(xt)
        riders = {"copy": copy, "pack": (vt, mass, ref3) + outs, "ctrl": ctrl, "ctrl_keep": 1, "publish": host}
        got = ops.nl_build_cell_fl(xt, bh, cs, m, ncell, nc, g, chunks, {}, riders=riders)
        torch.cuda.synchronize()
        # rows (real entries), first, counts, order, force list (the chunks a row may touch), degrees
        real = torch.arange(m, device="cuda")[None, :] < plain[2][:, None]
        used = torch.clamp((plain[5] + 4 * g - 1) // (4 * g), min=1)
        keep = torch.arange(chunks, device="cuda")[:, None] < used[None, :]
        assert torch.equal(plain[0][real], got[0][real]) and torch.equal(plain[4][keep], got[4][keep])
        for k in (1, 2, 3, 5):
            assert torch.equal(plain[k], got[k]), k
        assert torch.equal(copy, xt)
        st = got[6].cpu().numpy()
        assert np.array_equal(host.numpy()[:7], st[:7]) and host.numpy()[7] == 0 and st[0] == 0 and st[1] == int(plain[2].max())
        want = want_pack if ref3 is not None else ops.pack_langevin(xt, vt, mass, xt, plain[3], True)
        for a, b in zip(want, outs):
            assert torch.equal(a, b)
        reset = [0, 0, 0, 2**31 - 1, -2, -2, 0, 0, 77, 77, -1, -1, 2**31 - 1, 0, 0, 0]  # ctrl_keep bit 0: words 8, 9 stay
        assert ctrl.cpu().tolist() == reset
    # the one-call form, straight through ctypes, against the two halves ops.nl_build_cell_fl makes
    i32 = dict(dtype=torch.int32, device="cuda")
    arrs = dict(cell_of=torch.empty((n,), **i32), cell_count=torch.zeros((ncell + 1,), **i32), cell_start=torch.empty((ncell + 1,), **i32),
                ids=torch.empty((n,), **i32), status=torch.empty((8,), **i32), nl=torch.empty((n, m), **i32), first=torch.empty((n,), **i32),
                n_nbr=torch.empty((n,), **i32), fl=torch.empty((chunks, n, 4 * g), **i32), deg=torch.empty((n,), **i32),
                xs4=torch.empty((n, 4), dtype=torch.float32, device="cuda"))
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.chx_nl_build_cell_fl(p(xt), n, m, bh, C.c_float(cs), nc, p(arrs["cell_of"]), p(arrs["cell_count"]), p(arrs["cell_start"]),
                                  p(arrs["ids"]), p(arrs["status"]), p(arrs["nl"]), p(arrs["first"]), p(arrs["n_nbr"]), g, chunks,
                                  p(arrs["fl"]), p(arrs["deg"]), p(arrs["xs4"]), None, None)
    assert rc == 0, lib.chx_last_error()
    torch.cuda.synchronize()
    assert torch.equal(arrs["ids"], plain[3]) and torch.equal(arrs["n_nbr"], plain[2]) and torch.equal(arrs["deg"], plain[5])
    assert torch.equal(arrs["first"], plain[1]) and int(arrs["cell_count"].abs().sum()) == 0  # (left zeroed for the next build)
    real = torch.arange(m, device="cuda")[None, :] < plain[2][:, None]
    assert torch.equal(arrs["nl"][real], plain[0][real])


def test_build_without_half_rows_counts_them_and_writes_the_same_force_list(ca, O):
    """include/chx.h (round 4): nl_rows_out == first_out == NULL -- the rebuilds inside LangevinIntegrator.run() -- counts
    the half rows (n_nbr, status[1], [2]: the reference's capacity checks, chiron/neighbors.py:709-727) without writing
    them; force list, degrees, order and status must be the bits of the full build.  Both row kernels (one wavefront
    per particle below 8192 cells, one per cell above), and a capacity smaller than the rows (counted all the same)."""
    from chiron_amd import ops

    for n, rho, m in ((5000, 0.1, 180), (6000, 0.8, 40), (262144 // 2, 0.05, 180)):
        x, box = O.lj_lattice(n, rho, SIGMA, seed=5)
        x = O.periodic_wrap((x + np.random.default_rng(1).normal(0, 0.05, x.shape)).astype(f32), box)
        xt = dev(x)
        bh = ops.make_box(dev(box), True)
        cs = CUTOFF + SKIN
        ncell, nc = ops.cell_grid(bh, cs, n)
        assert ncell > 0 and (n < 100000 or ncell >= 8192)  # (the large system takes the cell-wise kernel)
        rho_n = n / float(np.prod(np.diag(box)))
        g = 4
        chunks = ops.fl_chunks_for(int(1.8 * rho_n * 4.19 * cs ** 3) + 24, g)
        sc_a, sc_b = {}, {}
        full = ops.nl_build_cell_fl(xt, bh, cs, m, ncell, nc, g, chunks, sc_a)
        host = torch.full((8,), -1, dtype=torch.int32).pin_memory()
        part = ops.nl_build_cell_fl(xt, bh, cs, m, ncell, nc, g, chunks, sc_b, riders={"publish": host}, half_rows=False)
        torch.cuda.synchronize()
        assert part[0] is None and part[1] is None
        for k in (2, 3, 5):
            assert torch.equal(full[k], part[k]), (n, k)
        used = torch.clamp((full[5] + 4 * g - 1) // (4 * g), min=1)
        keep = torch.arange(chunks, device="cuda")[:, None] < used[None, :]
        assert torch.equal(full[4][keep], part[4][keep])
        sa, sb = full[6].cpu().numpy(), part[6].cpu().numpy()
        assert np.array_equal(sa[:5], sb[:5]) and np.array_equal(host.numpy()[:5], sb[:5]) and sa[0] == 0 and sa[4] == 0
        assert sa[1] == int(full[2].max()) and sa[2] == int((full[2] == m).sum())
        if m == 40:
            assert sa[1] > m  # rows longer than the capacity: counted all the same


def test_half_rows_left_out_by_a_run_are_built_when_read(ca, O):
    """LangevinIntegrator.run() rebuilds write the force list and count the half rows (integrators._RUN_HALF_ROWS);
    `neighbor_list` / `neighbor_mask` read afterwards must be what NeighborListNsqrd.build() gives from the list's
    reference positions (chiron/integrators.py:241-242: the list the caller gets back is the last rebuild's), n_builds
    must not move when they are read, and a deep copy taken before the read must give the same."""
    import copy

    n, skin, steps = 2500, 0.05, 120
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    integ.list_carry = False
    state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    # (CHX_RUN_HALF_ROWS=1 / CHX_SYNC_REBUILD=1: every build writes the rows; the reads below must not care)
    assert ca.integrators._RUN_HALF_ROWS or not ca.integrators._SPECULATIVE_REBUILD or nbr.__dict__.get("_half_rows_pending")
    builds = nbr.n_builds
    assert builds >= 2  # (rebuilt inside the run)
    twin = copy.deepcopy(nbr)
    ref_pos = nbr.ref_positions.clone()
    counts = nbr.n_neighbors.clone()
    fl_before = nbr.force_list
    deg_before, perm_before = fl_before.deg.clone(), fl_before.perm.clone()
    got_nl, got_mask = nbr.neighbor_list.clone(), nbr.neighbor_mask.clone()
    assert nbr.n_builds == builds and not nbr.__dict__.get("_half_rows_pending")
    assert torch.equal(nbr.ref_positions, ref_pos) and torch.equal(nbr.n_neighbors, counts)
    assert torch.equal(nbr.force_list.deg, deg_before) and torch.equal(nbr.force_list.perm, perm_before)
    fresh = ca.neighbors.NeighborListNsqrd(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=nbr.cutoff, skin=nbr.skin,
                                           n_max_neighbors=nbr.n_max_neighbors)
    fresh.build(ref_pos, nbr.box_vectors)
    assert torch.equal(fresh.neighbor_list, got_nl) and torch.equal(fresh.neighbor_mask, got_mask)
    assert torch.equal(fresh.n_neighbors, counts)
    assert torch.equal(twin.neighbor_list, got_nl) and torch.equal(twin.n_neighbors, counts)
    # and the trajectory does not care whether somebody looked: the next run from here equals the one of an untouched twin
    x2, box2, pot2, state2, thermo2, nbr2, integ2 = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    integ2.list_carry = False
    state2, nbr2 = integ2.run(state2, thermo2, number_of_steps=steps, nbr_list=nbr2)
    a, nbr = integ.run(state, thermo, number_of_steps=40, nbr_list=nbr)
    b, nbr2 = integ2.run(state2, thermo2, number_of_steps=40, nbr_list=nbr2)
    assert torch.equal(a.positions, b.positions) and torch.equal(a.velocities, b.velocities)


def test_cell_slot_overflow_voids_the_build_and_the_retry_goes_without_slots(ca, O, monkeypatch):
    """include/chx.h, chx_build_riders_t::cell_slots (round 4: the binning keeps every particle's id in a per-cell slot
    array and needs no scatter launch).  A cell with more particles than slots sets status[5]; the synchronous build
    repeats itself without slots, the integrator's speculative rebuild is rejected and redone -- same list, same
    trajectory as with the slots off."""
    u = ca.unit
    n = 6000
    x, box = O.lj_lattice(n, 0.8, SIGMA, seed=5)
    x = O.periodic_wrap((x + np.random.default_rng(1).normal(0, 0.03, x.shape)).astype(f32), box)

    def build(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        nbr = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                            skin=SKIN * u.nanometer, n_max_neighbors=180)
        nbr.build(dev(x), dev(box))
        for k in env:
            monkeypatch.delenv(k)
        return nbr

    plain = build(CHX_BUILD_SLOTS="0")
    assert plain._scratch.get("slots") is None
    roomy = build()
    assert roomy._scratch["slots"] is not None and not roomy._scratch.get("no_slots")
    tight = build(CHX_BUILD_SLOT_CAP="8")
    assert tight._scratch.get("no_slots") and tight._scratch["slots"] is None
    for other in (roomy, tight):
        assert torch.equal(plain.neighbor_list, other.neighbor_list) and torch.equal(plain.n_neighbors, other.n_neighbors)
        assert torch.equal(plain.force_list.perm, other.force_list.perm) and torch.equal(plain.force_list.deg, other.force_list.deg)
    # the integrator's speculative rebuilds: slots too small from the second build on
    res = []
    for shrink in (False, True):
        xs, boxs, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, 2500, 0.3, 43, 0.05, 4.0)
        integ.list_carry = False
        state, nbr = integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
        if shrink:
            ncell = nbr._scratch["slots"][0].shape[0]
            nbr._scratch["slots"] = (torch.empty((ncell, 1), dtype=torch.int32, device="cuda"), 1)
        state, nbr = integ.run(state, thermo, number_of_steps=90, nbr_list=nbr)
        assert bool(nbr._scratch.get("no_slots")) == shrink
        res.append((state, nbr, integ.n_rebuilds))
    (a, na, ra), (b, nb, rb) = res
    assert ra == rb >= 1
    assert torch.equal(a.positions, b.positions) and torch.equal(a.velocities, b.velocities)
    assert torch.equal(na.neighbor_list, nb.neighbor_list)


def test_langevin_run_after_an_interrupted_run(ca, O, monkeypatch):
    """A run() that dies half-way (here: its first rebuild raises) leaves its generation of step launches open --
    the host-mapped control slot, launches possibly still queued.  The integrator must come back clean: the next
    run() on the same object gives the bits of a fresh integrator, also after enough runs to cycle through the whole
    ring of control slots."""
    n, skin, steps = 2500, 0.05, 60
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    rs = ca.integrators._RunState
    orig = rs._rebuild
    calls = {"n": 0}

    def boom(self, step=0, speculative=False):
        calls["n"] += 1
        if calls["n"] == 1:
            raise RuntimeError("interrupted")
        return orig(self, step, speculative)

    monkeypatch.setattr(rs, "_rebuild", boom)
    ca.utils.PRNG.set_seed(5)
    key = ca.utils.PRNG.get_random_key()
    rng = np.random.default_rng(11)
    v0 = O.initial_velocities(np.full(n, 39.948, f32), O.R_KJ * 300.0, rng.normal(size=(n, 3)).astype(f32))
    state.velocities = v0
    state.current_PRNG_key = key  # (run() draws its key from the state and advances it: reset before every run)
    with pytest.raises(RuntimeError, match="interrupted"):
        integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    monkeypatch.setattr(rs, "_rebuild", orig)
    outs = []
    for _ in range(6):  # more runs than control slots in the ring
        state.current_PRNG_key = key
        out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
        outs.append(out)
    x2, box2, pot2, state2, thermo2, nbr2, integ2 = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    state2.velocities = v0
    state2.current_PRNG_key = key
    ref, _ = integ2.run(state2, thermo2, number_of_steps=steps, nbr_list=nbr2)
    assert integ.n_rebuilds == integ2.n_rebuilds >= 2
    for out in outs:  # every run starts from the same state and key: same trajectory every time
        assert torch.equal(out.positions, ref.positions) and torch.equal(out.velocities, ref.velocities)


def test_deep_copied_integrator_and_sampler_run_on_fresh_slots(ca, O):
    """Advisor finding (round 2): MultiStateSampler deep-copies its MCMC sampler (chiron/multistate.py:266-268), and a
    LangevinIntegrator that has run owns host-mapped pinned slots the step kernel stores into.  A copy must allocate
    its own: run, deep-copy the move and the integrator, run the copies -- same trajectory as the original's second
    run, halts seen (rebuilds happen), nothing shared."""
    import copy

    n = 3000
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.1, 77, 0.05, 2.0)
    out1, nbr1 = integ.run(state, thermo, number_of_steps=150, nbr_list=nbr)
    import os

    windowed = True
    ring_name = "_gen_ring"
    assert integ.n_rebuilds >= 1 and ring_name in integ.__dict__  # the loop ran and left its host-mapped slots
    clone = copy.deepcopy(integ)
    assert ring_name not in clone.__dict__ and "_dev_scratch" not in clone.__dict__
    a, nbr_a = integ.run(copy.deepcopy(out1), thermo, number_of_steps=150, nbr_list=copy.deepcopy(nbr1))
    b, nbr_b = clone.run(copy.deepcopy(out1), thermo, number_of_steps=150, nbr_list=copy.deepcopy(nbr1))
    assert clone.n_rebuilds == integ.n_rebuilds >= 1
    assert torch.equal(a.positions, b.positions) and torch.equal(a.velocities, b.velocities)
    ring_a, ring_b = integ.__dict__[ring_name], clone.__dict__[ring_name]
    tensors = (lambda ring: [s.tensor for s in ring]) if windowed else (lambda ring: list(ring))
    assert all(ta.data_ptr() != tb.data_ptr() for ta in tensors(ring_a) for tb in tensors(ring_b))
    assert all(t.is_pinned() for t in tensors(ring_b))
    if windowed:
        assert all(s.host.ctypes.data == s.tensor.data_ptr() for s in ring_b)
    # the same through a move inside a sampler, as MultiStateSampler holds it
    move = ca.mcmc.LangevinDynamicsMove(timestep=2.0 * ca.unit.femtoseconds, number_of_steps=120)
    sampler = ca.mcmc.MCMCSampler(ca.mcmc.MoveSchedule([("langevin", move)]))
    s1, t1, n1 = sampler.run(copy.deepcopy(out1), thermo, n_iterations=1, nbr_list=copy.deepcopy(nbr1))
    sampler2 = copy.deepcopy(sampler)
    s2, _, _ = sampler2.run(copy.deepcopy(s1), thermo, n_iterations=1, nbr_list=copy.deepcopy(n1))
    s3, _, _ = sampler.run(copy.deepcopy(s1), thermo, n_iterations=1, nbr_list=copy.deepcopy(n1))
    assert torch.isfinite(s2.positions).all() and torch.equal(s2.positions, s3.positions)


# ---------------------------------------------------------------------------
# list carry-over between run() calls
# ---------------------------------------------------------------------------
def _carry_segments(ca, O, carry, segments, steps, n=4096, dt_fs=2.0, seed=67, tamper=None, max_steps=None):
    """`segments` consecutive run(steps) calls on one integrator and one list, each continuing from the state the
    previous one returned.  Returns per-segment records and the final objects."""
    import os

    from chiron_amd import integrators

    u = ca.unit
    x, box = O.lj_lattice(n, 0.1, SIGMA, seed=seed)
    pot = lj_pot(ca, n)
    state = make_state(ca, x, box)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    nbr = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                        skin=SKIN * u.nanometer, n_max_neighbors=180)
    integ = ca.integrators.LangevinIntegrator(timestep=dt_fs * u.femtoseconds, collision_rate=1.0 / u.picoseconds,
                                              list_carry=carry, **({} if max_steps is None else {"list_carry_max_steps": max_steps}))
    recs = []
    if True:
        for k in range(segments):
            if tamper is not None:
                tamper(k, state, integ, nbr)
            key_before = state.current_PRNG_key
            x_in, v_in = state.positions.clone(), None if state._velocities is None else state.velocities.clone()
            state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
            recs.append(dict(key=key_before, x_in=x_in, v_in=v_in, rebuild_steps=list(integ.rebuild_steps),
                             n_rebuilds=integ.n_rebuilds, x=state.positions.clone(), v=state.velocities.clone()))
    return x, box, pot, state, nbr, integ, recs


def _mi_max(a, b, L):
    d = (a - b).abs()
    return torch.minimum(d, (d - L).abs()).max().item()


def test_list_carry_over_keeps_the_reference_events_and_the_public_list(ca, O):
    """A run() that continues the trajectory of the previous one starts on the force list that one ended on: no list
    build on entry.  What the reference's run-start build means to the steps -- reference positions := x0
    (chiron/integrators.py:169, chiron/neighbors.py:671) -- still happens, so every segment has the reference's
    rebuild events at the reference's steps, the trajectory is the halting scheme's up to summation order, and the
    caller's list, resolved when somebody reads it, is the one the reference would hold: built from the positions of
    the last rebuild (or x0).  Checked against the same segments run with a list build on every entry, and against
    the C oracle on the same Philox streams."""
    from oracle import c_oracle as CO
    import chiron_amd.rng as R

    segments, steps, n = 10, 60, 4096
    x, box, pot, a, nbr_a, integ_a, ra = _carry_segments(ca, O, True, segments, steps)
    _, _, _, b, nbr_b, integ_b, rb = _carry_segments(ca, O, False, segments, steps)
    L = float(box[0, 0])
    assert getattr(integ_a, "n_carried_runs", 0) == segments - 1 and getattr(integ_a, "n_carry_rejected", 0) == 0
    assert getattr(integ_b, "n_carried_runs", 0) == 0
    # the force list ran out of its own guarantee at least once along the way and was renewed without a reference event
    assert getattr(integ_a, "n_internal_halts", 0) >= 1
    for k in range(segments):
        assert ra[k]["rebuild_steps"] == rb[k]["rebuild_steps"] and ra[k]["n_rebuilds"] == rb[k]["n_rebuilds"], k
        assert _mi_max(ra[k]["x"], rb[k]["x"], L) < 2e-4 and (ra[k]["v"] - rb[k]["v"]).abs().max().item() < 2e-2, k
    # the caller's list: pending after a carried run, counted as the reference counts its builds
    assert "_deferred" in nbr_a.__dict__ and "n_neighbors" not in nbr_a.__dict__
    assert nbr_a.is_built and nbr_a.n_builds == nbr_b.n_builds
    ref = nbr_a.ref_positions  # (first access: builds)
    assert "_deferred" not in nbr_a.__dict__
    assert _mi_max(ref, nbr_b.ref_positions, L) < 2e-4
    if not ra[-1]["rebuild_steps"]:  # no event in the last segment: the list is the run-start build's, from x0
        assert torch.equal(ref, ra[-1]["x_in"])
    mask, nl, nnbr = CO.nl_build_fixed(ref.cpu().numpy(), box, CUTOFF + SKIN, int(nbr_a.n_max_neighbors))
    assert np.array_equal(nbr_a.n_neighbors.cpu().numpy(), nnbr) and np.array_equal(u32(nbr_a.neighbor_list), nl)
    e_list = float(pot.compute_energy(a.positions, nbr_a))
    e64 = CO.lj_cell_f64(a.positions.cpu().numpy(), box, SIGMA, EPS, CUTOFF)[0]
    assert abs(e_list - e64) <= 2e-5 * abs(e64) + 1e-3
    # the C oracle, segment by segment: a list build (reference positions := x0) on every entry, the segment's own key
    mass = np.full(n, 39.948, f32)
    kT, ca_, cb_ = O.langevin_constants(300.0, 0.002, 1.0)
    _, run_key = R.split(ra[0]["key"])
    xo = x
    vo = O.initial_velocities(mass, kT, O.philox_normals3(n, 0, O.STREAM_VELOCITY, tuple(run_key)))
    for k in range(segments):
        _, run_key = R.split(ra[k]["key"])
        xo, vo, st = CO.langevin_run_cell_omp(xo, vo, box, mass, SIGMA, EPS, CUTOFF, SKIN, 0.002, ca_, cb_, kT, steps, None,
                                              tuple(run_key), threads=CO.effective_cores())
        assert st["rebuilds"] == ra[k]["n_rebuilds"], k
        do = np.abs(ra[k]["x"].cpu().numpy() - xo)
        assert np.minimum(do, np.abs(do - L)).max() < 5e-4, k
        # (each segment starts from the product's own state: the comparison does not accumulate)
        xo, vo = ra[k]["x"].cpu().numpy(), ra[k]["v"].cpu().numpy()


def test_list_carry_over_is_declined_for_positions_from_elsewhere(ca, O):
    """The offer is taken only by a run() that gets the very tensor the previous one returned, unwritten, with the same
    list object; anything else builds its list on entry as the reference does."""
    def tamper(k, state, integ, nbr):
        if k == 1:  # a copy of the positions (what a Monte Carlo move or another replica hands over)
            state.positions = state.positions.clone()
        if k == 2:  # written in place since
            state.positions.add_(0.0)
        if k == 3:  # the offer is for ONE next run: consumed or dropped, never kept
            assert integ.__dict__.get("_carry") is not None
    x, box, pot, a, nbr, integ, recs = _carry_segments(ca, O, True, 5, 20, n=2048, tamper=tamper)
    assert integ.n_carried_runs == 2  # segments 3 and 4
    assert "_deferred" in nbr.__dict__
    # a longer run than the carry-over serves takes the ordinary path, and finds the caller's list current afterwards
    u = ca.unit
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    from chiron_amd import integrators

    out, nbr = integ.run(a, thermo, number_of_steps=integ.list_carry_max_steps + 1, nbr_list=nbr)
    assert integ.n_carried_runs == 2 and "_deferred" not in nbr.__dict__ and "n_neighbors" in nbr.__dict__
    # another list object: declined as well
    nbr2 = ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=SKIN * u.nanometer, n_max_neighbors=180)
    nbr2.build_from_state(out)
    out2, nbr2 = integ.run(out, thermo, number_of_steps=20, nbr_list=nbr2)
    assert integ.n_carried_runs == 2


def test_list_carry_over_device_check_rejects_a_list_that_does_not_cover_x0(ca, O):
    """Should the host-side conditions ever let through positions the carried force list does not cover (here: forced,
    by moving particles in place and forging the version stamp), the device check halts launch 0, the list is renewed
    from x0 before anything has run, and the run is the one a list build on entry gives -- first kick included."""
    if not ca.integrators._CARRY_VALIDATE:
        pytest.skip("CHX_CARRY_VALIDATE=0: the device check this test exercises is switched off")
    moved = {}

    def tamper(k, state, integ, nbr):
        if k == 1:
            xs = state.positions
            xs[:64] += 0.31  # > skin / 2
            moved["x"] = xs.clone()
            offer = integ.__dict__.get("_carry")
            if offer is not None:
                offer["x_version"] = xs._version
    x, box, pot, a, nbr_a, integ_a, ra = _carry_segments(ca, O, True, 2, 40, n=2048, tamper=tamper)
    assert integ_a.n_carried_runs == 1 and integ_a.n_carry_rejected == 1
    _, _, _, b, nbr_b, integ_b, rb = _carry_segments(ca, O, False, 2, 40, n=2048, tamper=tamper)
    L = float(box[0, 0])
    assert torch.equal(ra[1]["x_in"], rb[1]["x_in"])
    assert ra[1]["rebuild_steps"] == rb[1]["rebuild_steps"]
    assert _mi_max(ra[1]["x"], rb[1]["x"], L) < 2e-5 and (ra[1]["v"] - rb[1]["v"]).abs().max().item() < 2e-3
    assert _mi_max(nbr_a.ref_positions, nbr_b.ref_positions, L) < 2e-5 and nbr_a.n_builds == nbr_b.n_builds


def test_list_carry_over_survives_a_rejected_force_list_renewal(ca, O, monkeypatch):
    """A renewal of the force list is enqueued without waiting for its status block; should the block later say that a
    capacity overflowed (here: the confirmation of the first renewal is made to fail), the launches since are void, the
    list is rebuilt the synchronous way, the state goes back to the renewal step with the reference's bookkeeping as it
    was there, and the run ends as the halting scheme's does."""
    import os

    from chiron_amd import neighbors

    if os.environ.get("CHX_SYNC_REBUILD") == "1":
        pytest.skip("every rebuild waits for its status block in this run: nothing is enqueued unconfirmed")
    calls = {"n": 0}
    orig = neighbors.NeighborListNsqrd._confirm_build

    def confirm(self, wait=True):
        spec = getattr(self, "_spec", None)
        if spec is not None and getattr(self, "_box_token", 0) is None and calls["n"] == 0:  # a private builder's first
            calls["n"] += 1
            orig(self, wait=True)
            self.n_builds -= 1
            self._adopt_positions = True
            try:
                self.build(self.ref_positions, self.box_vectors)
            finally:
                self._adopt_positions = False
            return False
        return orig(self, wait)

    monkeypatch.setattr(neighbors.NeighborListNsqrd, "_confirm_build", confirm)
    segments, steps = 8, 100
    # (list_carry_max_steps = 256: segments long enough for renewals inside them)
    x, box, pot, a, nbr_a, integ_a, ra = _carry_segments(ca, O, True, segments, steps, max_steps=256)
    monkeypatch.setattr(neighbors.NeighborListNsqrd, "_confirm_build", orig)
    assert calls["n"] == 1 and integ_a.n_internal_halts >= 2
    _, _, _, b, nbr_b, integ_b, rb = _carry_segments(ca, O, False, segments, steps)
    L = float(box[0, 0])
    for k in range(segments):
        assert ra[k]["rebuild_steps"] == rb[k]["rebuild_steps"], k
        assert _mi_max(ra[k]["x"], rb[k]["x"], L) < 2e-4, k
    assert nbr_a.n_builds == nbr_b.n_builds and _mi_max(nbr_a.ref_positions, nbr_b.ref_positions, L) < 2e-4


def test_list_carry_over_with_injected_noise_follows_the_oracle_step_for_step(ca, O):
    """The carried path against the NumPy oracle at the tolerance of a single run (injected noise, no chaos to hide
    behind): five 12-step segments under a thin skin, so that reference rebuild events AND renewals of the carried
    force list fall inside them; the oracle builds its list on every entry (chiron/integrators.py:169)."""
    import os

    from chiron_amd import integrators

    n, steps, segments, skin = 2500, 12, 5, 0.05
    x, box, pot, state, thermo, nbr, integ = _langevin_setup(ca, O, n, 0.3, 43, skin, 4.0)
    rng = np.random.default_rng(23)
    mass = np.full(n, 39.948, f32)
    kT = O.R_KJ * 300.0
    v0 = O.initial_velocities(mass, kT, rng.normal(size=(n, 3)).astype(f32))
    state.velocities = v0
    xo, vo = x, v0
    integ.list_carry = True
    events = 0
    if True:
        for k in range(segments):
            noise = rng.normal(size=(steps, n, 3)).astype(f32)
            state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr, noise=noise)
            xo, vo, rebuilds, _ = O.langevin_run(xo, vo, box, mass, SIGMA, EPS, CUTOFF, skin, 180, 0.004, 1.0, 300.0, steps,
                                                 lambda s: noise[s])
            assert integ.n_rebuilds == rebuilds, k
            events += rebuilds
            assert np.abs(state.positions.cpu().numpy() - xo).max() < 5e-6, k
            assert np.abs(state.velocities.cpu().numpy() - vo).max() < 5e-4 * max(1.0, np.abs(vo).max()), k
    assert integ.n_carried_runs == segments - 1 and getattr(integ, "n_carry_rejected", 0) == 0
    assert events >= 2 and integ.n_internal_halts >= 1  # both kinds of rebuild happened inside carried segments
