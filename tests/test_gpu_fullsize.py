"""GPU parity at BASELINE.json's full sizes (C4: N = 262,144; C3: N = 32,768).

The NumPy oracle cannot sweep these sizes, so parity is pinned through (a) oracle results on
random SAMPLES of rows/particles (bit-exact for neighbor rows, fp64 tolerance for forces) and
(b) size-independent properties: pair-count symmetry, Newton's third law, agreement of two
independent kernels/layouts, permutation and translation invariance, independence of the
trajectory from how steps are batched into launches.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

f32 = np.float32
SIGMA, EPS = 0.34, 0.238 * 4.184
CUTOFF, SKIN = 3 * SIGMA, 0.5
N4, N3 = 262144, 32768


@pytest.fixture(scope="module")
def O():
    from oracle import chiron_oracle

    return chiron_oracle


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from chiron_amd import _lib, integrators, mcmc, neighbors, ops, potential, states, unit, utils

    _lib.load()
    return type("NS", (), dict(unit=unit, neighbors=neighbors, potential=potential, states=states, ops=ops,
                               utils=utils, integrators=integrators, mcmc=mcmc))


def dev(a):
    return torch.as_tensor(np.asarray(a, dtype=f32)).cuda()


def _pot(ca, n):
    from chiron_amd.topology import Topology

    u = ca.unit
    return ca.potential.LJPotential(Topology.homogeneous(n), sigma=SIGMA * u.nanometer,
                                    epsilon=EPS * u.kilojoule_per_mole, cutoff=CUTOFF * u.nanometer)


def _nbr(ca, m=180, skin=SKIN):
    u = ca.unit
    return ca.neighbors.NeighborListCell(ca.neighbors.OrthogonalPeriodicSpace(), cutoff=CUTOFF * u.nanometer,
                                         skin=skin * u.nanometer, n_max_neighbors=m)


@pytest.fixture(scope="module")
def c4(ca, O):
    x, box = O.lj_lattice(N4, 0.1, SIGMA, seed=1234)
    nbr = _nbr(ca)
    nbr.build(x, box)
    assert nbr._perm is not None, "cell path was not taken"
    return x, box, nbr


def _oracle_row(O, x, box, i, cutoff_skin):
    """Row i of the reference half list: ascending j > i with d < cutoff + skin (chiron/neighbors.py:548-626)."""
    n = x.shape[0]
    xi = np.broadcast_to(x[i], (n, 3))
    _, d = O.periodic_displacement(xi, x, box)
    mask = (d < f32(cutoff_skin)) & (np.arange(n) > i)
    return np.nonzero(mask)[0].astype(np.uint32)


def test_c4_neighbor_rows_match_oracle_on_a_sample(ca, O, c4):
    x, box, nbr = c4
    nl = nbr.neighbor_list.cpu().numpy().view(np.uint32)
    n_nbr = nbr.n_neighbors.cpu().numpy()
    mask = nbr.neighbor_mask.cpu().numpy()
    m = nl.shape[1]
    rng = np.random.default_rng(2024)
    rows = np.concatenate([rng.integers(0, N4, 60), [0, 1, N4 - 2, N4 - 1]])
    for i in rows:
        row = _oracle_row(O, x, box, int(i), CUTOFF + SKIN)
        k = len(row)
        assert n_nbr[i] == k
        assert np.array_equal(nl[i, :k], row)
        # padding rule (:603-623): fill = first neighbor (0 if none), +1 if that is i itself
        fill = row[0] if k else 0
        fill = fill + 1 if fill == i else fill
        assert np.all(nl[i, k:] == fill)
        assert np.array_equal(mask[i], (np.arange(m) < k).astype(np.int32))
    # global structure: every listed id is larger than its row, rows ascend, counts are symmetric
    real = np.arange(m)[None, :] < n_nbr[:, None]
    assert np.all(nl[real] > np.repeat(np.arange(N4, dtype=np.uint32), n_nbr))
    asc = (np.diff(nl.astype(np.int64), axis=1) > 0) | ~real[:, 1:]
    assert asc.all()
    p_nl = int(n_nbr.sum())
    assert 3.5e6 < p_nl < 6e6  # ~4.9e6 list pairs at rho* = 0.1
    fl = nbr.force_list
    assert int(fl.deg.sum()) == 2 * p_nl
    # full degree of particle i = own row + appearances in other rows
    appear = np.bincount(nl[real], minlength=N4)
    deg_user = np.empty(N4, np.int64)
    deg_user[fl.perm.cpu().numpy().view(np.uint32)] = fl.deg.cpu().numpy()
    assert np.array_equal(deg_user, n_nbr + appear)


def test_c4_force_energy_properties(ca, O, c4):
    x, box, nbr = c4
    pot = _pot(ca, N4)
    rng = np.random.default_rng(7)
    x2 = (x + rng.normal(0, 0.04, size=x.shape)).astype(f32)  # inside the skin guarantee
    e, f = pot.compute_energy_and_force(x2, nbr)
    e, f = float(e), f.cpu().numpy()
    fmax = np.abs(f).max()
    # Newton's third law: the net force vanishes up to fp32 summation noise
    assert np.abs(f.astype(np.float64).sum(axis=0)).max() < 1e-5 * fmax * np.sqrt(N4) + 1e-3
    # two independent kernels and data layouts (force list in cell order vs. the public half list)
    f_h, e_h = ca.ops.lj_nl_energy_force(dev(x2), nbr.neighbor_list, nbr.n_neighbors, nbr._box_h, SIGMA, EPS, CUTOFF)
    assert abs(e - float(e_h[0])) <= 2e-6 * abs(e)
    assert np.abs(f - f_h.cpu().numpy()).max() <= 2e-5 * max(1.0, fmax)
    # fp64 brute force over all N for a sample of particles
    L = np.diag(box).astype(np.float64)
    x64 = x2.astype(np.float64)
    for i in rng.integers(0, N4, 48):
        r = x64[i] - x64
        r -= L * np.round(r / L)
        d2 = (r * r).sum(axis=1)
        sel = (d2 < CUTOFF ** 2) & (d2 > 0)
        s6 = (SIGMA ** 2 / d2[sel]) ** 3
        fi = ((24 * EPS * (2 * s6 * s6 - s6) / d2[sel])[:, None] * r[sel])
        ref = fi.sum(axis=0)
        assert np.abs(f[i] - ref).max() <= 1e-5 * max(1.0, np.abs(fi).sum(axis=0).max())
    # permutation invariance: the same configuration with shuffled particle ids
    perm = rng.permutation(N4)
    nbr_p = _nbr(ca)
    nbr_p.build(x[perm], box)
    e_p, f_p = pot.compute_energy_and_force(x2[perm], nbr_p)
    assert abs(float(e_p) - e) <= 2e-6 * abs(e)
    assert np.abs(f_p.cpu().numpy() - f[perm]).max() <= 2e-5 * max(1.0, fmax)
    assert int(nbr_p.n_neighbors.sum()) == int(nbr.n_neighbors.sum())
    # translation invariance under the periodic boundary conditions
    shift = np.array([3.7, -11.2, 20.05], f32)
    xs = O.periodic_wrap((x + shift).astype(f32), box)
    nbr_s = _nbr(ca)
    nbr_s.build(xs, box)
    e0 = float(pot.compute_energy(x, nbr))
    e_s = float(pot.compute_energy(xs, nbr_s))
    assert abs(e_s - e0) <= 1e-5 * abs(e0)


@pytest.fixture(scope="module")
def CO():
    from oracle import c_oracle

    c_oracle.lib()  # (compiles oracle/oracle.c on first use, hash-checked: oracle/build_oracle.py)
    return c_oracle


@pytest.mark.parametrize("n", [N3, N4], ids=["c3-32768", "c4-262144"])
def test_every_list_row_and_every_force_match_the_oracle_at_full_size(ca, O, CO, n):
    """ALL N particles, not a sample (round-2 review): (a) every row of the neighbor list -- counts and ids, bit
    exact -- against the C oracle's cell-grid restatement of chiron/neighbors.py:548-626 (itself pinned to the
    O(N^2) oracle in the CPU suite); (b) every particle's force and the total energy from the force-list kernel
    against fp64 (chiron/potential.py:193-213, :322-332) at SURVEY 8(c)'s tolerances; (c) the same for the
    public-half-list kernel."""
    x, box = O.lj_lattice(n, 0.1, SIGMA, seed=1234)
    nbr = _nbr(ca)
    nbr.build(x, box)
    assert nbr._perm is not None, "cell path was not taken"
    rows, deg = CO.full_rows_cell(x, box, CUTOFF + SKIN, threads=CO.effective_cores())
    want_n, want_flat = CO.half_list_from_full_rows(rows, deg)
    n_nbr = nbr.n_neighbors.cpu().numpy()
    assert np.array_equal(n_nbr, want_n)
    nl = nbr.neighbor_list.cpu().numpy().view(np.uint32)
    m = nl.shape[1]
    real = np.arange(m)[None, :] < n_nbr[:, None]
    assert np.array_equal(nl[real], want_flat)  # every real entry of every row, in order
    # padding rule (:603-623) and mask on every row
    first = np.where(n_nbr > 0, nl[:, 0], 0).astype(np.int64)
    fill = np.where(first == np.arange(n), first + 1, first)
    assert np.array_equal(nl.astype(np.int64)[~real], np.broadcast_to(fill[:, None], nl.shape)[~real])
    assert np.array_equal(nbr.neighbor_mask.cpu().numpy(), real.astype(np.int32))
    del rows, want_flat, real
    # forces / energy on perturbed positions inside the skin guarantee
    pot = _pot(ca, n)
    rng = np.random.default_rng(7)
    x2 = (x + rng.normal(0, 0.04, size=x.shape)).astype(f32)
    e64, f64, eabs, fabs_max, pairs = CO.lj_cell_f64(x2, box, SIGMA, EPS, CUTOFF, threads=CO.effective_cores())
    assert pairs > 5 * n // 2  # ~5.6 partners inside the cutoff at rho* = 0.1
    e, f = pot.compute_energy_and_force(x2, nbr)
    f = f.cpu().numpy().astype(np.float64)
    assert abs(float(e) - e64) <= 1e-5 * eabs, (float(e), e64, eabs)
    err = np.abs(f - f64).max()
    assert err <= 1e-5 * max(1.0, fabs_max), (err, fabs_max)
    f_h, e_h = ca.ops.lj_nl_energy_force(dev(x2), nbr.neighbor_list, nbr.n_neighbors, nbr._box_h, SIGMA, EPS, CUTOFF)
    assert abs(float(e_h[0]) - e64) <= 1e-5 * eabs
    assert np.abs(f_h.cpu().numpy().astype(np.float64) - f64).max() <= 2e-5 * max(1.0, fabs_max)  # (float atomics)


def _run_langevin(ca, O, n, chunk, steps, skin, seed=1234):
    u = ca.unit
    x, box = O.lj_lattice(n, 0.1, SIGMA, seed=seed)
    pot = _pot(ca, n)
    ca.utils.PRNG.set_seed(1234)
    state = ca.states.SamplerState(u.Quantity(x, u.nanometer), ca.utils.PRNG.get_random_key(),
                                   box_vectors=u.Quantity(box, u.nanometer))
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    nbr = _nbr(ca, skin=skin)
    integ = ca.integrators.LangevinIntegrator(timestep=2.0 * u.femtoseconds, collision_rate=1.0 / u.picoseconds,
                                              chunk_size=chunk)
    out, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
    return out, nbr, integ


def test_c4_langevin_is_independent_of_launch_batching(ca, O):
    """N = 262,144, fused step kernel, device RNG: chunk sizes 5 and 32 give the same trajectory bit for
    bit, rebuild at the same steps, and keep the kinetic temperature (chiron/integrators.py:179-195)."""
    a, nbr_a, integ_a = _run_langevin(ca, O, N4, 5, 96, 0.06)
    b, nbr_b, integ_b = _run_langevin(ca, O, N4, 32, 96, 0.06)
    assert integ_a.n_rebuilds == integ_b.n_rebuilds and integ_a.n_rebuilds >= 1
    assert torch.equal(a.positions, b.positions) and torch.equal(a.velocities, b.velocities)
    assert torch.equal(nbr_a.ref_positions, nbr_b.ref_positions)
    assert torch.equal(nbr_a.n_neighbors, nbr_b.n_neighbors)
    v = a.velocities.cpu().numpy().astype(np.float64)
    kT = O.R_KJ * 300.0
    assert abs((39.948 * v ** 2).mean() / kT - 1.0) < 0.02
    x = a.positions.cpu().numpy()
    L = float(a.box_vectors[0][0])
    assert x.min() >= 0.0 and x.max() < L * (1 + 1e-6)


@pytest.mark.parametrize("n", [N3, N4], ids=["c3-32768", "c4-262144"])
def test_launch_window_width_does_not_change_the_trajectory_at_full_size(ca, O, n, monkeypatch):
    """The metric configuration's loop at two extremes of the bounded launch queue (chx_langevin_fused_run): the default
    window (12 launches ahead of the device, the halt seen within a handful of no-op launches) and a window as long as
    the run (everything enqueued at once, every halt drained as dozens of no-op launches) end in the same bits --
    positions, velocities, rebuild count, rebuilt list -- at N = 32,768 (lanes per particle 4, per-particle build
    kernel) and N = 262,144 (two lanes, cell-wise build kernel).  Thin skin: several rebuilds inside a short run."""
    orig = ca.integrators.LangevinIntegrator.__init__
    res = []
    for window in (12, 4096):  # (12 = the default; set explicitly so that a CHX_LAUNCH_WINDOW override does not move it)
        def init(self, *a, _w=window, **k):
            orig(self, *a, **k)
            self.launch_window = _w
        monkeypatch.setattr(ca.integrators.LangevinIntegrator, "__init__", init)
        res.append(_run_langevin(ca, O, n, 32, 160 if n == N3 else 96, 0.06))
    monkeypatch.setattr(ca.integrators.LangevinIntegrator, "__init__", orig)
    (a, nbr_a, integ_a), (b, nbr_b, integ_b) = res
    assert integ_a.launch_window == 12 and integ_b.launch_window == 4096
    assert integ_a.n_rebuilds == integ_b.n_rebuilds >= 1
    assert torch.equal(a.positions, b.positions) and torch.equal(a.velocities, b.velocities)
    assert torch.equal(nbr_a.ref_positions, nbr_b.ref_positions) and torch.equal(nbr_a.n_neighbors, nbr_b.n_neighbors)
    assert (integ_a.prune_passes, integ_a.full_list_launches) == (integ_b.prune_passes, integ_b.full_list_launches)
    # the long queue drains at least as many dead launches (a timing-dependent diagnostic: with launches that cost the
    # host as much as the device a dozen are rarely queued, and the two counts can meet)
    assert getattr(integ_b, "noop_launches", 0) >= getattr(integ_a, "noop_launches", 0)


def test_c3_mc_device_loop_equals_host_loop(ca, O, monkeypatch):
    """MonteCarloDisplacementMove at N = 32,768: device-resident loop == reference-shaped host loop."""
    u = ca.unit
    res = []
    for host in ("1", "0"):
        monkeypatch.setenv("CHX_MC_HOST_LOOP", host)
        x, box = O.lj_lattice(N3, 0.1, SIGMA, seed=1234)
        pot = _pot(ca, N3)
        ca.utils.PRNG.set_seed(1234)
        state = ca.states.SamplerState(u.Quantity(x, u.nanometer), ca.utils.PRNG.get_random_key(),
                                       box_vectors=u.Quantity(box, u.nanometer))
        thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
        nbr = _nbr(ca, skin=0.02)
        nbr.build_from_state(state)
        move = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=0.0012 * u.nanometer, number_of_moves=40)
        s2, _, n2 = move.update(state, thermo, nbr)
        res.append((s2.positions.cpu().numpy(), tuple(s2._current_PRNG_key), move.statistics, n2.ref_positions.cpu().numpy()))
    assert res[0][2] == res[1][2] and 0 < res[0][2]["n_accepted"]
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][3], res[1][3])
    assert not np.array_equal(res[0][3], O.lj_lattice(N3, 0.1, SIGMA, seed=1234)[0])  # at least one rebuild happened


@pytest.mark.parametrize("subset", [False, True], ids=["all-particles", "atom-subset"])
def test_c4_mc_inner_list_on_a_grid_that_is_not_co_resident(ca, O, monkeypatch, subset):
    """Round-4 review (high): the trial launch tested the refresh word that its own tail books the NEXT refresh into.
    On a grid that is not co-resident -- N = 262,144 with four lanes per particle = 4096 workgroups, twice what the chip
    holds -- the last workgroups of a refreshing launch start after the first have booked trial + 1, skip the refresh
    and walk inner-list rows nobody has written for this list.  The booking fires whenever consecutive displacements
    differ by more than inner skin / 2 = 0.05 nm: sigma_d = 0.03 nm does that in every trial.  Now launch t reads the
    word of slot t & 1 and books into the other one.  Energies (every report), decisions, key chain and positions must
    be those of the loop without an inner list, bit for bit.  (The round-4 binary also passes this test on the MI355X:
    the word is read with a scalar load, and the scalar cache of a CU kept serving the value its first workgroup had
    fetched -- an accident of the cache, not a guarantee, so the test guards the configuration, it does not reproduce
    the fault.)"""
    u = ca.unit
    monkeypatch.setenv("CHX_FL_G", "4")
    monkeypatch.setenv("CHX_MC_HOST_LOOP", "0")
    x, box = O.lj_lattice(N4, 0.1, SIGMA, seed=1234)
    pot = _pot(ca, N4)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    ids = [int(i) for i in np.random.default_rng(5).choice(N4, 3000, replace=False)] if subset else None
    runs = {}
    for tag, s_in in (("off", 0.0), ("on", 0.1)):
        monkeypatch.setattr(ca.mcmc, "_MC_INNER_SKIN", s_in)
        ca.utils.PRNG.set_seed(1234)
        state = ca.states.SamplerState(u.Quantity(x, u.nanometer), ca.utils.PRNG.get_random_key(),
                                       box_vectors=u.Quantity(box, u.nanometer))
        nbr = _nbr(ca)
        nbr.build_from_state(state)
        assert nbr.force_list.g == 4
        from chiron_amd.reporters import MCReporter

        rep = MCReporter("mc_c4_inner", 1)
        move = ca.mcmc.MonteCarloDisplacementMove(displacement_sigma=(0.05 if subset else 0.03) * u.nanometer,
                                                  number_of_moves=48, atom_subset=ids, reporter=rep, report_interval=1)
        assert move._device_loop_applies(state, thermo, nbr)
        s2, _, n2 = move.update(state, thermo, nbr)
        runs[tag] = dict(x=s2.positions.cpu().numpy(), key=tuple(s2._current_PRNG_key), stats=dict(move.statistics),
                         e=np.asarray(rep.get_property("potential_energy"), dtype=np.float64),
                         # (current and PROPOSED energy of the last two trials: with every proposal rejected the
                         # reports alone would repeat the starting energy whatever the launches summed)
                         e_last=move._dev_ws["energy"].cpu().numpy().copy(),
                         refreshes=move.inner_list_refreshes)
    on, off = runs["on"], runs["off"]
    assert off["refreshes"] == 0 and on["refreshes"] >= 24, on["refreshes"]  # (the booking fires in most trials)
    assert len(off["e"]) == 48 and np.array_equal(on["e"], off["e"])
    assert np.array_equal(on["e_last"], off["e_last"]) and np.isfinite(off["e_last"]).all()
    assert off["stats"]["n_proposed"] == 48
    if subset:
        assert 0 < off["stats"]["n_accepted"]


@pytest.mark.parametrize("n", [N3, N4], ids=["c3-32768", "c4-262144"])
def test_step_kernel_step_63_matches_oracle_at_full_size(ca, O, CO, n):
    """The fused step kernel (k_lj_step, steady-state instantiation, pruned inner list, device Philox) at the
    BASELINE sizes: a 63-step run gives frame 63 (x, v after the trailing kick); a 64-step run from the same
    start is bit-identical up to there and adds step 63.  For EVERY particle that step is redone by the
    oracle -- fp64 forces on frame 63 (oracle/oracle.c: orc_lj_cell_f64), O.baoa_half (chiron/integrators.py:181-188)
    with the oracle's own Philox normals for (particle, step 63), wrap (:239), fp64 forces on the new frame,
    trailing kick (:195) -- and compared with the kernel's frame 64: x to 5e-6 nm, v to 5e-4."""
    u = ca.unit
    steps = 64
    x0, box = O.lj_lattice(n, 0.1, SIGMA, seed=1234)
    pot = _pot(ca, n)
    thermo = ca.states.ThermodynamicState(pot, temperature=300 * u.kelvin)
    outs = []
    for k in (steps - 1, steps):
        ca.utils.PRNG.set_seed(1234)
        state = ca.states.SamplerState(u.Quantity(x0, u.nanometer), ca.utils.PRNG.get_random_key(),
                                       box_vectors=u.Quantity(box, u.nanometer))
        key0 = tuple(state._current_PRNG_key)
        integ = ca.integrators.LangevinIntegrator(timestep=1.0 * u.femtoseconds, collision_rate=1.0 / u.picoseconds,
                                                  chunk_size=32)
        out, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=_nbr(ca))
        outs.append((out.positions.cpu().numpy(), out.velocities.cpu().numpy(), integ.n_rebuilds,
                     getattr(integ, "prune_passes", None)))
    (xa, va, rb_a, _), (xb, vb, rb_b, prunes) = outs
    assert rb_a == rb_b == 0  # dt = 1 fs from the lattice: the 0.5 nm skin holds for 64 steps
    if prunes is not None and os.environ.get("CHX_PRUNE_EVERY") is None and os.environ.get("CHX_INNER_SKIN") is None:
        assert prunes >= 2  # the inner list was refreshed at steps 0 and 32: step 63 walked the pruned table
    # every particle (round 2 compared 48): fp64 forces over the oracle's own cell grid
    idx = np.arange(n)
    mass = np.full(n, 39.948, f32)
    kT, a, b = O.langevin_constants(300.0, 0.001, 1.0)
    run_key = O.key_split(key0)[1]  # what SamplerState.new_PRNG_key hands to run() (chiron/integrators.py:124)
    xi = O.philox_normals3(n, steps - 1, O.STREAM_LANGEVIN, run_key)
    cores = CO.effective_cores()
    f_a = CO.lj_cell_f64(xa, box, SIGMA, EPS, CUTOFF, threads=cores)[1]
    x1, v1 = O.baoa_half(xa, va, f_a.astype(f32), mass, 0.001, a, b, kT, xi)
    x1 = O.periodic_wrap(x1, box)
    # (a coordinate that lands within rounding of a box face may wrap on one side only: compare modulo L)
    L = float(box[0, 0])
    dx = np.abs(xb - x1)
    dx = np.minimum(dx, np.abs(dx - L))
    assert dx.max() < 5e-6, dx.max()
    f_b = CO.lj_cell_f64(xb, box, SIGMA, EPS, CUTOFF, threads=cores)[1]
    v2 = O.kick(v1, f_b.astype(f32), mass, 0.001)
    assert np.abs(vb - v2).max() < 5e-4 * max(1.0, np.abs(v2).max()), np.abs(vb - v2).max()
    # the step really moved things: frame 64 is not frame 63
    assert np.abs(xb[idx] - xa[idx]).max() > 1e-5
    # and the system is a sane 300 K fluid after 64 steps
    assert abs((39.948 * vb.astype(np.float64) ** 2).mean() / kT - 1.0) < 0.05


def test_c5_replica_exchange_rehearsal_four_replicas_on_one_gpu():
    """Config C5's workflow on real kernels with the one GPU this box has: `bench.py --gpus 4` launches four ranks
    itself (no torchrun), each propagating its own C3 replica (temperature ladder) with the HIP kernels on the shared
    device and exchanging labels every 1000 steps; the collective runs over gloo because RCCL refuses two ranks on
    one device -- the RCCL transport itself is what the driver's 8-GPU run exercises."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--backend", "gloo", "--share-gpu",
                          "--steps", "3000", "--warmup", "200", "--steady-steps", "0", "--cpu-steps", "0", "--extra-c4", "0"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = lines[0]
    cfg = d["config"]
    assert d["n_gpus"] == 4 and cfg["group_world_size"] == 4 and cfg["collective_backend"] == "gloo" and cfg["shared_gpu"]
    assert cfg["replica_exchanges"] == 3 and len(cfg["per_rank_steps_per_s"]) == 4
    assert all(v > 1000 for v in cfg["per_rank_steps_per_s"])  # every rank really stepped its replica
    assert d["value"] > 4 * 1000 and d["scaling"] == "weak"
    assert cfg["swap_acceptance"] is not None and 0.0 <= cfg["swap_acceptance"] <= 1.0
    assert d["roofline"]["kernel_us"] > 0
