"""CPU-side tests: unit shim, state containers, error conventions, PRNG keys, and that
libchx.so loads and exports every symbol include/chx.h declares (no kernels run here)."""
import copy
import math
import os
import re

import numpy as np
import pytest
import torch

from chiron_amd import unit
from chiron_amd.rng import PRNGKey, fold_in, philox4x32, split, uniform
from chiron_amd.states import SamplerState, ThermodynamicState
from chiron_amd.topology import Element, Topology
from chiron_amd.utils import PRNG, get_list_of_mass, get_nr_of_particles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------
# unit shim against the values openmm.unit produces
# ---------------------------------------------------------------------------
def test_unit_conversions_md_system():
    md = unit.md_unit_system
    assert math.isclose((0.238 * unit.kilocalories_per_mole).value_in_unit_system(md), 0.995792)
    assert math.isclose((3.4 * unit.angstrom).value_in_unit_system(md), 0.34)
    assert math.isclose((1.0 * unit.femtoseconds).value_in_unit_system(md), 0.001)
    assert math.isclose((1.0 / unit.picoseconds).value_in_unit_system(md), 1.0)
    assert math.isclose((39.948 * unit.amu).value_in_unit_system(md), 39.948)
    kT = (unit.BOLTZMANN_CONSTANT_kB * unit.AVOGADRO_CONSTANT_NA * (300 * unit.kelvin)).value_in_unit_system(md)
    assert math.isclose(kT, 8.31446261815324e-3 * 300, rel_tol=1e-12)
    k = 100.0 * unit.kilocalories_per_mole / unit.angstroms**2
    assert math.isclose(k.value_in_unit_system(md), 41840.0)
    assert math.isclose((16.04 * unit.gram / unit.mole).value_in_unit(unit.amu), 16.04)


def test_unit_algebra_and_collapse():
    beta = 1.0 / (unit.BOLTZMANN_CONSTANT_kB * (300 * unit.kelvin))
    red = unit.Quantity(-1000.0, unit.kilojoule_per_mole) / unit.AVOGADRO_CONSTANT_NA
    red = red + (1.0 * unit.atmosphere) * (1000.0 * unit.nanometer**3)
    u = beta * red
    assert isinstance(u, float)
    assert math.isclose(u, (-1000.0e3 / 6.02214076e23 + 101325.0 * 1000e-27) / (1.380649e-23 * 300), rel_tol=1e-12)
    assert (1.0 * unit.nanometer) == (10.0 * unit.angstrom)
    assert not ((1.1 * unit.nanometer) == (1.0 * unit.kelvin))
    assert (3.0 * (0.34 * unit.nanometer)).value_in_unit(unit.nanometer) == pytest.approx(1.02)
    q = 0.1 * unit.nanometer
    q *= 1.1
    q /= 1.1
    assert q.value_in_unit(unit.nanometer) == pytest.approx(0.1)
    assert unit.nanometer.is_compatible(unit.angstrom) and not unit.nanometer.is_compatible(unit.radian)
    with pytest.raises(TypeError):
        (1.0 * unit.nanometer).value_in_unit(unit.kelvin)
    arr = np.ones((2, 3)) * unit.angstrom
    assert isinstance(arr, unit.Quantity) and arr.shape == (2, 3)
    assert np.allclose(arr.value_in_unit_system(unit.md_unit_system), 0.1)
    t = torch.ones(2, 3) * unit.nanometer
    assert isinstance(t, unit.Quantity) and isinstance(t[0][0], unit.Quantity)
    box = np.diag([10.0, 10.0, 10.0]) * unit.nanometer
    vol = box[0][0] * box[1][1] * box[2][2]
    assert vol.unit.is_compatible(unit.nanometer**3) and float(vol.value_in_unit(unit.nanometer**3)) == 1000.0
    assert np.all(copy.deepcopy(arr) == arr)


def test_topology_and_masses():
    top = Topology()
    el = Element.getBySymbol("Ar")
    chain = top.addChain()
    res = top.addResidue("system", chain)
    for _ in range(8):
        top.addAtom("Ar", el, res)
    assert get_nr_of_particles(top) == 8
    m = get_list_of_mass(top)
    assert np.allclose(m.value_in_unit_system(unit.md_unit_system), 39.948)
    ch4 = Element(1000, "CH4", "CH4", unit.Quantity(16.04, unit.gram / unit.mole))
    assert math.isclose(ch4.mass.value_in_unit(unit.amu), 16.04)


# ---------------------------------------------------------------------------
# PRNG keys
# ---------------------------------------------------------------------------
def test_prng_keys():
    PRNG.set_seed(1234)
    a = PRNG.get_random_key()
    b = PRNG.get_random_key()
    PRNG.set_seed(1234)
    assert PRNG.get_random_key() == a and a != b
    k = PRNGKey.from_seed(1234)
    k1, s1 = split(k)
    assert k1 != s1 and split(k) == (k1, s1)
    assert fold_in(k, 5) != fold_in(k, 6)
    assert 0.0 < uniform(k) < 1.0
    assert philox4x32((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)


def test_host_philox_c_helper_and_python_rounds_agree():
    """chiron_amd.rng.philox4x32 goes through the library's host helper (include/chx.h: chx_philox4x32_h) once it is
    loaded; the Python rounds stay the definition.  Same bits on the Random123 vectors and on random blocks."""
    from chiron_amd import _lib, rng

    _lib.load()  # (rng never loads the library itself: it binds to one the process has loaded for its kernels)
    rng.philox4x32((0, 0, 0, 0), (0, 0))
    assert rng._fast, "libchx is loaded: the C helper must have been bound"
    r = np.random.default_rng(3)
    for _ in range(500):
        c = tuple(int(v) for v in r.integers(0, 2**32, 4))
        k = tuple(int(v) for v in r.integers(0, 2**32, 2))
        assert rng.philox4x32(c, k) == rng.philox4x32_py(c, k)
    assert rng.philox4x32_py((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF)) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)


def test_key_handling_alone_neither_loads_the_library_nor_touches_the_gpu():
    """Round-4 review: the first split / fold_in / PRNG.get_random_key of a process used to load libchx -- and with it
    share torch's HIP runtime and initialise the device context -- as a hidden side effect.  A launcher that only
    hands out keys must stay clear of the GPU.  In a fresh interpreter: key handling leaves chiron_amd._lib unloaded
    and torch.cuda uninitialised, uses the Python rounds, and binds the C helper once something else has loaded the
    library (nothing is latched)."""
    import subprocess
    import sys

    code = (
        "import sys, torch\n"
        "from chiron_amd import _lib, rng\n"
        "from chiron_amd.utils import PRNG\n"
        "PRNG.set_seed(1234); k = PRNG.get_random_key(); a = rng.split(k); b = rng.fold_in(k, 3); u = rng.uniform(k)\n"
        "assert _lib._lib is None and not rng._fast, 'key handling loaded the library'\n"
        "assert not torch.cuda.is_initialized(), 'key handling initialised the GPU'\n"
        "assert 'libchx' not in open('/proc/self/maps').read()\n"
        "_lib.load(); a2 = rng.split(k)\n"
        "assert rng._fast and a2 == a and rng.fold_in(k, 3) == b and rng.uniform(k) == u\n"
        "print('ok')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_mc_key_chain_helper_matches_the_host_key_handling():
    """chx_mc_key_chain_h (include/chx.h): the chain the device Monte Carlo loop runs on -- per trial two splits
    (chiron/mcmc.py:724, :531), the proposal key and the uniform of the decision -- equals what the host loop gets from
    chiron_amd.rng.split / uniform, trial by trial."""
    import ctypes

    from chiron_amd import _lib, rng

    lib = _lib.load()
    n = 40
    tab = (ctypes.c_uint32 * (4 * (n + 1)))()
    chain = (ctypes.c_uint32 * (2 * (n + 1)))()
    assert lib.chx_mc_key_chain_h(2024, 7, n, tab, chain) == 0
    key = rng.PRNGKey(2024, 7)
    for i in range(n + 1):
        assert (chain[2 * i], chain[2 * i + 1]) == tuple(key)
        key, sub = rng.split(key)  # SamplerState.new_PRNG_key: the proposal key
        assert (tab[4 * i], tab[4 * i + 1]) == tuple(sub)
        key, acc = rng.split(key)  # the acceptance key
        u32 = np.frombuffer(np.uint32(tab[4 * i + 2]).tobytes(), np.float32)[0]
        assert u32 == np.float32(rng.uniform(acc))
        # word 3: fp32 log of that uniform (the device decides by comparison and evaluates exp only near the boundary)
        lu = np.frombuffer(np.uint32(tab[4 * i + 3]).tobytes(), np.float32)[0]
        assert abs(float(lu) - np.log(np.float64(u32))) <= 1.2e-7 * (abs(np.log(np.float64(u32))) + 1e-30) + 1e-45
    assert lib.chx_mc_key_chain_h(1, 2, -1, None, None) < 0


def test_host_philox_matches_oracle():
    from oracle import chiron_oracle as O

    key = PRNGKey(123456789, 987654321)
    o = O.philox4x32(np.arange(5), 7, 0, 3, key[0], key[1])
    for i in range(5):
        assert philox4x32((i, 7, 0, 3), key) == tuple(int(w[i]) for w in o)
    assert uniform(key, 7) == float(O.u32_to_uniform(np.uint32(philox4x32((0, 7, 0, 3), key)[0])))


# ---------------------------------------------------------------------------
# state containers (reference: chiron/tests/test_states.py, test_mcmc.py:289-324)
# ---------------------------------------------------------------------------
def _key():
    PRNG.set_seed(1234)
    return PRNG.get_random_key()


def test_sampler_state_conversion_and_inputs():
    s = SamplerState(unit.Quantity(np.array([[10.0, 10.0, 10.0]]), unit.nanometer), _key())
    assert torch.allclose(s.positions.cpu(), torch.tensor([[10.0, 10.0, 10.0]]))
    s = SamplerState(unit.Quantity(np.array([[10.0, 10.0, 10.0]]), unit.angstrom), _key())
    assert torch.allclose(s.positions.cpu(), torch.tensor([[1.0, 1.0, 1.0]]))
    assert s.positions.dtype == torch.float32 and s.number_of_particles == 1
    with pytest.raises(TypeError):
        SamplerState(positions=np.array([1, 2, 3]))
    with pytest.raises(ValueError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.radians), _key())
    with pytest.raises(TypeError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), _key(), velocities=np.array([1, 2, 3]))
    with pytest.raises(ValueError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), _key(),
                     velocities=unit.Quantity(np.array([1, 2, 3]), unit.nanometers))
    with pytest.raises(TypeError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), _key(), box_vectors=np.array([1, 2, 3]))
    with pytest.raises(ValueError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), _key(),
                     box_vectors=unit.Quantity(np.eye(3), unit.radians))
    with pytest.raises(ValueError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), _key(),
                     box_vectors=unit.Quantity(np.array([[1, 0, 0], [0, 1, 0]]), unit.nanometers))
    with pytest.raises(ValueError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), None)
    openmm_box = [
        unit.Quantity([4.031145745088737, 0.0, 0.0], unit.nanometer),
        unit.Quantity([0.0, 4.031145745088737, 0.0], unit.nanometer),
        unit.Quantity([0.0, 0.0, 4.031145745088737], unit.nanometer),
    ]
    st = SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), _key(), box_vectors=openmm_box)
    assert np.allclose(st.box_vectors.cpu().numpy(), np.eye(3) * 4.0311456, atol=1e-4)
    with pytest.raises(TypeError):
        SamplerState(unit.Quantity(np.array([[1, 2, 3]]), unit.nanometers), _key(), box_vectors=[123])
    # setters and key advance
    st.positions = np.zeros((1, 3))
    with pytest.raises(ValueError):
        st.velocities = np.zeros((2, 3))
    st.velocities = np.ones((1, 3))
    k0 = st.current_PRNG_key
    sub = st.new_PRNG_key
    assert st.current_PRNG_key != k0 and sub != st.current_PRNG_key
    c = copy.deepcopy(st)
    c.positions = np.ones((1, 3))
    assert float(st.positions.sum()) == 0.0


def test_thermodynamic_state_inputs_and_reduced_potential():
    from chiron_amd.potential import IdealGasPotential, LJPotential

    pot = IdealGasPotential(Topology.homogeneous(8))
    with pytest.raises(TypeError):
        ThermodynamicState(pot, temperature=300)
    with pytest.raises(ValueError):
        ThermodynamicState(pot, temperature=300 * unit.angstrom)
    with pytest.raises(TypeError):
        ThermodynamicState(pot, volume=1000)
    with pytest.raises(ValueError):
        ThermodynamicState(pot, volume=1000 * unit.kelvin)
    with pytest.raises(TypeError):
        ThermodynamicState(pot, pressure=100)
    with pytest.raises(ValueError):
        ThermodynamicState(pot, pressure=100 * unit.kelvin)
    ts = ThermodynamicState(pot, temperature=300 * unit.kelvin, volume=30 * unit.angstrom**3)
    assert ts.temperature == 300 * unit.kelvin and ts.pressure is None and ts.nr_of_particles == 8
    assert ts.check_variables() == ["temperature", "volume"]
    ts = ThermodynamicState(pot, temperature=300 * unit.kelvin, pressure=1.0 * unit.atmosphere)
    st = SamplerState(unit.Quantity(np.zeros((8, 3)), unit.nanometer), _key(),
                      box_vectors=unit.Quantity(np.diag([10.0, 10.0, 10.0]), unit.nanometer))
    from oracle import chiron_oracle as O

    assert math.isclose(ts.get_reduced_potential(st), O.reduced_potential(0.0, 300.0, 1.0, 1000.0), rel_tol=1e-9)
    with pytest.raises(TypeError):
        LJPotential(1)
    with pytest.raises(TypeError):
        IdealGasPotential(1)


def test_lj_and_neighbor_constructor_errors_without_gpu():
    from chiron_amd.neighbors import NeighborListNsqrd, OrthogonalPeriodicSpace, PairListNsqrd
    from chiron_amd.potential import LJPotential

    with pytest.raises(ValueError):
        LJPotential(None, sigma=1.0 * unit.kilocalories_per_mole)
    lj = LJPotential(None, sigma=0.34 * unit.nanometer, epsilon=0.238 * unit.kilocalories_per_mole,
                     cutoff=1.02 * unit.nanometer)
    assert math.isclose(lj.epsilon, 0.995792) and lj.cutoff == 1.02
    sp = OrthogonalPeriodicSpace()
    with pytest.raises(TypeError):
        NeighborListNsqrd(123)
    with pytest.raises(ValueError):
        NeighborListNsqrd(sp, cutoff=unit.Quantity(1, unit.radian))
    nbr = NeighborListNsqrd(sp, cutoff=1.0 * unit.nanometer, skin=0.2 * unit.nanometer, n_max_neighbors=7)
    assert not nbr.is_built and nbr.n_max_neighbors == 7
    nbr.is_built = True
    nbr.cutoff = 1.1 * unit.nanometer  # setters invalidate the list (neighbors.py:485)
    assert nbr.is_built is False
    with pytest.raises(TypeError):
        nbr.build_from_state(123)
    with pytest.raises(ValueError):
        PairListNsqrd(sp, cutoff=1.0 * unit.kelvin)
    with pytest.raises(TypeError):
        PairListNsqrd(5)
    with pytest.raises(ValueError):
        lj.compute_energy(np.zeros((2, 3), np.float32), nbr)  # unbuilt list


def test_kernels_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly without the device."""
    from chiron_amd import ops
    from chiron_amd._lib import ChxError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ChxError):
        ops.wrap(torch.zeros(4, 3), ops.make_box(np.eye(3, dtype=np.float32), True))
    with pytest.raises(ChxError):
        ops.lj_allpairs(torch.zeros(4, 3), ops.make_box(None, False), 1.0, 1.0, 1.0)


# ---------------------------------------------------------------------------
# Metropolis / autotune host logic
# ---------------------------------------------------------------------------
def test_mcmove_accept_rule_and_autotune():
    from chiron_amd.mcmc import MonteCarloBarostatMove, MonteCarloDisplacementMove, MoveSchedule

    mv = MonteCarloDisplacementMove(displacement_sigma=0.1 * unit.nanometer, number_of_moves=1)
    key = PRNGKey(1, 2)
    u = uniform(key)
    assert mv._accept_or_reject(0.3, key, "Metropolis-Hastings") is True
    assert mv._accept_or_reject(math.log(u) * 0.99, key, "Metropolis-Hastings") is True
    assert mv._accept_or_reject(math.log(u) * 1.01, key, "Metropolis-Hastings") is False
    mv.statistics = dict(n_accepted=7, n_proposed=10)
    mv._autotune()
    assert mv.displacement_sigma.value_in_unit(unit.nanometer) == pytest.approx(0.11)
    mv.statistics = dict(n_accepted=3, n_proposed=10)
    mv._autotune()
    assert mv.displacement_sigma.value_in_unit(unit.nanometer) == pytest.approx(0.1)
    b = MonteCarloBarostatMove(volume_max_scale=0.29, number_of_moves=1)
    b.statistics = dict(n_accepted=9, n_proposed=10)
    b._autotune()
    assert b.volume_max_scale == 0.3
    b.statistics = dict(n_accepted=1, n_proposed=10)
    b._autotune()
    assert b.volume_max_scale == pytest.approx(0.3 / 1.1)
    with pytest.raises(ValueError):
        MoveSchedule([("x", object())])


# ---------------------------------------------------------------------------
# the C ABI: library loads and exports everything the header declares
# ---------------------------------------------------------------------------
def _header_functions():
    text = open(os.path.join(ROOT, "include", "chx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from chiron_amd import _lib

    names = _header_functions()
    assert len(names) >= 25
    lib = _lib.load()  # raises if libchx.so is missing or a bound symbol does not resolve
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/chx.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.chx_version() == _lib.ABI_VERSION == 5
    # struct layouts agree with the header (x86-64 SysV)
    assert ctypes.sizeof(_lib.Box) == 16
    assert ctypes.sizeof(_lib.LangevinParams) == 272 == lib.chx_sizeof_langevin_t()
    assert _lib.LangevinParams.host_ctrl.offset == 216 and _lib.LangevinParams.snap4.offset == 224
    assert _lib.LangevinParams.ev_log_cap.offset == 232 and _lib.LangevinParams.ev_log.offset == 240
    assert _lib.LangevinParams.fl_inner.offset == 176 and _lib.LangevinParams.prune_phase.offset == 208
    # struct-by-value marshalling rule of include/chx.h: no float scalars before a box that follows
    # five or more integer/pointer arguments (mis-marshalled by this image's libffi)
    for name, (_, args) in _lib.SIGNATURES.items():
        if _lib.Box in args:
            i = args.index(_lib.Box)
            ints = sum(1 for a in args[:i] if a not in (ctypes.c_float, ctypes.c_double))
            floats = i - ints
            assert floats == 0 or ints < 5, name
    assert ctypes.sizeof(_lib.McParams) == 216
    assert _lib.McParams.beta.offset == 112 and _lib.McParams.ws.offset == 152


def test_cutoff_threshold_helper_is_exact():
    from chiron_amd import ops

    rng = np.random.default_rng(0)
    for c in list(rng.uniform(0.01, 5.0, 200).astype(np.float32)) + [np.float32(1.52), np.float32(1.02), np.float32(0.25)]:
        T = np.float32(ops.cutoff_threshold(float(c)))
        assert np.sqrt(T) < c and not (np.sqrt(np.nextafter(T, np.float32(np.inf))) < c)
    assert ops.cutoff_threshold(0.0) < 0.0


def test_cell_grid_helper():
    from chiron_amd import ops

    total, nc = ops.cell_grid(ops.make_box(np.diag([46.8805] * 3).astype(np.float32), True), 1.52, 262144)
    assert list(nc) == [30, 30, 30] and total == 27000
    total, nc = ops.cell_grid(ops.make_box(np.diag([23.4402] * 3).astype(np.float32), True), 1.52, 32768)
    assert list(nc) == [15, 15, 15]
    total, nc = ops.cell_grid(ops.make_box(np.diag([4.0] * 3).astype(np.float32), True), 1.52, 100)
    assert total == 0  # fewer than 3 cells per axis -> O(N^2) kernel
    total, nc = ops.cell_grid(ops.make_box(None, False), 1.52, 100)
    assert total == 0


def test_branch_free_minimum_image_formula_is_exact():
    """The hot kernels evaluate mod(d + L/2, L) - L/2 (chiron/neighbors.py:79) for |d| < L as the
    unsigned-integer minimum of {t - L, t, t + L}, t = d + L/2 (csrc/chx_common.h: min_image_fast,
    csrc/chx_lj.hip: min_image2).  Same arithmetic restated in NumPy fp32: bit-identical to the
    reference formula on random and adversarial inputs."""
    f32 = np.float32
    rng = np.random.default_rng(0)
    for L in (1.0, 4.275021399280942, 5.8601, 23.4402, 46.8805, 1e-3, 977.0, 3.0000002):
        L = f32(L)
        hL = f32(L * f32(0.5))
        d = rng.uniform(-1.0, 1.0, 2_000_000).astype(f32) * L
        edge = [0.0, -0.0, hL, -hL, np.nextafter(hL, f32(0)), np.nextafter(-hL, f32(0)), np.nextafter(hL, L),
                np.nextafter(-hL, -L), np.nextafter(L, f32(0)), -np.nextafter(L, f32(0)), 1e-30, -1e-30, 1e-45, -1e-45,
                np.nextafter(f32(0), f32(1)) * 3, L * f32(0.75), -L * f32(0.75)]
        tiny = (rng.uniform(-1, 1, 1000) * 1e-6).astype(f32)
        near = np.concatenate([hL + tiny * L, -hL + tiny * L, L * (1 - np.abs(tiny)), -L * (1 - np.abs(tiny))]).astype(f32)
        d = np.concatenate([d, np.array(edge, dtype=f32), near])
        d = d[np.abs(d) < L]
        ref = (np.mod(d + hL, L) - hL).astype(f32)
        t = (d + hL).astype(f32)
        a, c = (t - L).astype(f32), (t + L).astype(f32)
        m = np.minimum(np.minimum(a.view(np.uint32), t.view(np.uint32)), c.view(np.uint32)).view(f32)
        fast = (m - hL).astype(f32)
        assert np.array_equal(fast.view(np.uint32), ref.view(np.uint32)), float(L)


def test_row_filter_never_drops_a_pair_the_reference_predicate_keeps():
    """The filtered row kernel of the rebuilds inside run() (csrc/chx_cell_fl.hip: k_nl_build_sub_fl, sub_filter_of) tests
    every candidate with a CHEAP nearest image -- d - L rint(d / L), fused multiply-adds, one orientation -- against a
    widened threshold and hands only the survivors to the reference's predicate (mod(d + L/2, L) - L/2 in the pair's
    own orientation, chiron/neighbors.py:79, 773-782; r2 <= T(c)).  It may pass pairs the predicate then rejects; it must
    never reject one the predicate would keep.  Both restated in NumPy fp32 on pairs piled up around the cutoff, across
    the periodic seam, in boxes from 3 to 1000 nm (cubic and not), with coordinates up to five box lengths outside."""
    from oracle import chiron_oracle as O

    f32, f64 = np.float32, np.float64
    rng = np.random.default_rng(7)

    def threshold(c):  # largest fp32 t with sqrt(t) < c (chx_cutoff_threshold_h)
        t = f32(c) * f32(c)
        while np.sqrt(t) < c:
            t = np.nextafter(t, f32(np.inf))
        while not (np.sqrt(t) < c):
            t = np.nextafter(t, f32(-np.inf))
        return t

    def fma(a, b, c):  # one rounding: the product of two fp32 numbers is exact in fp64
        return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)

    kept_total = dropped_far = 0
    for box_l in ([3.7, 3.7, 3.7], [5.8601, 5.8601, 5.8601], [23.4402] * 3, [46.8805] * 3, [211.3, 211.3, 211.3], [997.0] * 3,
                  [4.1, 23.0, 9.7], [60.0, 5.0, 310.0]):
        L = np.array(box_l, dtype=f32)
        box = np.diag(L).astype(f32)
        for c in (f32(1.22), f32(1.52)):
            if 3 * c > L.min():
                continue  # (fewer than three cells per axis: the cell build is not used)
            T = threshold(c)
            lmax = L.max()
            il = (f32(1.0) / L).astype(f32)
            t_wide = f32(T * f32(1.001) + f32(6e-5) * lmax * np.sqrt(T))  # sub_filter_of
            m = 400_000
            xi = (rng.uniform(0, 1, (m, 3)) * L).astype(f32)
            u = rng.normal(size=(m, 3))
            u /= np.linalg.norm(u, axis=1, keepdims=True)
            # distances: most within a few ulp .. 1e-3 of the cutoff on either side, some well inside, some far outside
            rel = np.concatenate([rng.uniform(-1e-6, 1e-6, m // 4), rng.uniform(-1e-4, 1e-4, m // 4), rng.uniform(-2e-3, 2e-3, m // 4),
                                  rng.uniform(-0.9, 1.0, m - 3 * (m // 4))])
            s = (f64(c) * (1.0 + rel))[:, None]
            shift = rng.integers(-1, 2, size=(m, 3)) * L.astype(f64)  # the neighbour through another image
            xk = (xi.astype(f64) + s * u + shift).astype(f32)
            far_out = rng.integers(-5, 6, size=(m, 3)) * (rng.uniform(size=(m, 1)) < 0.1)  # a tenth of the pairs unwrapped
            xk = (xk.astype(f64) + far_out * L.astype(f64)).astype(f32)
            # the reference's predicate, in both orientations (row i of the half list takes x_i - x_k, row k the opposite)
            r_ik, _ = O.periodic_displacement(xi, xk, box)
            r_ki, _ = O.periodic_displacement(xk, xi, box)
            keep = np.zeros(m, dtype=bool)
            for r in (r_ik, r_ki):
                d2 = ((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2]).astype(f32)
                keep |= d2 <= T
            # the filter, exactly as phase 1 evaluates it
            d = (xi - xk).astype(f32)
            q = np.rint((d * il).astype(f32)).astype(f32)
            r = fma(-q, np.broadcast_to(L, q.shape), d)
            d2f = fma(r[:, 2], r[:, 2], fma(r[:, 1], r[:, 1], (r[:, 0] * r[:, 0]).astype(f32)))
            passed = (d2f <= t_wide) | (np.abs(q).max(axis=1) > f32(1.5))
            assert not np.any(keep & ~passed), (box_l, float(c), int(np.sum(keep & ~passed)))
            kept_total += int(keep.sum())
            dropped_far += int(np.sum(~passed))
    assert kept_total > 1_000_000 and dropped_far > 500_000  # (the test saw both sides of the cutoff, and the filter filters)


def test_c_abi_rejects_bad_arguments_before_touching_the_gpu():
    """Error convention of include/chx.h: a negative status plus a message, never an exception or a
    launch, when the arguments cannot be right (these paths return before any HIP call)."""
    import ctypes

    from chiron_amd import _lib

    lib = _lib.load()
    box = _lib.Box(1, 10.0, 10.0, 10.0)
    f = ctypes.c_float

    def failed(rc, needle):
        assert rc < 0
        msg = lib.chx_last_error().decode()
        assert needle in msg, msg

    failed(lib.chx_nl_build_nsq(None, -1, 8, box, f(1.0), None, None, None, None), "chx_nl_build_nsq")
    failed(lib.chx_validate_snap(None, None, 8, box, f(0.5), 0, None, None, None), "null pointer")
    failed(lib.chx_validate_snap(ctypes.c_void_p(64), ctypes.c_void_p(64), 8, box, f(0.0), 0, ctypes.c_void_p(64), None, None), "skin")
    failed(lib.chx_nl_build_nsq(None, 4, 8, box, f(1.0), None, None, None, None), "null pointer")
    failed(lib.chx_nl_build_nsq(ctypes.c_void_p(64), 4, 8, _lib.Box(1, 0.0, 1.0, 1.0), f(1.0), ctypes.c_void_p(64),
                                ctypes.c_void_p(64), ctypes.c_void_p(64), None), "box lengths")
    failed(lib.chx_lj_fl_energy_force(None, None, None, 8, 4, 1, box, f(1), f(1), f(1), None, None, None, None),
           "nothing to compute")
    failed(lib.chx_lj_fl_energy_force(None, None, None, 8, 4, 0, box, f(1), f(1), f(1), ctypes.c_void_p(64), None,
                                      None, None), "null pointer")
    failed(lib.chx_fl_fill(ctypes.c_void_p(64), ctypes.c_void_p(64), 8, 4, None, 3, ctypes.c_void_p(64),
                           ctypes.c_void_p(64), None), "power of two")
    failed(lib.chx_langevin_fused_chunk(None, None, None, 0, 1, 1, 1, None), "null params")
    sent = ctypes.c_int32(-1)
    failed(lib.chx_langevin_fused_run(None, None, None, 0, 1, 1, 4, ctypes.byref(sent), None), "null pointer")
    lp = _lib.LangevinParams()
    lp.n = 4
    lp.v = lp.mass = lp.ctrl = lp.fl = lp.deg = 64
    lp.fl_chunks = 1
    failed(lib.chx_langevin_fused_run(ctypes.byref(lp), ctypes.c_void_p(64), ctypes.c_void_p(128), 0, 1, 1, 4,
                                      ctypes.byref(sent), None), "host_ctrl")
    hc = (ctypes.c_int32 * _lib.CTRL_WORDS)(6, 0, 0, 7, -2, -2, 0, 0, 0, -1, -1, -1, 2**31 - 1, 0, 0, 0)
    lp.host_ctrl = ctypes.addressof(hc)
    failed(lib.chx_langevin_fused_run(ctypes.byref(lp), ctypes.c_void_p(64), ctypes.c_void_p(128), 0, 1, 1, 0,
                                      ctypes.byref(sent), None), "window")
    # the halt word (7) already rules out step 7: the launch loop stops before its first launch
    lib.chx_langevin_fused_run(ctypes.byref(lp), ctypes.c_void_p(64), ctypes.c_void_p(128), 7, 5, 2, 4, ctypes.byref(sent), None)
    assert sent.value == 0
    failed(lib.chx_mc_displace_chunk(None, 0, 1, None), "null parameter block")
    failed(lib.chx_lj_nl_energy_force(None, None, None, 8, 0, box, f(1), f(1), f(1), None, None, None, None), "m >= 1")
    failed(lib.chx_baoab_step(None, None, None, None, -3, box, 1, f(1), f(1), f(1), f(1), None, f(1), 0, 0, 0, None,
                              None, None), "n < 0")
    failed(lib.chx_kick(None, None, None, 5, f(0.1), None), "null pointer")
    p = _lib.McParams()
    p.n = 0
    failed(lib.chx_mc_displace_chunk(ctypes.byref(p), 0, 1, None), "n <= 0")
    # the threshold helper and the grid helper are pure host functions
    assert lib.chx_version() == _lib.ABI_VERSION == 5


def test_sampler_state_copy_with_and_box_token():
    """``_copy_with`` (what LangevinIntegrator.run returns) equals deepcopy + assignment
    (chiron/integrators.py:210-214): fresh arrays, a cloned box, the caller's state untouched; the box
    token -- how the integrator recognises "same box as at the list's last build" without reading the
    device -- is shared by copies and renewed whenever a box is assigned."""
    import copy

    x = np.arange(12, dtype=np.float32).reshape(4, 3)
    box = np.diag([2.0, 3.0, 4.0]).astype(np.float32)
    Quantity = unit.Quantity
    st = SamplerState(Quantity(x, unit.nanometer), PRNGKey(3, 4), velocities=Quantity(x * 0.5, unit.nanometer / unit.picosecond),
                      box_vectors=Quantity(box, unit.nanometer))
    x2, v2 = torch.full((4, 3), 7.0), torch.full((4, 3), -1.0)
    new = st._copy_with(x2, v2)
    ref = copy.deepcopy(st)
    ref.positions, ref.velocities = x2, v2
    assert torch.equal(new.positions, ref.positions) and torch.equal(new.velocities, ref.velocities)
    assert torch.equal(new.box_vectors, st.box_vectors) and new.box_vectors is not st.box_vectors
    assert new.current_PRNG_key == st.current_PRNG_key and new.number_of_particles == 4
    assert torch.equal(st.positions, torch.as_tensor(x))  # the source keeps its arrays
    new.box_vectors[0, 0] = 9.0
    assert float(st.box_vectors[0, 0]) == 2.0
    # the token follows the box VALUES
    assert new._box_token is st._box_token and ref._box_token is st._box_token
    old = st._box_token
    st.box_vectors = Quantity(box * 2, unit.nanometer)
    assert st._box_token is not old and new._box_token is old
    other = SamplerState(Quantity(x, unit.nanometer), PRNGKey(3, 4), box_vectors=Quantity(box, unit.nanometer))
    assert other._box_token is not old


def test_as_tensor_passes_ready_tensors_through():
    from chiron_amd._device import as_tensor

    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    out = as_tensor(t)
    assert out.dtype == torch.float32 and torch.equal(out, t)
    assert as_tensor(t.double()).dtype == torch.float32
    nc = torch.arange(12, dtype=torch.float32).reshape(3, 4).t()
    assert as_tensor(nc).is_contiguous() and torch.equal(as_tensor(nc), nc)
    assert as_tensor([[1, 2, 3]]).shape == (1, 3)


def test_host_side_of_the_c_abi_under_address_sanitizer(tmp_path):
    """the host half of libchx (entry points, argument validation, host helpers) is
    compiled from the library's own sources with -fsanitize=address,undefined (scripts/build_host_asan.sh) and
    driven by tests/host_asan/driver.cpp: helper sweeps plus an invalid call of every entry point.  Any
    sanitizer report or failed expectation makes the driver exit non-zero.  No GPU involved."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "asan")
    res = subprocess.run(["bash", os.path.join(root, "scripts", "build_host_asan.sh"), out], capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([os.path.join(out, "host_asan_driver")], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert "host ABI checks passed" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


def test_build_is_keyed_on_a_source_hash_not_on_mtimes(monkeypatch):
    """build() must recompile exactly when sources/flags differ from what the shipped .so was built from."""
    from chiron_amd import _lib, build as b

    assert os.path.exists(b.OUT), "run __graft_entry__.build() first"
    have = b.binary_hash()
    assert have is not None and re.fullmatch(r"[0-9a-f]{64}", have)
    assert have == b.source_hash() and not b.needs_build()
    # touching a source (newer mtime, same bytes) does not trigger a rebuild ...
    src = b.sources()[0]
    st = os.stat(src)
    os.utime(src, (st.st_atime, st.st_mtime + 3600))
    try:
        assert not b.needs_build()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    # ... a different flag set or different bytes does
    assert b.source_hash(b.FLAGS + ["-DX"]) != have
    monkeypatch.setattr(b, "FLAGS", b.FLAGS + ["-DSOMETHING_ELSE"])
    assert b.needs_build()
    # and the loader refuses a library that does not match the tree
    lib = _lib.load()
    assert lib.chx_source_hash().decode() == have
    assert lib.chx_sizeof_langevin_t() == __import__("ctypes").sizeof(_lib.LangevinParams)
    assert lib.chx_sizeof_mc_t() == __import__("ctypes").sizeof(_lib.McParams)
    assert lib.chx_sizeof_build_riders_t() == __import__("ctypes").sizeof(_lib.BuildRiders) == 128
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.ChxError, match="built from other sources"):
        _lib.load()
    monkeypatch.setenv("CHX_ALLOW_STALE_LIB", "1")
    assert _lib.load() is not None


def test_threefry_stream_host_twins_follow_jax():
    """chiron_amd.rng's JaxKey handling (product code) against the Random123 known answers and against the oracle's
    independent NumPy restatement of jax.random (oracle/jax_prng.py, pinned by the reference's literals)."""
    from chiron_amd import rng
    from oracle import jax_prng as JX

    # Random123 kat_vectors, threefry2x32 20 rounds
    assert rng.threefry2x32((0, 0), 0, 0) == (0x6B200159, 0x99BA4EFE)
    assert rng.threefry2x32((0xFFFFFFFF, 0xFFFFFFFF), 0xFFFFFFFF, 0xFFFFFFFF) == (0x1CB996FC, 0xBB002BE7)
    assert rng.threefry2x32((0x13198A2E, 0x03707344), 0x243F6A88, 0x85A308D3) == (0xC4923A9C, 0x483DF7A0)
    for seed in (0, 1234, 2**33 + 17):
        k = rng.JaxKey.from_seed(seed)
        assert tuple(k) == JX.PRNGKey(seed)
        for _ in range(3):
            new, sub = rng.split(k)
            a, b = JX.split(tuple(k))
            assert isinstance(new, rng.JaxKey) and isinstance(sub, rng.JaxKey)
            assert tuple(new) == a and tuple(sub) == b
            assert rng.uniform(sub) == float(JX.uniform(b))
            assert 2.0 * rng.uniform(sub) - 1.0 == float(JX.uniform(b, (), -1.0, 1.0))
            for n in (1, 2, 5, 8):
                assert rng._jax_bits(sub, n) == [int(v) for v in JX.random_bits(b, n)]
            k = new
    # the default stream is untouched and the two key types do not mix
    p = rng.PRNGKey(1, 2)
    assert type(rng.split(p)[0]) is rng.PRNGKey and rng.split(p) != rng.split(rng.JaxKey(1, 2))
    PRNG.set_seed(7, stream="threefry")
    try:
        k1 = PRNG.get_random_key()
        assert isinstance(k1, rng.JaxKey) and tuple(k1) == JX.split(JX.PRNGKey(7))[1]
        st = SamplerState(unit.Quantity(np.zeros((2, 3)), unit.nanometer), k1)
        sub = st.new_PRNG_key  # chiron/states.py:150-154 on the jax stream
        assert tuple(sub) == JX.split(tuple(k1))[1] and tuple(st.current_PRNG_key) == JX.split(tuple(k1))[0]
        import copy as _copy

        assert isinstance(_copy.deepcopy(st).current_PRNG_key, rng.JaxKey)
        with pytest.raises(ValueError):
            PRNG.set_seed(1, stream="mt19937")
    finally:
        PRNG.set_seed(1234)


def test_no_kernel_reads_the_dispatch_packet(tmp_path):
    """Guard against a 5x performance cliff found in round 2: when the compiler parks a private array in LDS it
    indexes it by the flat work-item id, which needs the workgroup size from the AQL dispatch packet -- a scalar
    load from HOST memory in every wavefront (the fused step kernel took 24-30 us per launch instead of 5).  No
    kernel of the library may need the dispatch pointer, and the step kernel's LDS use must be exactly its force
    table.  (Device assembly only: no GPU involved.)"""
    import shutil
    import subprocess

    from chiron_amd import build as b

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC")]
    for src in b.sources():
        out = str(tmp_path / (os.path.basename(src) + ".s"))
        subprocess.run([hipcc] + flags + ["-S", "--offload-device-only", "-o", out, src], check=True, capture_output=True, timeout=600)
        text = open(out).read()
        assert ".amdhsa_user_sgpr_dispatch_ptr 1" not in text, f"{os.path.basename(src)}: a kernel reads the dispatch packet"
        # ... and none may spill to or park a struct in scratch memory (a RowLoads returned through a conditional
        # expression once put 224 bytes per lane there)
        spilled = re.findall(r"\.amdhsa_kernel (\S+)\s+(?:\.amdhsa_group_segment_fixed_size \d+\s+)?\.amdhsa_private_segment_fixed_size ([1-9]\d*)", text)
        # (one deliberate exception: the steady-state batched step kernel is compiled for SEVEN waves per SIMD -- eight
        # replicas in one grid are a throughput problem -- and pays for the seventh with a few spilled registers; measured
        # in round 6: 428 k replica-steps/s at six waves without scratch, 441 k at seven with 40 bytes, 348 k at eight)
        spilled = [(k, b_) for k, b_ in spilled if not (k.startswith("_Z15k_lj_step_batchILi") and "ELb1EE" in k and int(b_) <= 64)]
        assert not spilled, f"{os.path.basename(src)}: kernels with scratch memory: {spilled[:4]}"
        if src.endswith("chx_lj.hip"):
            sizes = {}
            for m in re.finditer(r"\.amdhsa_kernel _Z9k_lj_stepILi(\d+)ELb[01]ELb[01]ELb[01]EE\S*\s+\.amdhsa_group_segment_fixed_size (\d+)", text):
                sizes.setdefault(int(m.group(1)), set()).add(int(m.group(2)))
            assert sizes, "no k_lj_step instantiation found"
            for g, lds in sizes.items():
                assert lds == {3 * (256 // g) * 4}, (g, lds)  # sF[3 * PPB] floats and nothing else


def test_pin_process_cpus_picks_a_neighbourhood_and_can_be_switched_off():
    """``pin_process_cpus``: an aligned group of neighbouring CPUs around the current one, all
    of them allowed before; CHX_PIN_CPUS=0 leaves the affinity alone, CHX_PIN_CPUS=<n> sets the width.  Run in child
    processes: the affinity of the test runner itself must not change."""
    import os
    import subprocess
    import sys

    if not hasattr(os, "sched_setaffinity") or len(os.sched_getaffinity(0)) < 4:
        pytest.skip("needs sched_setaffinity and at least four CPUs")
    code = ("import os, json; before = sorted(os.sched_getaffinity(0)); "
            "from chiron_amd._device import pin_process_cpus; got = pin_process_cpus(); "
            "print(json.dumps([before, got, sorted(os.sched_getaffinity(0))]))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def child(env_value, local_rank=None):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("CHX_PIN_CPUS", None)
        env.pop("LOCAL_RANK", None)
        if local_rank is not None:
            env["LOCAL_RANK"] = str(local_rank)
        if env_value is not None:
            env["CHX_PIN_CPUS"] = env_value
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        import json

        return json.loads(r.stdout.strip().splitlines()[-1])

    before, got, after = child(None)
    if len(before) > 4:
        assert got == after and len(got) == 4 and set(got) <= set(before)
        at = before.index(got[0])
        assert at % 4 == 0 and got == before[at:at + 4]
    else:
        assert got is None and after == before
    before, got, after = child("0")
    assert got is None and after == before
    before, got, after = child("2")
    assert got == after and len(got) == 2 and set(got) <= set(before)
    if len(before) >= 8:  # the ranks of one node get different groups
        g0, g1 = child(None, 0)[1], child(None, 1)[1]
        assert len(g0) == len(g1) == 4 and not set(g0) & set(g1)


def test_gpu_local_cpus_reads_the_numa_neighbourhood_from_sysfs(tmp_path, monkeypatch):
    """``gpu_local_cpus`` (what ``pin_process_cpus`` chooses its CPUs from): KFD topology order of the compute nodes,
    PCI address from location_id / domain, the device's local_cpulist; integer VISIBLE_DEVICES lists are applied,
    anything else (UUIDs, a missing tree) gives None and the caller falls back to the CPUs it is allowed."""
    from chiron_amd import _device as D

    def mk(rel, text):
        path = tmp_path / rel
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(text)

    mk("class/kfd/kfd/topology/nodes/0/properties", "cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    mk("class/kfd/kfd/topology/nodes/2/properties", "simd_count 1024\nlocation_id 3328\ndomain 0\n")      # 0000:0d:00.0
    mk("class/kfd/kfd/topology/nodes/10/properties", "simd_count 1024\nlocation_id 58624\ndomain 0\n")    # 0000:e5:00.0
    mk("bus/pci/devices/0000:0d:00.0/local_cpulist", "0-63,128-191\n")
    mk("bus/pci/devices/0000:e5:00.0/local_cpulist", "64-127,192-255\n")
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    root = str(tmp_path)
    assert D.gpu_local_cpus(0, root) == list(range(0, 64)) + list(range(128, 192))
    assert D.gpu_local_cpus(1, root) == list(range(64, 128)) + list(range(192, 256))  # node 10 sorts after node 2
    assert D.gpu_local_cpus(2, root) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert D.gpu_local_cpus(0, root)[0] == 64
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-0123abcd")
    assert D.gpu_local_cpus(0, root) is None
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert D.gpu_local_cpus(0, str(tmp_path / "nowhere")) is None
    assert D._parse_cpulist("3,5-7\n") == [3, 5, 6, 7]


# ---------------------------------------------------------------------------
# round 3: advisor findings
# ---------------------------------------------------------------------------
def test_integrator_copies_drop_per_process_device_state():
    """MultiStateSampler deep-copies its sampler (chiron/multistate.py:266-268).  What a run() leaves on the
    integrator -- host-mapped pinned slots the step kernel stores into, device scratch, identity-keyed caches --
    must not be cloned (a deep-copied pinned tensor is pageable, its numpy view no longer aliases it) nor shared."""
    import pickle

    from chiron_amd.integrators import LangevinIntegrator
    from chiron_amd.mcmc import LangevinDynamicsMove

    integ = LangevinIntegrator(timestep=2.0 * unit.femtoseconds, report_interval=7)
    marker = object()
    for name in LangevinIntegrator._PER_PROCESS_STATE:
        setattr(integ, name, marker)
    integ.n_rebuilds = 3
    for clone in (copy.deepcopy(integ), copy.copy(integ), pickle.loads(pickle.dumps(LangevinIntegrator(report_interval=7)))):
        for name in LangevinIntegrator._PER_PROCESS_STATE:
            assert name not in clone.__dict__, name
        assert clone.report_interval == 7
    clone = copy.deepcopy(integ)
    assert clone.n_rebuilds == 3 and clone.timestep is not integ.timestep
    assert clone.timestep.value_in_unit(unit.femtoseconds) == 2.0
    assert getattr(integ, "_gen_ring") is marker  # the original keeps its resources
    # through the objects that hold an integrator
    move = LangevinDynamicsMove(timestep=1.0 * unit.femtoseconds, number_of_steps=5)
    move.integrator._gen_ring = marker
    m2 = copy.deepcopy(move)
    assert m2.integrator is not move.integrator and "_gen_ring" not in m2.integrator.__dict__
    assert move.integrator._gen_ring is marker


def test_list_carry_over_offer_is_taken_only_by_the_run_that_continues_the_trajectory():
    """Host side of the list carry-over (LangevinIntegrator._take_carry): the offer a run() leaves is taken by the next
    one only for the same list object, the very position tensor that run returned (unwritten since), the same box
    values, cutoff and skin, a short enough run and no reporter -- and it is an offer for ONE call."""
    import torch

    from chiron_amd import integrators
    from chiron_amd.integrators import LangevinIntegrator
    from chiron_amd.neighbors import NeighborListNsqrd, OrthogonalPeriodicSpace

    class State:
        pass

    class Owner:
        _last_max_n = 40

    def setup():
        nbr = NeighborListNsqrd(OrthogonalPeriodicSpace(), cutoff=1.0 * unit.nanometer, skin=0.5 * unit.nanometer,
                                n_max_neighbors=180)
        nbr.is_built, nbr._cs_md, nbr._box_token = True, 1.5, object()
        st = State()
        st.positions, st._box_token = torch.zeros((8, 3)), nbr._box_token
        integ = LangevinIntegrator()
        owner = Owner()
        integ._carry = dict(owner=owner, public=nbr, x=st.positions, x_version=st.positions._version, cs_md=1.5,
                            skin=nbr._skin, box_token=nbr._box_token)
        return integ, nbr, st, owner

    integ, nbr, st, owner = setup()
    assert integ._take_carry(st, nbr, 20) is owner
    assert integ.__dict__.get("_carry") is None and integ._take_carry(st, nbr, 20) is None  # consumed
    for spoil in ("steps", "zero", "copy", "written", "box", "skin", "list", "reporter", "traj", "capacity", "switch"):
        integ, nbr, st, owner = setup()
        steps, other = 20, nbr
        if spoil == "steps":
            steps = integ.list_carry_max_steps + 1
        elif spoil == "zero":
            steps = 0
        elif spoil == "copy":
            st.positions = st.positions.clone()
        elif spoil == "written":
            st.positions.add_(1.0)
        elif spoil == "box":
            st._box_token = object()
        elif spoil == "skin":
            nbr.skin = 0.5 * unit.nanometer  # (a new object, and is_built / _cs_md are void with it)
        elif spoil == "list":
            other = NeighborListNsqrd(OrthogonalPeriodicSpace())
        elif spoil == "reporter":
            integ.reporter = object()
        elif spoil == "traj":
            integ.save_traj_in_memory = True
        elif spoil == "capacity":
            owner._last_max_n = 170  # > 0.85 * n_max_neighbors: the reference's growth rule must see every build
        elif spoil == "switch":
            integ.list_carry = False  # (LangevinIntegrator(list_carry=False))
        assert integ._take_carry(st, other, steps) is None, spoil
        assert integ.__dict__.get("_carry") is None, spoil
    # an offer that names the caller's own list needs that list's last confirmed cell build: none here
    integ, nbr, st, owner = setup()
    integ._carry["owner"] = nbr
    assert integ._take_carry(st, nbr, 20) is None


def test_launch_window_poll_policy_and_its_diagnostics_are_host_calls():
    """``chx_set_poll_sleep_ns`` (how the launch window waits; 0 = busy-wait) returns the previous setting, and
    ``chx_poll_stats`` reads / resets the window's counters: both are plain host calls (no GPU needed)."""
    import ctypes

    from chiron_amd import _lib

    lib = _lib.load()
    old = lib.chx_set_poll_sleep_ns(0)
    try:
        assert lib.chx_set_poll_sleep_ns(7000) == 0
        assert lib.chx_set_poll_sleep_ns(-5) == 7000  # (negative: clamped to busy-waiting)
        assert lib.chx_set_poll_sleep_ns(old) == 0
    finally:
        lib.chx_set_poll_sleep_ns(old)
    st = (ctypes.c_int64 * 4)(1, 2, 3, 4)
    assert lib.chx_poll_stats(st, 1) == 0
    assert all(v >= 0 for v in st)
    assert lib.chx_poll_stats(st, 0) == 0 and list(st) == [0, 0, 0, 0]  # (nothing has polled since the reset)
    assert lib.chx_poll_stats(None, 0) == 0


def test_cell_build_refuses_a_grid_axis_beyond_its_packed_coordinates():
    """The cell kernels carry a cell's grid coordinates in 10 bits per axis; a caller-supplied grid with a longer
    axis would alias cells silently (advisor finding, round 2).  Refused before any launch."""
    import ctypes

    from chiron_amd import _lib

    lib = _lib.load()
    box = _lib.Box(1, 4000.0, 10.0, 10.0)
    nc = (ctypes.c_int32 * 3)(2000, 3, 3)
    p = ctypes.c_void_p(64)
    rc = lib.chx_nl_build_cell(p, 8, 8, box, ctypes.c_float(1.5), nc, p, p, p, p, p, p, p, p, None)
    assert rc < 0 and "1024 cells per axis" in lib.chx_last_error().decode()
    rc = lib.chx_nl_build_cell_fl(p, 8, 8, box, ctypes.c_float(1.5), nc, p, p, p, p, p, p, p, p, 4, 1, p, p, p, None, None)
    assert rc < 0 and "1024 cells per axis" in lib.chx_last_error().decode()
    # the helper itself never hands out such a grid
    total = lib.chx_cell_grid_h(_lib.Box(1, 4000.0, 4000.0, 4000.0), ctypes.c_float(1.5), 1 << 30, nc)
    assert max(nc) <= 1024 and total == nc[0] * nc[1] * nc[2]


def test_reporter_npz_round_trip(tmp_path):
    """the on-disk log keeps the reference's layout -- one array per key, first axis = frames
    (chiron/reporters.py:133-144) -- and reads back what was reported, across flushes and a second save."""
    from chiron_amd.reporters import BaseReporter, LangevinDynamicsReporter, MCReporter

    BaseReporter.set_directory(str(tmp_path / "logs"))
    try:
        rep = LangevinDynamicsReporter(buffer_size=2)
        assert str(rep.log_file_path).endswith("langevin_reporter.npz")
        frames = []
        for i in range(5):
            x = torch.full((4, 3), float(i))
            frames.append(x.numpy().copy())
            rep.report({"positions": x, "potential_energy": np.float32(-i), "step": i, "box_vectors": np.eye(3, dtype=np.float32)})
        path = rep.save()
        back = LangevinDynamicsReporter.load(path)
        assert sorted(back) == ["box_vectors", "positions", "potential_energy", "step"]
        assert back["positions"].shape == (5, 4, 3) and back["box_vectors"].shape == (5, 3, 3)
        assert np.array_equal(back["positions"], np.stack(frames))
        assert np.array_equal(back["step"], np.arange(5)) and np.array_equal(back["potential_energy"], -np.arange(5, dtype=np.float32))
        for key in back:  # the file equals what get_property serves from memory
            assert np.array_equal(back[key], rep.get_property(key)), key
        rep.report({"positions": torch.zeros(4, 3), "potential_energy": np.float32(9), "step": 5, "box_vectors": np.eye(3, dtype=np.float32)})
        again = LangevinDynamicsReporter.load(rep.save())
        assert again["positions"].shape[0] == 6 and int(again["step"][-1]) == 5
        # the reference's keyword (chiron/reporters.py:220-224, :278, :303-308): file_name=; positional and the old name= too
        from chiron_amd.reporters import MultistateReporter

        for cls in (LangevinDynamicsReporter, MCReporter, MultistateReporter):
            assert str(cls(file_name="by_keyword", buffer_size=3).log_file_path).endswith("by_keyword.npz")
            assert str(cls("positional").log_file_path).endswith("positional.npz")
            assert str(cls(name="old_spelling").log_file_path).endswith("old_spelling.npz")
        mc = MCReporter(1)
        mc.report({"n_accepted": 3})
        assert MCReporter.load(mc.save(tmp_path / "mc.npz"))["n_accepted"].tolist() == [3]
    finally:
        BaseReporter.set_directory(None)


def test_list_carry_is_a_constructor_argument_and_the_environment_only_its_default():
    """Round-3 review item 10: the list carry-over (a semantic deviation from chiron/integrators.py:169 -- the caller's
    list is rebuilt lazily) is chosen by an argument a caller can see; CHX_LIST_CARRY / CHX_LIST_CARRY_MAX_STEPS only
    supply the defaults."""
    import subprocess
    import sys

    from chiron_amd import integrators

    a = integrators.LangevinIntegrator()
    assert a.list_carry is integrators._CARRY_DEFAULT and a.list_carry_max_steps == integrators._CARRY_MAX_STEPS_DEFAULT
    b = integrators.LangevinIntegrator(list_carry=False, list_carry_max_steps=7, launch_window=0)
    assert b.list_carry is False and b.list_carry_max_steps == 7 and b.launch_window == 1  # (a window of 0 launches cannot make progress)
    assert integrators.LangevinIntegrator(list_carry=True).list_carry is True
    code = ("from chiron_amd import integrators as I; i = I.LangevinIntegrator(); "
            "j = I.LangevinIntegrator(list_carry=True, list_carry_max_steps=9); print(i.list_carry, i.list_carry_max_steps, j.list_carry, j.list_carry_max_steps)")
    env = dict(os.environ, CHX_LIST_CARRY="0", CHX_LIST_CARRY_MAX_STEPS="5")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["False", "5", "True", "9"]


def test_sampler_state_box_copies_come_from_a_pool_and_stay_independent():
    """SamplerState._box_copy (round 4): the box a returned state carries is a private copy (the reference deep-copies
    the state, chiron/integrators.py:210), made 32 at a time instead of one clone launch per run(); copies are
    independent memory, the states of one trajectory share the pool, a new box starts a new one, a deep copy none."""
    import copy

    from chiron_amd.states import SamplerState

    box = unit.Quantity(np.eye(3, dtype=np.float32) * 3.0, unit.nanometer)
    s = SamplerState(unit.Quantity(np.zeros((4, 3), np.float32), unit.nanometer), PRNGKey(1, 2), box_vectors=box)
    t, pools, ptrs = s, set(), set()
    for _ in range(70):
        t = t._copy_with(torch.zeros(4, 3), torch.zeros(4, 3))
        pools.add(id(t.__dict__["_box_pool"][1]))
        ptrs.add(t.box_vectors.data_ptr())
        assert torch.equal(t.box_vectors, s.box_vectors) and t.box_vectors.data_ptr() != s.box_vectors.data_ptr()
    assert len(pools) == 3 and len(ptrs) == 70  # 70 copies out of three pools of 32, every one its own memory
    a, b = s._box_copy(), s._box_copy()
    a._value.mul_(2.0)
    assert torch.equal(b._value, s.box_vectors) and not torch.equal(a._value, b._value)
    d = copy.deepcopy(t)
    assert "_box_pool" not in d.__dict__ and torch.equal(d.box_vectors, t.box_vectors)
    s.box_vectors = np.eye(3, dtype=np.float32) * 5.0
    assert float(s._box_copy()._value[0, 0]) == 5.0
    # in-place edits of a state's box (the property hands out the tensor itself) are carried into the next copy, as
    # with the clone() the pool replaced -- whether the state holds the tensor the pool was filled from or one of the
    # pool's own blocks -- and an edit of one state's block does not leak into its siblings (round-4 review)
    r1 = s._copy_with(torch.zeros(4, 3), torch.zeros(4, 3))
    r2 = r1._copy_with(torch.zeros(4, 3), torch.zeros(4, 3))
    r1.box_vectors.mul_(2.0)  # a block of the pool
    r3 = r1._copy_with(torch.zeros(4, 3), torch.zeros(4, 3))
    assert float(r3.box_vectors[0, 0]) == 10.0 and float(r2.box_vectors[0, 0]) == 5.0
    r4 = r2._copy_with(torch.zeros(4, 3), torch.zeros(4, 3))
    assert float(r4.box_vectors[0, 0]) == 5.0 and r4.box_vectors.data_ptr() != r2.box_vectors.data_ptr()
    s.box_vectors[0, 0] = 7.0  # the tensor the first pool was filled from
    assert float(s._copy_with(torch.zeros(4, 3), torch.zeros(4, 3)).box_vectors[0, 0]) == 7.0
    assert float(r4._copy_with(torch.zeros(4, 3), torch.zeros(4, 3)).box_vectors[0, 0]) == 5.0


def test_host_philox_is_safe_to_call_from_several_threads():
    """rng.philox4x32 goes through the library's host helper, and ctypes releases the GIL for the call: the words must
    not travel through a buffer shared between threads (MultiStateSampler(concurrent_replicas=2) splits the keys of two
    replicas at once; with a shared buffer one replica occasionally got the other's key -- round 4)."""
    import threading

    from chiron_amd import _lib, rng

    _lib.load()
    want = {k: rng.philox4x32_py((k, 1, 2, 3), (k, 7)) for k in range(64)}
    assert rng.philox4x32((5, 1, 2, 3), (5, 7)) == want[5]
    assert rng._fast  # (the library's helper is what the threads below race on)
    bad = []

    def work(lane):
        for rep in range(4000):
            k = (lane * 17 + rep) % 64
            if rng.philox4x32((k, 1, 2, 3), (k, 7)) != want[k]:
                bad.append((lane, rep))

    threads = [threading.Thread(target=work, args=(lane,)) for lane in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad[:5]


def test_step_code_object_is_current_and_holds_every_step_kernel():
    """The step queue (csrc/chx_aql.hip) dispatches the fused step kernels from libchx_step.hsaco, the device half of
    csrc/chx_lj.hip compiled by itself (chiron_amd/build.py).  CPU check: the file is the one built from this tree,
    it holds every instantiation the launcher can ask for, and their kernarg segments are the sizes the hand-laid
    argument blocks cover (StepKernarg 320 bytes, ApKernarg 280 for a 276-byte segment: chx_aql_dispatch refuses a block shorter than the kernel's)."""
    import shutil
    import subprocess

    from chiron_amd import build

    assert build.step_object_current(), "libchx_step.hsaco is missing or was built from another tree: run build()"
    readelf = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not available")
    notes = subprocess.run([readelf, "--notes", build.STEP_OUT], capture_output=True, text=True, timeout=120).stdout
    sizes = {}
    size = None
    for line in notes.splitlines():
        line = line.strip()
        if line.startswith(".kernarg_segment_size:"):
            size = int(line.split(":")[1])
        elif line.startswith(".name:") and size is not None:
            sizes[line.split(":", 1)[1].strip()] = size
            size = None
    for g in (1, 2, 4, 8):  # (the lane counts a list builder chooses; round 6 dropped 16 / 32 / 64 from the fused kernels)
        for hot in (0, 1):
            for prune in (0, 1):
                for defer in (0, 1):
                    names = [n for n in sizes if n.startswith(f"_Z9k_lj_stepILi{g}ELb{hot}ELb{prune}ELb{defer}EE")]
                    assert len(names) == 1, (g, hot, prune, defer)
                    assert sizes[names[0]] == 320
            # the batched launch (several replicas in one grid): CHX_BATCH_MAX = 8 items of 304 bytes
            names = [n for n in sizes if n.startswith(f"_Z15k_lj_step_batchILi{g}ELb{hot}EE")]
            assert len(names) == 1, (g, hot)
            assert sizes[names[0]] == 8 * 304, sizes[names[0]]
    assert not [n for n in sizes if n.startswith("_Z9k_lj_stepILi16E") or n.startswith("_Z9k_lj_stepILi64E")]
    # the units whose kernels ride on the queue while a run() is in progress (csrc/chx_launch.h): unpack, the list build
    # chain, pack -- one code object each, every kernel the rebuild launches present
    want = {"core": ["k_unpack_langevin", "k_permute_unpack"],
            "neighbors": ["k_cell_assign", "k_cell_scan", "k_cell_scatter", "k_cell_sort"],
            "cell_fl": ["k_nl_build_cell_fl", "k_nl_build_cellwise_fl", "k_build_epilogue", "k_pack_langevin",
                        "k_pack_validate_langevin", "k_validate_snap"],
            "mc": ["k_mc_trial", "k_mc_prologue", "k_mc_decide"]}
    for unit, kernels in want.items():
        path = build.queue_object({v: k for k, v in build.QUEUE_UNITS.items()}[unit])
        assert os.path.exists(path), path
        text = subprocess.run([readelf, "--notes", path], capture_output=True, text=True, timeout=120).stdout
        for k in kernels:
            assert k in text, (unit, k)
    ap = [n for n in sizes if n.startswith("_Z9k_ap_stepILi64ELb1EE") or n.startswith("_Z9k_ap_stepILi16ELb0EE")]
    assert len(ap) == 2 and all(276 <= sizes[n] <= 280 for n in ap), {n: sizes[n] for n in ap}  # (sizeof(ApKernarg) = 280)


def test_library_refuses_to_load_without_the_step_queues_code_objects():
    """The step queue falls back to HIP launches when its code objects cannot be loaded -- a working path, but a silently
    slower one.  With the sources beside the library (an in-tree build) a missing or stale libchx_step.hsaco is therefore
    refused at load time, like a stale libchx.so; CHX_AQL=0 (no queue wanted) loads without."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from chiron_amd import build\n"
            "build.step_object_current = lambda: False\n"
            "from chiron_amd import _lib\n"
            "try:\n"
            "    _lib.load()\n"
            "    print('LOADED')\n"
            "except _lib.ChxError as e:\n"
            "    print('REFUSED', 'step queue' in str(e))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("CHX_AQL", "CHX_LIB", "CHX_ALLOW_STALE_LIB")}
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split()[-2:] == ["REFUSED", "True"], (out.stdout, out.stderr[-1500:])
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(env, CHX_AQL="0"), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and out.stdout.split()[-1] == "LOADED", (out.stdout, out.stderr[-1500:])
