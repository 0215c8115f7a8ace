"""World-size-2 (gloo, CPU) tests of the replica-exchange layer: the all-gather, the shared
deterministic swap decisions and the label bookkeeping.  The energies are injected (no kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chiron_amd import unit
from chiron_amd.multistate import MultiStateSampler, ReplicaExchange, neighbor_swap_decisions
from chiron_amd.potential import IdealGasPotential
from chiron_amd.rng import PRNGKey
from chiron_amd.states import ThermodynamicState
from chiron_amd.topology import Topology


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ladder(k, pressure=None):
    pot = IdealGasPotential(Topology.homogeneous(4))
    return [ThermodynamicState(pot, temperature=300.0 * (1.5 ** r) * unit.kelvin, pressure=pressure) for r in range(k)]


def _energies(rank: int, it: int) -> float:
    # replica 0 sits in a much HIGHER energy basin than replica 1: swapping it to the hot state is downhill
    return [-100.0, -900.0][rank] + 3.0 * it


def _worker(rank: int, world: int, port: int, out_dir: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rex = ReplicaExchange(_ladder(world), PRNGKey(11, 22))
        log = []
        for it in range(6):
            idx, scale = rex.exchange(torch.tensor([_energies(rank, it)], dtype=torch.float32))
            log.append((idx, scale, rex.state_of_replica.copy(), rex.u_kn.copy()))
        np.save(os.path.join(out_dir, f"states_{rank}.npy"), np.array([l[2] for l in log]))
        np.save(os.path.join(out_dir, f"scales_{rank}.npy"), np.array([l[1] for l in log]))
        np.save(os.path.join(out_dir, f"ukn_{rank}.npy"), np.array([l[3] for l in log]))
        np.save(os.path.join(out_dir, f"acc_{rank}.npy"), rex.n_accepted)
    finally:
        dist.destroy_process_group()


def test_replica_exchange_two_ranks_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s0 = np.load(tmp_path / "states_0.npy")
    s1 = np.load(tmp_path / "states_1.npy")
    assert np.array_equal(s0, s1), "ranks disagree on the replica -> state map"
    assert np.array_equal(np.load(tmp_path / "ukn_0.npy"), np.load(tmp_path / "ukn_1.npy"))
    assert np.array_equal(np.load(tmp_path / "acc_0.npy"), np.load(tmp_path / "acc_1.npy"))
    # every map is a permutation and matches the serial rule on the gathered matrix
    states = _ladder(world)
    key = PRNGKey(11, 22)
    cur = np.arange(world)
    ukn = np.load(tmp_path / "ukn_0.npy")
    betas = np.array([1.0 / (8.31446261815324e-3 * 300.0 * 1.5 ** r) for r in range(world)])
    for it in range(6):
        e = np.array([np.float32(_energies(r, it)) for r in range(world)], dtype=np.float64)
        assert np.allclose(ukn[it], e[:, None] * betas[None, :], rtol=1e-6)
        cur, _, _ = neighbor_swap_decisions(ukn[it], cur, it, key)
        assert sorted(cur) == list(range(world))
        assert np.array_equal(cur, s0[it])
    # the first even iteration is a strongly downhill swap: it must be accepted, with the matching velocity rescale
    assert list(s0[0]) == [1, 0]
    sc0 = np.load(tmp_path / "scales_0.npy")
    sc1 = np.load(tmp_path / "scales_1.npy")
    assert np.isclose(sc0[0], np.sqrt(1.5)) and np.isclose(sc1[0], np.sqrt(1 / 1.5))
    assert np.load(tmp_path / "acc_0.npy").sum() >= 2


def test_swap_rule_detailed_balance_and_single_rank():
    # symmetric energies: log_p = 0 -> always accepted; NaN -> rejected
    key = PRNGKey(1, 1)
    u = np.zeros((2, 2))
    new, n_prop, n_acc = neighbor_swap_decisions(u, np.arange(2), 0, key)
    assert list(new) == [1, 0] and n_prop[0, 1] == 1 and n_acc[0, 1] == 1
    u[0, 1] = np.nan
    new, _, n_acc = neighbor_swap_decisions(u, np.arange(2), 0, key)
    assert list(new) == [0, 1] and n_acc.sum() == 0
    # odd iterations pair states (1,2), (3,4)...: with K = 2 nothing is proposed
    new, n_prop, _ = neighbor_swap_decisions(np.zeros((2, 2)), np.arange(2), 1, key)
    assert n_prop.sum() == 0
    # acceptance frequency follows min(1, exp(-delta)) over many shared-key draws
    K = 2
    delta = 1.0
    u = np.array([[0.0, delta], [0.0, 0.0]])  # u[a,k+1]+u[b,k]-u[a,k]-u[b,k+1] = delta
    acc = sum(int(neighbor_swap_decisions(u, np.arange(K), 2 * i, PRNGKey(5, i))[2].sum() > 0) for i in range(4000))
    assert abs(acc / 4000 - np.exp(-delta)) < 0.03
    # world_size 1 works without a process group
    rex = ReplicaExchange(_ladder(1), key, rank=0, world_size=1)
    idx, scale = rex.exchange(torch.tensor([1.0]))
    assert idx == 0 and scale == 1.0
    with pytest.raises(RuntimeError):
        ReplicaExchange(_ladder(3), key, rank=0, world_size=2)


def test_reduced_row_with_pressure():
    states = _ladder(2, pressure=1.0 * unit.atmosphere)
    rex = ReplicaExchange(states, PRNGKey(3, 4), rank=0, world_size=2)
    u = rex._reduced_row(np.array([-50.0, 20.0]), np.array([1000.0, 1200.0]))
    from oracle import chiron_oracle as O

    for r, (e, v) in enumerate([(-50.0, 1000.0), (20.0, 1200.0)]):
        for k in range(2):
            assert np.isclose(u[r, k], O.reduced_potential(e, 300.0 * 1.5 ** k, 1.0, v), rtol=1e-9)


class _FakeMCMC:
    """Stands in for MCMCSampler.run in the serial sampler's bookkeeping test (no kernels)."""

    def run(self, sampler_state, thermodynamic_state, n_iterations=1, nbr_list=None):
        return sampler_state, thermodynamic_state, nbr_list


def test_multistate_sampler_skeleton():
    from chiron_amd.mcmc import MCMCSampler, MoveSchedule, MonteCarloBarostatMove
    from chiron_amd.states import SamplerState

    K = 3
    pot = IdealGasPotential(Topology.homogeneous(4))
    thermo = [ThermodynamicState(pot, temperature=(300.0 + 50 * k) * unit.kelvin, pressure=1.0 * unit.atmosphere)
              for k in range(K)]
    box = unit.Quantity(np.diag([5.0, 5.0, 5.0]), unit.nanometer)
    samplers = [SamplerState(unit.Quantity(np.zeros((4, 3)), unit.nanometer), PRNGKey(7, k), box_vectors=box)
                for k in range(K)]
    ms = MultiStateSampler([_FakeMCMC() for _ in range(K)])
    with pytest.raises(RuntimeError):
        ms.create(thermo, samplers[:2], [None] * 2)
    ms.create(thermo, samplers, [None] * K)
    assert ms.iteration == 0 and ms.number_of_replicas == K and ms.number_of_thermodynamic_states == K
    assert ms._energy_thermodynamic_states.shape == (K, K) and ms._n_proposed_matrix.shape == (K, K)
    ms.run(4)
    assert ms.iteration == 4
    u_kn = ms._reporter.get_property("u_kn")
    assert u_kn.shape == (5, K, K)  # iteration 0 is logged as well (multistate.py:581-584)
    from oracle import chiron_oracle as O

    for k in range(K):
        assert np.isclose(u_kn[-1][k, 0], O.reduced_potential(0.0, 300.0 + 50 * k, 1.0, 125.0), rtol=1e-6)
    # reference behaviour: no swaps
    assert list(ms._replica_thermodynamic_states) == [0, 1, 2]
    ms2 = MultiStateSampler([_FakeMCMC() for _ in range(K)], swap="neighbors")
    ms2.create(thermo, samplers, [None] * K)
    ms2.run(3)
    assert sorted(ms2._replica_thermodynamic_states) == [0, 1, 2]


# ---------------------------------------------------------------------------
# bench.py as its own launcher (the driver's `python bench.py --gpus N` must start N ranks)
# ---------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                         timeout=timeout)
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    return res, lines


def test_bench_launcher_starts_n_ranks_gloo_dry_run():
    """`python bench.py --gpus 2` with no RANK in the environment launches two ranks itself
    (chiron/multistate.py:497-531 is the loop being sharded); the dry-run switch swaps RCCL for gloo
    and skips every kernel, so the launcher, the group and the exchange collective run on CPU."""
    res, lines = _run_bench(["--gpus", "2", "--steps", "3000", "--warmup", "0", "--dry-run-backend", "gloo"])
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1, res.stdout  # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] is None
    assert line["config"]["group_world_size"] == 2 and line["config"]["ranks"] == [0, 1]
    assert line["config"]["replica_exchanges"] == 3  # 3000 steps, one attempt per 1000
    assert sorted(line["config"]["state_of_replica"]) == [0, 1]


def test_bench_launcher_requalifies_slow_ranks_before_the_group_forms():
    """Rate-based rank requalification (round-3 review item 4): every rank reports its stand-alone rate before it joins
    the process group; the launcher replaces a rank below 0.97 x the median with a FRESH process (never a re-exec), at
    most twice, then releases all ranks.  Fake rates: rank 1 is slow on its first attempt (0.90), rank 3 on its first
    two (0.80, 0.95) -- both get replaced until they pass; rank 2 at 0.98 is inside the tolerance and stays."""
    fake = "1:0=0.90;3:0=0.80;3:1=0.95;2:0=0.98"
    res, lines = _run_bench(["--gpus", "4", "--steps", "2000", "--warmup", "0", "--dry-run-backend", "gloo"],
                            env_extra={"CHX_BENCH_FAKE_RATES": fake})
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["group_world_size"] == 4 and cfg["ranks"] == [0, 1, 2, 3]
    q = cfg["requalification"]
    assert sorted((r, a) for r, a, _ in q["replaced"]) == [(1, 0), (3, 0), (3, 1)]
    assert q["final_rates"] == {"0": 100000.0, "1": 100000.0, "2": 98000.0, "3": 100000.0}
    assert res.stderr.count("replaced by a fresh process") == 3
    # a rank that stays slow is kept after two replacements (the run must not hang on an unlucky host)
    res, lines = _run_bench(["--gpus", "3", "--steps", "1000", "--warmup", "0", "--dry-run-backend", "gloo"],
                            env_extra={"CHX_BENCH_FAKE_RATES": "2:0=0.5;2:1=0.5;2:2=0.5"})
    assert res.returncode == 0 and len(lines) == 1
    q = lines[0]["config"]["requalification"]
    assert [(r, a) for r, a, _ in q["replaced"]] == [(2, 0), (2, 1)] and q["final_rates"]["2"] == 50000.0
    # switched off: ranks start as they come, nothing is reported
    res, lines = _run_bench(["--gpus", "2", "--steps", "1000", "--warmup", "0", "--dry-run-backend", "gloo", "--requalify", "0"],
                            env_extra={"CHX_BENCH_FAKE_RATES": "1:0=0.5"})
    assert res.returncode == 0 and lines[0]["config"]["requalification"] is None


def test_bench_single_gpu_run_is_chosen_among_fresh_candidates():
    """`bench.py --gpus 1` started by hand: the process that runs the bench is chosen by the same rate-based
    qualification (bench.launch_single) -- two fresh candidates measure their stand-alone rate in turn and park; one
    below 0.97 x the better is replaced (at most twice); the earliest good one runs the bench, the others go.  Fake
    rates: a slow first candidate (0.85) -> a third is needed to have two that agree, and the second one runs."""
    args = ["--gpus", "1", "--steps", "500", "--warmup", "0", "--dry-run-backend", "gloo"]
    res, lines = _run_bench(args, env_extra={"CHX_BENCH_FAKE_RATES": "0:0=0.85"})
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["attempt"] == 1 and cfg["group_world_size"] == 1
    # (what the candidates that were NOT chosen read stands beside the chosen one's: round-5 review; the K-step figures are
    # measured on the GPU path only)
    assert cfg["requalification"] == {"ratio": 0.97, "candidates_steps_per_s": [85000.0, 100000.0, 100000.0], "chosen_attempt": 1,
                                      "chosen_rate_steps_per_s": 100000.0, "chosen_value_k_steps_per_s": None,
                                      "unselected_rate_steps_per_s": [85000.0, 100000.0],
                                      "unselected_value_k_steps_per_s": [None, None]}
    assert "running the bench in attempt 1" in res.stderr
    # two candidates that agree: the first runs, nothing is said
    res, lines = _run_bench(args, env_extra={"CHX_BENCH_FAKE_RATES": "0:1=0.99"})
    assert res.returncode == 0 and len(lines) == 1
    assert lines[0]["config"]["requalification"]["chosen_attempt"] == 0 and lines[0]["config"]["attempt"] == 0
    assert len(lines[0]["config"]["requalification"]["candidates_steps_per_s"]) == 2 and "running the bench in" not in res.stderr
    # an unlucky host (every candidate different): four candidates at most, the best one runs
    res, lines = _run_bench(args, env_extra={"CHX_BENCH_FAKE_RATES": "0:0=0.5;0:1=0.6;0:2=0.7;0:3=0.9"})
    assert res.returncode == 0 and len(lines) == 1
    assert lines[0]["config"]["requalification"]["chosen_attempt"] == 3
    # switched off, or under a profiler (whose preloaded tool has initialised the GPU: no child may be started): in-process
    for extra_args, extra_env in ((["--requalify", "0"], {}), ([], {"ROCPROFILER_TEST_MARKER": "1"})):
        res, lines = _run_bench(args + extra_args, env_extra=dict(extra_env, CHX_BENCH_FAKE_RATES="0:0=0.5"))
        assert res.returncode == 0 and len(lines) == 1 and lines[0]["config"]["requalification"] is None
    # a candidate that dies before it reports: its exit code comes back, no line
    res, lines = _run_bench(args, env_extra={"CHX_BENCH_FAKE_RATES": "0:0=not-a-number"})
    assert res.returncode != 0 and not lines


def test_requalify_decisions_on_fake_children():
    """The parent's decision rule by itself (bench.requalify): median of the ranks, 0.97 threshold, at most two
    replacements per rank, two ranks measured against the faster one, dead children (no rate) left alone."""
    import sys

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench

    class Child:
        def __init__(self, rank, attempt, table):
            self.rank, self.attempt, self.table, self.gone = rank, attempt, table, False
            self.rate = None

        def read_rate(self, timeout):
            self.rate = self.table.get((self.rank, self.attempt), 100.0)
            return self.rate

        def discard(self):
            self.gone = True

    def run(n, table):
        made = []

        def spawn(r, a):
            c = Child(r, a, table)
            made.append(c)
            return c

        first = [spawn(r, 0) for r in range(n)]
        final, log = bench.requalify(first, spawn)
        return first, final, made, log

    first, final, made, log = run(8, {(5, 0): 96.0, (5, 1): 99.0})
    assert [c.attempt for c in final] == [0, 0, 0, 0, 0, 1, 0, 0] and first[5].gone and not any(c.gone for c in final)
    assert len(made) == 9 and sum(e["replaced"] for e in log) == 1
    first, final, made, log = run(8, {(r, 0): 97.5 for r in range(8)})  # uniformly "slow": nobody is below the median
    assert len(made) == 8
    first, final, made, log = run(2, {(0, 0): 90.0})  # two ranks: the faster one is the yardstick
    assert final[0].attempt == 1 and final[1].attempt == 0
    first, final, made, log = run(4, {(2, 0): None})  # a child that never reported: not replaced here (it fails loudly later)
    assert len(made) == 4


def test_bench_launcher_eight_ranks_gloo_dry_run_and_torchrun():
    """The configuration the driver measures at round end, C5: eight ranks.  (a) `python bench.py --gpus 8` as its own
    launcher; (b) under the driver's own command line, `python -m torch.distributed.run --nproc-per-node 8 ... bench.py
    --gpus 8` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Dry run: gloo, no kernels -- launcher,
    group, barrier + max-over-ranks timing and the exchange collective with all eight ranks in it."""
    import json
    import socket
    import subprocess
    import sys

    res, lines = _run_bench(["--gpus", "8", "--steps", "4000", "--warmup", "0", "--dry-run-backend", "gloo"], timeout=400)
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1, res.stdout
    cfg = lines[0]["config"]
    assert lines[0]["n_gpus"] == 8 and cfg["group_world_size"] == 8 and cfg["ranks"] == list(range(8))
    assert cfg["replica_exchanges"] == 4 and sorted(cfg["state_of_replica"]) == list(range(8))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2000",
                          "--warmup", "0", "--dry-run-backend", "gloo"], env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["config"]["group_world_size"] == 8
    assert lines[0]["config"]["replica_exchanges"] == 2


def test_bench_rejects_world_size_mismatch_and_relays_failures():
    # under a launcher that started a different number of ranks than --gpus says: refuse, do not mislabel
    res, lines = _run_bench(["--gpus", "4", "--dry-run-backend", "gloo"],
                            {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert res.returncode != 0 and not lines and "WORLD_SIZE=1" in res.stderr
    # a rank that dies makes the launcher exit non-zero (no GPU here: the real path refuses to run on CPU)
    if not torch.cuda.is_available():
        res, lines = _run_bench(["--gpus", "2", "--steps", "10", "--warmup", "0"])
        assert res.returncode != 0 and not lines
        assert "launcher: rank" in res.stderr


def _skewed_worker(rank: int, world: int, port: int, out_dir: str):
    import time

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rex = ReplicaExchange(_ladder(world), PRNGKey(5, 6))
        t0 = time.perf_counter()
        for it in range(4):
            if rank == (it % world):
                time.sleep(0.25)  # one straggler per segment: the others must simply wait in the all-gather
            e = torch.tensor([-500.0 + 37.0 * ((rank * 3 + it) % 5)], dtype=torch.float32)
            rex.exchange(e)
        np.save(os.path.join(out_dir, f"sk_states_{rank}.npy"), rex.state_of_replica)
        np.save(os.path.join(out_dir, f"sk_ukn_{rank}.npy"), rex.u_kn)
        np.save(os.path.join(out_dir, f"sk_t_{rank}.npy"), np.array([time.perf_counter() - t0]))
    finally:
        dist.destroy_process_group()


def test_replica_exchange_four_ranks_with_stragglers(tmp_path):
    """World size 4 with a different rank sleeping in every segment: nobody deadlocks at the all-gather,
    all ranks end with the same permutation and the same fp64 energy matrix."""
    world = 4
    mp.spawn(_skewed_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    states = [np.load(tmp_path / f"sk_states_{r}.npy") for r in range(world)]
    ukn = [np.load(tmp_path / f"sk_ukn_{r}.npy") for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(states[0], states[r]) and np.array_equal(ukn[0], ukn[r])
    assert sorted(states[0]) == list(range(world))
    assert ukn[0].dtype == np.float64
    # every rank waited for the four stragglers (4 x 0.25 s), none ran ahead
    for r in range(world):
        assert float(np.load(tmp_path / f"sk_t_{r}.npy")[0]) >= 0.95


def test_gathered_energies_are_fp64_exact():
    rex = ReplicaExchange(_ladder(1), PRNGKey(1, 2), rank=0, world_size=1)
    v = 1234.56789012345
    got = rex.gather_energies(torch.tensor([v], dtype=torch.float64))
    assert got.dtype == np.float64 and got[0] == v
    e32 = np.float32(-7720.9896)
    assert rex.gather_energies(torch.tensor([e32], dtype=torch.float32))[0] == float(e32)


def test_bench_refuses_to_become_a_multi_rank_launcher_under_a_profiler():
    """Round-4 review: `rocprofv3 ... -- python3 bench.py --gpus N` (N > 1, no RANK) used to turn the profiled process --
    whose GPU the tool's preloaded library has already initialised -- into a launcher that starts N child interpreters.
    It must exit non-zero with advice instead, without starting anything (also for the CPU rehearsal mode)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ROCPROFILER_REGISTER_FORCE_LOAD"] = "1"  # (what a rocprofv3-started process finds in its environment)
    for extra in ([], ["--dry-run-backend", "gloo"]):
        out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"] + extra, cwd=root,
                             env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 2, (out.returncode, out.stderr[-1500:])
        assert "under a profiler" in out.stderr and "launcher" in out.stderr
        assert out.stdout.strip() == ""  # no rank ran, no JSON line


# ---------------------------------------------------------------------------------------------------------------------
# round 6: K thermodynamic states over W ranks (a rank propagates its K / W replicas side by side)
# ---------------------------------------------------------------------------------------------------------------------
def _worker_many(rank: int, world: int, port: int, out_dir: str, k_states: int):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rex = ReplicaExchange(_ladder(k_states), PRNGKey(11, 22))
        per = k_states // world
        assert rex.replicas_per_rank == per and list(rex.my_replicas) == list(range(rank * per, (rank + 1) * per))
        maps, scales = [], []
        for it in range(6):
            mine = torch.tensor([-100.0 * (r + 1) + 7.0 * it * ((r % 3) - 1) for r in rex.my_replicas], dtype=torch.float32)
            moves = rex.exchange_many(mine)
            assert [m[0] for m in moves] == rex.my_state_indices
            maps.append(rex.state_of_replica.copy())
            scales.append([m[1] for m in moves])
        np.save(os.path.join(out_dir, f"maps_{rank}.npy"), np.array(maps))
        np.save(os.path.join(out_dir, f"scales_{rank}.npy"), np.array(scales))
        np.save(os.path.join(out_dir, f"ukn_{rank}.npy"), rex.u_kn)
        with pytest.raises(RuntimeError):
            rex.exchange(torch.tensor([1.0]))  # (several replicas on this rank: the one-replica call refuses)
        with pytest.raises(ValueError):
            rex.gather_energies(torch.zeros(per + 1))
    finally:
        dist.destroy_process_group()


def test_replica_exchange_eight_states_over_two_ranks_gloo(tmp_path):
    """BASELINE's C5 has K = W = 8; with fewer cards a rank owns K / W replicas and contributes K / W energies to the one
    all-gather.  Two ranks, eight states: both ranks reach the same maps, which are the serial rule's on the gathered
    matrix; every rank's velocity scales are those of ITS replicas' moves."""
    world, K = 2, 8
    port = _free_port()
    mp.spawn(_worker_many, args=(world, port, str(tmp_path), K), nprocs=world, join=True)
    m0, m1 = np.load(tmp_path / "maps_0.npy"), np.load(tmp_path / "maps_1.npy")
    assert np.array_equal(m0, m1) and np.array_equal(np.load(tmp_path / "ukn_0.npy"), np.load(tmp_path / "ukn_1.npy"))
    betas = np.array([1.0 / (8.31446261815324e-3 * 300.0 * 1.5 ** r) for r in range(K)])
    cur = np.arange(K)
    key = PRNGKey(11, 22)
    temps = [300.0 * 1.5 ** r for r in range(K)]
    scales = np.concatenate([np.load(tmp_path / "scales_0.npy"), np.load(tmp_path / "scales_1.npy")], axis=1)
    swapped = 0
    for it in range(6):
        e = np.array([np.float32(-100.0 * (r + 1) + 7.0 * it * ((r % 3) - 1)) for r in range(K)], dtype=np.float64)
        u = e[:, None] * betas[None, :]
        new, _, _ = neighbor_swap_decisions(u, cur, it, key)
        assert sorted(new) == list(range(K)) and np.array_equal(new, m0[it])
        for r in range(K):
            want = np.sqrt(temps[new[r]] / temps[cur[r]]) if new[r] != cur[r] else 1.0
            assert np.isclose(scales[it][r], want)
            swapped += int(new[r] != cur[r])
        cur = new
    assert swapped >= 4  # (the colder replicas sit in the deeper basins: downhill swaps happen)


def test_batchable_samplers_and_fallbacks():
    """What MultiStateSampler may step side by side (chiron_amd.mcmc.MCMCSampler.batchable; no GPU involved): schedules of
    Langevin moves only, the same step counts in each, no reporters, one sampler per replica; anything else keeps the
    reference's serial loop.  ``batch_applies`` turns CPU tensors away, so on this box ``run_batch`` is the serial calls."""
    from chiron_amd.integrators import LangevinIntegrator, batch_applies
    from chiron_amd.mcmc import LangevinDynamicsMove, MCMCSampler, MonteCarloDisplacementMove, MoveSchedule
    from chiron_amd.states import SamplerState

    def sampler(steps=10, extra=None, reporter=None):
        sched = [("l", LangevinDynamicsMove(number_of_steps=steps, reporter=reporter))]
        if extra is not None:
            sched.append(("x", extra))
        return MCMCSampler(MoveSchedule(sched))

    assert MCMCSampler.batchable([sampler(), sampler(), sampler()])
    assert not MCMCSampler.batchable([sampler()])                      # one replica: nothing to batch
    same = sampler()
    assert not MCMCSampler.batchable([same, same])                     # one sampler object for two replicas
    assert not MCMCSampler.batchable([sampler(10), sampler(11)])       # different step counts
    assert not MCMCSampler.batchable([sampler(), sampler(extra=MonteCarloDisplacementMove(number_of_moves=2))])
    assert not MCMCSampler.batchable([sampler(extra=MonteCarloDisplacementMove(number_of_moves=2)) for _ in range(2)])
    assert not MCMCSampler.batchable([sampler(reporter=object()), sampler()])
    assert not MCMCSampler.batchable([_FakeMCMC(), _FakeMCMC()])
    pot = IdealGasPotential(Topology.homogeneous(4))
    box = unit.Quantity(np.diag([5.0, 5.0, 5.0]), unit.nanometer)
    states = [SamplerState(unit.Quantity(np.zeros((4, 3)), unit.nanometer), PRNGKey(7, k), box_vectors=box) for k in range(2)]
    thermo = [ThermodynamicState(pot, temperature=300.0 * unit.kelvin) for _ in range(2)]
    assert not batch_applies([LangevinIntegrator(), LangevinIntegrator()], states, thermo, [None, None], 5)
