#!/usr/bin/env python
"""Benchmark of the LJ pair-force / Langevin hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one BAOAB Langevin step (fused integrate kernel + pair-force kernel, with
neighbor-list rebuilds whenever the skin check fires) of a synthetic LJ fluid; the default
workload is BASELINE.json's metric configuration: N = 32,768 argon-like particles at
rho* = 0.1, Verlet list (cutoff 1.02 nm + skin 0.5 nm), 300 K, dt = 1 fs, fp32.  With
N > 1 GPUs every rank propagates its own replica (temperature ladder 300*1.05^r K) and the
replicas exchange labels every --exchange-interval steps through one RCCL all-gather
(weak scaling: value = total replica-steps per second).

Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel
(the fused step launch, algorithmic bytes 96 N + 16 P_nl per launch, HIP-event timed; the
pair-force kernels at C3 / C4 / liquid density and the batched step beside it) and a CPU
baseline (the C restatement of the reference algorithm, oracle/oracle.c).  On one GPU the
line also carries BASELINE's C5 shape on that one card (`c5_one_gpu`: eight replicas stepped
side by side, label exchange through a one-rank RCCL group).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

WORKLOADS = {
    # name: (N, reduced density)
    "c2": (512, 0.1),
    "c3": (32768, 0.1),
    "c4": (262144, 0.1),
    "c3_liquid": (32768, 0.8),
    "c4_liquid": (262144, 0.8),
    # intermediate sizes at the BASELINE density (tuning sweeps)
    "n8k": (8192, 0.1),
    "n16k": (16384, 0.1),
    "n64k": (65536, 0.1),
    "n128k": (131072, 0.1),
}
METRIC = "Langevin steps/sec + pair-force GB/s vs HBM roofline, LJ N=32k @1/8 GPU"
SIGMA_NM, EPS_KCAL, SKIN_NM = 0.34, 0.238, 0.5
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--build", default="cell", choices=["cell", "nsq"], help="neighbor-list build kernel")
    ap.add_argument("--chunk", type=int, default=32, help="steps enqueued per host round trip")
    ap.add_argument("--exchange-interval", type=int, default=1000, help="steps between replica-exchange attempts (1 ps at dt = 1 fs)")
    ap.add_argument("--kernel-reps", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=0,
                    help="repetitions of the contract's measurement (W untimed warm-up steps, then EXACTLY K timed steps between barrier + "
                         "synchronize); `value` is K / the median time, every sample is in config.value_samples_steps_per_s.  0 = auto: "
                         "5 when K <= 1000 (a single 200 us sample carries the host's jitter: the driver's K = 20 read 71.8 k in round 4 "
                         "where the same binary's 300-call median read 94.5 k), else 1")
    ap.add_argument("--cpu-steps", type=int, default=-1, help="steps of the CPU baseline (-1 = auto, 0 = skip)")
    ap.add_argument("--extra-c4", type=int, default=1, help="also report the N=262,144 pair-force roofline (1 GPU runs)")
    ap.add_argument("--c4-equilibration", type=int, default=20000,
                    help="steps (fs) the N=262,144 fluid is propagated from the lattice start before anything is quoted on it")
    ap.add_argument("--extra-liquid", type=int, default=1,
                    help="also report the liquid-density secondary workload of SURVEY 8(d) (rho* = 0.8, N = 32,768 and 262,144; 1 GPU runs)")
    ap.add_argument("--liquid-equilibration", type=int, default=4000)
    ap.add_argument("--extra-c2", type=int, default=1, help="also report C2 (N=512, all-pairs path) steps/s and pairs/s (1 GPU runs)")
    ap.add_argument("--extra-mc", type=int, default=1, help="also report Monte Carlo displacement trials/s at the workload (1 GPU runs)")
    ap.add_argument("--extra-overlap", type=int, default=1,
                    help="also report the aggregate rate of TWO independent replicas of the workload driven from two "
                         "threads / HIP streams on the one card (1 GPU runs; not the headline: one replica per GPU)")
    ap.add_argument("--extra-c5", type=int, default=1,
                    help="also run BASELINE's C5 shape on the ONE card (1 GPU runs): --c5-replicas replicas of the workload on a "
                         "temperature ladder, stepped side by side (one launch per step for all of them), label exchange "
                         "through a one-rank RCCL group")
    ap.add_argument("--c5-replicas", type=int, default=8)
    ap.add_argument("--c5-steps", type=int, default=4000, help="steps per timed window of the C5 leg (three windows)")
    ap.add_argument("--steady-steps", type=int, default=20000,
                    help="timed steps of the steady-state figure reported next to `value` (0 = skip)")
    ap.add_argument("--steady-warmup", type=int, default=2000)
    ap.add_argument("--clock-warmup-s", type=float, default=3.0,
                    help="seconds of untimed stepping in front of the --warmup steps: the card's graphics clock takes "
                         "0-2 s of load to leave its idle level (scripts/mode_probe.py), 10 %% in steps/s")
    ap.add_argument("--requalify", type=int, default=1,
                    help="launcher (--gpus N without a torchrun environment): every rank measures its stand-alone rate "
                         "before it joins the process group and the launcher replaces ranks below 0.97 x the median by "
                         "fresh processes (at most twice); with --gpus 1 the one process is chosen the same way among "
                         "fresh candidates; 0 = start the processes as they come")
    ap.add_argument("--qualify-seconds", type=float, default=1.0,
                    help="stand-alone stepping in front of the three rate windows of the requalification")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend of the real run (nccl = RCCL over xGMI; gloo only for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r uses device r mod device_count (RCCL "
                         "refuses two ranks on one device, so combine with --backend gloo); the line says shared_gpu")
    ap.add_argument("--dry-run-backend", default=None, choices=["gloo"],
                    help="launcher/plumbing rehearsal without a GPU: ranks form a gloo group, skip every kernel "
                         "and run the replica-exchange collective on synthetic energies (the line says dry_run)")
    return ap.parse_args()


QUALIFY_RATIO = 0.97      # a rank whose stand-alone rate is below this fraction of the ranks' median is replaced
QUALIFY_MAX_REPLACEMENTS = 2  # per rank


class _RankChild:
    """One rank of the launcher: a fresh child interpreter of this script plus the two pipes of the requalification
    protocol (child -> parent: its stand-alone rate; parent -> child: "go")."""

    def __init__(self, args, rank: int, port: int, attempt: int, qualify: bool, single: bool = False):
        import subprocess

        self.rank, self.attempt, self.rate, self.value_k = rank, attempt, None, None
        env = dict(os.environ)
        if single:  # (a candidate for the one process of `--gpus 1`: no rank environment, so no process group either)
            env.update(CHX_BENCH_CHILD="1", CHX_BENCH_ATTEMPT=str(attempt))
        else:
            env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CHX_BENCH_ATTEMPT=str(attempt))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this host driver
        fds = ()
        self.report_r = self.go_w = None
        if qualify:
            self.report_r, report_w = os.pipe()
            go_r, self.go_w = os.pipe()
            env["CHX_BENCH_QUALIFY_FDS"] = f"{report_w},{go_r}"
            fds = (report_w, go_r)
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, pass_fds=fds)
        for fd in fds:
            os.close(fd)

    def read_rate(self, timeout_s: float):
        """The child's stand-alone steps/s (one line on the report pipe), or None if it died / timed out."""
        import select

        buf = b""
        t_end = time.time() + timeout_s
        while time.time() < t_end and b"\n" not in buf:
            if self.proc.poll() is not None and not select.select([self.report_r], [], [], 0)[0]:
                return None
            if select.select([self.report_r], [], [], 0.1)[0]:
                chunk = os.read(self.report_r, 256)
                if not chunk:
                    return None
                buf += chunk
        try:
            fields = buf.split(b"\n")[0].split()
            self.rate = float(fields[0])
            self.value_k = float(fields[1]) if len(fields) > 1 else None  # (the driver's K-step figure in that process)
        except (ValueError, IndexError):
            self.rate = None
        return self.rate

    def release(self, summary: dict):
        try:
            os.write(self.go_w, (json.dumps(summary) + "\n").encode())
        except OSError:  # the child is gone (it failed before it parked): the launcher's wait loop reports its exit code
            pass
        for fd in (self.go_w, self.report_r):
            try:
                os.close(fd)
            except OSError:
                pass

    def discard(self):
        """Replace, never re-exec: the child is told to go away and a FRESH interpreter takes its rank."""
        self.proc.terminate()
        try:
            self.proc.wait(timeout=20)
        except Exception:  # pragma: no cover
            self.proc.kill()
        for fd in (self.go_w, self.report_r):
            try:
                os.close(fd)
            except OSError:
                pass


def requalify(children, spawn, ratio: float = QUALIFY_RATIO, max_replacements: int = QUALIFY_MAX_REPLACEMENTS,
              timeout_s: float = 300.0):
    """Rate-based rank requalification (round-3 review item 4).  One fresh process in eight launches 3-12 % slower for
    its whole life (the HIP runtime's slow launch regime, profiles/r03_launch_regime.md), and behind a barrier the
    slowest of N ranks sets the pace.  The regime shows in a rank's own stand-alone rate within a second, so every child
    measures that rate BEFORE it joins the process group and parks; the parent replaces any rank below `ratio` x the
    median of the ranks' rates with a fresh child (at most `max_replacements` times per rank), then releases all of them.
    `children`: list of objects with .rank / .read_rate(timeout) / .discard(); `spawn(rank, attempt)` makes a new one.
    Returns (children, log) with log = [{"rank", "attempt", "rate", "median", "replaced"}...]."""
    log = []
    rates = {c.rank: c.read_rate(timeout_s) for c in children}
    attempts = {c.rank: 0 for c in children}
    by_rank = {c.rank: c for c in children}
    for _ in range(max_replacements):
        known = sorted(v for v in rates.values() if v)
        if not known:
            break
        median = known[len(known) // 2] if len(known) > 2 else known[-1]  # (two ranks: the faster one is the yardstick)
        slow = [r for r, v in rates.items() if v and v < ratio * median and attempts[r] < max_replacements]
        for r, v in rates.items():
            log.append({"rank": r, "attempt": attempts[r], "rate": v, "median": median, "replaced": r in slow})
        if not slow:
            return [by_rank[r] for r in sorted(by_rank)], log
        for r in slow:
            by_rank[r].discard()
            attempts[r] += 1
            by_rank[r] = spawn(r, attempts[r])
        for r in slow:
            rates[r] = by_rank[r].read_rate(timeout_s)
    known = sorted(v for v in rates.values() if v)
    median = (known[len(known) // 2] if len(known) > 2 else known[-1]) if known else None
    for r, v in rates.items():
        log.append({"rank": r, "attempt": attempts[r], "rate": v, "median": median, "replaced": False})
    return [by_rank[r] for r in sorted(by_rank)], log


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` started by hand (no RANK in the environment): become a launcher.

    Starts N fresh child interpreters of this script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT as torchrun would set them), requalifies them by their stand-alone rate (`requalify`),
    relays their output -- rank 0 prints the JSON line -- and returns non-zero if any rank failed.  The launcher
    itself never initialises the GPU and no process that has touched the GPU is ever re-exec'ed: children are ordinary
    subprocesses, and a replaced rank is a NEW subprocess.  Under torchrun there is no parent of ours and the ranks run
    as they are started.  Replica loop being sharded: chiron/multistate.py:497-531.
    """
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    qualify = bool(args.requalify)
    spawn = lambda r, attempt: _RankChild(args, r, port, attempt, qualify)
    children = [spawn(r, 0) for r in range(args.gpus)]
    if qualify:
        children, qlog = requalify(children, spawn)
        replaced = [e for e in qlog if e["replaced"]]
        summary = {"ratio": QUALIFY_RATIO, "replaced": [[e["rank"], e["attempt"], e["rate"]] for e in replaced],
                   "final_rates": {str(c.rank): c.rate for c in children}}
        for r, a, v in summary["replaced"]:
            print(f"bench.py launcher: rank {r} (attempt {a}) ran {v:.0f} steps/s stand-alone, below {QUALIFY_RATIO} x the "
                  f"median: replaced by a fresh process", file=sys.stderr, flush=True)
        for c in children:
            c.release(summary)
    procs = [c.proc for c in children]
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0:
                print(f"bench.py launcher: rank {r} exited with code {code}", file=sys.stderr, flush=True)
                rc = rc or (code if code > 0 else 1)
                for q in pending.values():  # a dead rank leaves the others waiting in a collective
                    q.terminate()
        time.sleep(0.05)
    return rc


def under_a_profiler() -> bool:
    """rocprofv3 preloads a tool library that initialises the GPU in the process it starts -- from such a process no
    child interpreter may be started (the pool's exec guard), so the launcher modes stand aside."""
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCTRACER")) for k in os.environ):
        return True
    return any(tag in os.environ.get("LD_PRELOAD", "").lower() for tag in ("rocprof", "roctracer", "rocprofiler"))


def launch_single(args):
    """`python bench.py --gpus 1` started by hand: the one process that will run the bench is CHOSEN among fresh
    candidates by the rate-based qualification the multi-rank launcher applies to ranks (`requalify`): one fresh process
    in eight draws the HIP runtime's slow launch regime for its whole life (profiles/r03_launch_regime.md; the first
    process of profiles/r04's collection run read 69 k steps/s on the driver's command where the next ones read 85-88 k).
    Two candidates measure their stand-alone rate one after the other (never at the same time: they would share the
    card) and park; a candidate below 0.97 x the better one is replaced by a fresh one, at most twice; then the best one
    is released, runs the bench and prints the line, and the others are dismissed.  The launcher never touches the GPU,
    nothing is re-exec'ed.  Returns the exit code, or None when the protocol could not run (the caller then runs the
    bench in this very process, as before)."""
    spawn = lambda attempt: _RankChild(args, 0, 0, attempt, True, single=True)
    cands, log = [], []
    try:
        for attempt in range(2 + QUALIFY_MAX_REPLACEMENTS):
            c = spawn(attempt)
            rate = c.read_rate(300.0)
            log.append(rate)
            if rate is None:
                try:  # (a child whose report pipe closed is on its way out: give it a moment to be reaped)
                    code = c.proc.wait(timeout=10)
                except Exception:
                    code = None
                for o in cands + [c]:
                    o.discard()
                if code is None:  # alive but silent for five minutes: give up on the protocol, run in this process
                    print("bench.py launcher: a candidate did not report its rate; running the bench in the launcher's own "
                          "process", file=sys.stderr, flush=True)
                    return None
                return code if code else 1  # died before it could report: its exit code is the answer
            cands.append(c)
            best = max(o.rate for o in cands)
            if len(cands) >= 2 and sum(1 for o in cands if o.rate >= QUALIFY_RATIO * best) >= 2:
                break  # two processes agree on what this box gives: neither is a slow draw
        best = max(o.rate for o in cands)
        chosen = next(o for o in cands if o.rate >= QUALIFY_RATIO * best)  # (the earliest good one)
        for o in cands:
            if o is not chosen:
                o.discard()
        others = [o for o in cands if o is not chosen]
        summary = {"ratio": QUALIFY_RATIO, "candidates_steps_per_s": log, "chosen_attempt": chosen.attempt,
                   # (round-5 review: the process is picked by its stand-alone rate -- what the one NOT picked read, on the
                   # driver's own K-step figure as well, stands beside the headline)
                   "chosen_rate_steps_per_s": chosen.rate, "chosen_value_k_steps_per_s": chosen.value_k,
                   "unselected_rate_steps_per_s": [o.rate for o in others],
                   "unselected_value_k_steps_per_s": [o.value_k for o in others]}
        if len([r for r in log if r < QUALIFY_RATIO * best]):
            print(f"bench.py launcher: candidate rates {['%.0f' % r for r in log]} steps/s stand-alone; running the bench in "
                  f"attempt {chosen.attempt}", file=sys.stderr, flush=True)
        chosen.release(summary)
        return chosen.proc.wait()
    except BaseException:
        for o in cands:
            try:
                o.discard()
            except Exception:
                pass
        raise


def qualification_phase(measure):
    """Child side of `requalify`: report this process's stand-alone rate and park until the parent says go.  Returns the
    parent's summary (a dict), or None when this process was not started by the launcher with the protocol on.
    `measure()` -> steps/s; it runs BEFORE the process joins any process group."""
    spec = os.environ.get("CHX_BENCH_QUALIFY_FDS")
    if not spec:
        return None
    report_w, go_r = (int(v) for v in spec.split(","))
    rate = measure()
    if isinstance(rate, tuple):  # (stand-alone rate, the driver's K-step figure in this process)
        os.write(report_w, f"{rate[0]:.3f} {rate[1]:.3f}\n".encode())
    else:
        os.write(report_w, f"{rate:.3f}\n".encode())
    os.close(report_w)
    buf = b""
    while b"\n" not in buf:
        chunk = os.read(go_r, 4096)
        if not chunk:  # the parent went away
            raise SystemExit("bench.py: the launcher closed the requalification pipe")
        buf += chunk
    os.close(go_r)
    return json.loads(buf.split(b"\n")[0])


def make_system(workload: str, temperature_K: float, chunk: int, build: str):
    from chiron_amd import unit
    from chiron_amd.integrators import LangevinIntegrator
    from chiron_amd.neighbors import NeighborListCell, NeighborListNsqrd, OrthogonalPeriodicSpace
    from chiron_amd.potential import LJPotential
    from chiron_amd.states import SamplerState, ThermodynamicState
    from chiron_amd.testsystems import LennardJonesFluid
    from chiron_amd.utils import PRNG

    n, rho = WORKLOADS[workload]
    sigma = SIGMA_NM * unit.nanometer
    eps = EPS_KCAL * unit.kilocalories_per_mole
    cutoff = 3.0 * sigma
    fluid = LennardJonesFluid(n, reduced_density=rho, sigma=sigma, epsilon=eps, seed=1234)
    pot = LJPotential(fluid.topology, sigma=sigma, epsilon=eps, cutoff=cutoff)
    PRNG.set_seed(1234)
    state = SamplerState(fluid.positions, PRNG.get_random_key(), box_vectors=fluid.box_vectors)
    thermo = ThermodynamicState(pot, temperature=temperature_K * unit.kelvin)
    nbr, integ = make_list_and_integrator(workload, chunk, build)
    return fluid, pot, state, thermo, nbr, integ


def make_list_and_integrator(workload: str, chunk: int = 32, build: str = "cell"):
    """The neighbor list and the integrator of a workload as every leg of the bench uses them (also what the
    production-state parity tests start their branches with: tests/test_gpu_production_state.py)."""
    from chiron_amd import unit
    from chiron_amd.integrators import LangevinIntegrator
    from chiron_amd.neighbors import NeighborListCell, NeighborListNsqrd, OrthogonalPeriodicSpace

    _, rho = WORKLOADS[workload]
    cls = NeighborListCell if build == "cell" else NeighborListNsqrd
    m0 = 180 if rho <= 0.2 else 400
    nbr = cls(OrthogonalPeriodicSpace(), cutoff=3.0 * SIGMA_NM * unit.nanometer, skin=SKIN_NM * unit.nanometer, n_max_neighbors=m0)
    integ = LangevinIntegrator(timestep=1.0 * unit.femtoseconds, collision_rate=1.0 / unit.picoseconds, chunk_size=chunk)
    return nbr, integ


def two_replica_overlap(args, steps: int):
    """Two independent replicas of the benched workload (300 K and 315 K, their own lists, integrators and PRNG keys) on
    ONE card, each driven by a thread of its own on a HIP stream of its own: a step at this size is a latency chain
    that leaves the card mostly idle, so the two overlap.  Reported beside the headline, never as it (the BASELINE
    configuration is one replica per GPU)."""
    import threading

    import torch

    systems = [make_system(args.workload, t, args.chunk, args.build) for t in (300.0, 315.0)]
    streams = [torch.cuda.Stream() for _ in systems]
    rates = [0.0, 0.0]

    def drive(k: int, n_steps: int, record: bool):
        fluid, pot, state, thermo, nbr, integ = systems[k]
        with torch.cuda.stream(streams[k]):
            t0 = time.perf_counter()
            state, nbr = integ.run(state, thermo, number_of_steps=n_steps, nbr_list=nbr)
            streams[k].synchronize()
            if record:
                rates[k] = n_steps / (time.perf_counter() - t0)
        systems[k] = (fluid, pot, state, thermo, nbr, integ)

    def both(n_steps: int, record: bool):
        threads = [threading.Thread(target=drive, args=(k, n_steps, record)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()

    both(2000, False)  # lists, scratch, pinned slots
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    both(steps, True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"replicas": 2, "steps_each": steps, "aggregate_steps_per_s": round(2 * steps / wall, 2),
            "per_replica_steps_per_s": [round(r, 2) for r in rates],
            "note": "two threads, two HIP streams, one process, one card; not the headline configuration"}


def c5_one_gpu_block(args):
    """BASELINE's C5 -- K thermodynamic states of the metric's fluid, replica exchange -- on ONE card: the K replicas are
    propagated side by side (chiron_amd.integrators.run_batch: one launch per MD step for all of them; the reference's
    loop over replicas is serial, chiron/multistate.py:497-510) and exchange labels every --exchange-interval steps through
    the job's collective -- ``ReplicaExchange`` over a ONE-RANK RCCL group here (the very all_gather_into_tensor of fp64
    device tensors the eight-rank job makes over xGMI).  Reports the aggregate rate of three windows, the serial loop's rate
    on the same replicas, and the batched kernel event-timed with its roofline entry."""
    import socket

    import torch
    import torch.distributed as dist

    from chiron_amd import ops, unit
    from chiron_amd.integrators import _RunState, langevin_constants, run_batch
    from chiron_amd.multistate import ReplicaExchange
    from chiron_amd.rng import PRNGKey
    from chiron_amd.states import ThermodynamicState

    K = args.c5_replicas
    reps = []
    for k in range(K):
        fluid, pot, state, thermo, nbr, integ = make_system(args.workload, 300.0 * (1.05 ** k), args.chunk, args.build)
        integ.list_carry = False
        state.current_PRNG_key = PRNGKey(1234, k)
        nbr.set_replicas_sharing_card(K)  # (what MultiStateSampler tells its lists: lanes per particle for K x N in flight)
        reps.append([integ, state, thermo, nbr, pot, fluid])
    n = reps[0][5].n_particles
    own_group = False
    backend = None
    if not (dist.is_available() and dist.is_initialized()):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime

        # (a one-rank group: nothing to wait for but the local store; a minute is generous, and a failure here only costs
        # this leg -- the caller catches it and the line goes out without `c5_one_gpu`)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()),
                                timeout=datetime.timedelta(seconds=60))
        own_group = True
    try:
        backend = dist.get_backend()
        ladder = [ThermodynamicState(reps[0][4], temperature=300.0 * (1.05 ** k) * unit.kelvin) for k in range(K)]
        rex = ReplicaExchange(ladder, PRNGKey(2024, 7))
        rex.gather_energies(torch.zeros(K, dtype=torch.float64, device="cuda"))  # (communicator set-up: not in a window)

        def segment(n_steps, batched=True):
            thermo = rex.my_states
            if batched:
                out = run_batch([r[0] for r in reps], [r[1] for r in reps], thermo, n_steps, [r[3] for r in reps])
            else:
                out = [r[0].run(r[1], t, number_of_steps=n_steps, nbr_list=r[3]) for r, t in zip(reps, thermo)]
            for r, (st, nb) in zip(reps, out):
                r[1], r[3] = st, nb

        def exchange():
            e = torch.cat([r[4].compute_energy(r[1].positions, r[3]).reshape(1) for r in reps]).to(torch.float64)
            for r, (_, scale) in zip(reps, rex.exchange_many(e)):
                if scale != 1.0:
                    r[1].velocities = ops.scale(r[1].velocities, scale)

        def window(n_steps, batched=True):
            done = ex = 0
            t0 = time.perf_counter()
            while done < n_steps:
                seg = min(args.exchange_interval, n_steps - done)
                segment(seg, batched)
                done += seg
                if done % args.exchange_interval == 0:
                    exchange()
                    ex += 1
            torch.cuda.synchronize()
            return K * n_steps / (time.perf_counter() - t0), ex

        window(2 * args.exchange_interval)  # off the lattice, buffers, first exchange
        rates, exs = zip(*[window(args.c5_steps) for _ in range(3)])
        serial_rates = [window(args.c5_steps, batched=False)[0] for _ in range(2)]
        mid = sorted(rates)[1]
        # the batched launch alone, event-timed on the replicas as the windows left them (skin check off, every sample
        # from the same state: see time_kernels)
        sts = []
        for integ, state, thermo, nbr, pot, fluid in reps:
            half_dt, a, b = langevin_constants(integ.timestep, integ.collision_rate)
            nbr.build_from_state(state)
            from chiron_amd._device import as_tensor
            from chiron_amd.utils import get_list_of_mass

            mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
            kT = float((integ.kB * thermo.temperature).value_in_unit_system(unit.md_unit_system))
            st = _RunState(integ, pot, nbr, state.positions, state.velocities, mass, kT, half_dt, a, b, state.current_PRNG_key, None, "fused")
            st.params.skin = 1.0e6
            sts.append(st)
        step = [0]
        per_call = 32

        def one_chunk():
            ent = []
            for st in sts:
                even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
                ent.append((st.params, even, odd, step[0], per_call, 2))
            ops.langevin_batch_run(ent, 1 << 20)
            step[0] += per_call

        def fresh_state():
            for st, r in zip(sts, reps):
                st._load(r[1].positions, r[1].velocities, 0)
                st.params.skin = 1.0e6
            step[0] = 0
            one_chunk()
            one_chunk()

        samples = []
        for _ in range(7):
            fresh_state()
            ops.steps_join()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                one_chunk()
            ops.steps_join()
            e1.record()
            torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e-3 / (10 * per_call))
        t_k = sorted(samples)[3]
        p_nl = sum(int(torch.minimum(r[3].n_neighbors, torch.tensor(r[3].n_max_neighbors, device="cuda")).sum().item()) for r in reps)
        algorithmic = 96 * n * K + 16 * p_nl
        out = {
            "workload": f"{K} replicas of LJ fluid N={n} (temperature ladder 300 K x 1.05^k) on ONE MI355X, one launch per MD step for "
                        f"all of them (k_lj_step_batch), label exchange every {args.exchange_interval} steps",
            "replicas": K, "n_particles_each": n,
            "aggregate_replica_steps_per_s": round(mid, 1), "windows": [round(v, 1) for v in rates],
            "timed_steps_per_window": args.c5_steps, "exchanges_per_window": exs[0],
            "serial_loop_replica_steps_per_s": round(max(serial_rates), 1),
            "speedup_over_serial_loop": round(mid / max(serial_rates), 3),
            "collective_backend": backend, "group_world_size": dist.get_world_size(), "replicas_per_rank": rex.replicas_per_rank,
            "swap_acceptance": float(rex.n_accepted.sum() / max(1, rex.n_proposed.sum())),
            "state_of_replica": [int(v) for v in rex.state_of_replica],
            "kernel": f"k_lj_step_batch<{reps[0][3].force_list.g}, true> (the fused step of every replica in one grid: blockIdx.y = replica)",
            "lanes_per_particle": reps[0][3].force_list.g,
            "kernel_us": round(1e6 * t_k, 3), "kernel_us_samples": [round(1e6 * v, 3) for v in samples],
            "algorithmic_bytes": algorithmic, "list_pairs_all_replicas": p_nl,
            "achieved": round(algorithmic / t_k / 1e9, 2), "frac": round(algorithmic / t_k / 1e9 / HBM_PEAK_GBS, 5),
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "note": "frac by ALGORITHMIC bytes (96 N + 16 P_nl per replica, SURVEY 8d) like the one-replica step's: useful work per "
                    "second against the HBM peak, not measured traffic",
        }
        return out
    finally:
        if own_group:
            try:
                dist.destroy_process_group()
            except Exception:  # pragma: no cover
                pass


def read_sclk_mhz(device_index: int = 0):
    """Graphics clock of the card as rocm-smi shows it right now (MHz), or None.  Context for the reader of the JSON
    line: the same binary runs 152-159 k steps/s at the card's idle clock level (2.1-2.2 GHz) and 175-179 k at the
    loaded one (2.4 GHz), and the governor moves between the two on its own schedule (scripts/mode_probe.py)."""
    import re
    import shutil
    import subprocess

    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([exe, "--showclocks"], capture_output=True, text=True, timeout=15).stdout
    except Exception:  # pragma: no cover - diagnostics only
        return None
    for line in out.splitlines():
        m = re.match(rf"\s*GPU\[{device_index}\]\s*:\s*sclk clock level:\s*\S+\s*\((\d+)Mhz\)", line)
        if m:
            return int(m.group(1))
    return None


def time_kernels(fluid, pot, state, nbr, integ, reps: int, spin_s: float = 1.0, force_only: bool = False):
    """HIP-event timing (on the launch stream) of the two hot kernels on the current list:
    k_lj_step (fused pair force + BAOAB, what the timed region launches once per step) and
    k_lj_fl (the pair-force kernel alone, what compute_force() launches).
    Returns (t_step_s, t_force_s, P_nl, force_list); the samples behind the two times ride on force_list.timing_samples_us."""
    import ctypes as C

    import torch

    from chiron_amd import _lib, ops, unit
    from chiron_amd._device import as_tensor
    from chiron_amd.integrators import _RunState, langevin_constants
    from chiron_amd.utils import get_list_of_mass

    lib = _lib.load()
    fl = nbr.force_list
    n = fl.n
    box = nbr._box_h
    x4 = fl.pack(state.positions, 4)
    f3 = torch.empty((n, 3), dtype=torch.float32, device=x4.device)
    args = (ops._ptr(x4), ops._ptr(fl.fl), ops._ptr(fl.deg), n, fl.g, fl.chunks, box, C.c_float(pot.sigma),
            C.c_float(pot.epsilon), C.c_float(pot.cutoff), ops._ptr(f3), None, ops._ptr(fl.ws))

    force_launch = lambda: lib.chx_lj_fl_energy_force(*args, ops._stream())

    def spin(seconds: float):
        """Load the card with the (state-neutral) pair-force kernel: by the time the kernels are timed the host has
        been busy building systems and lists, and the graphics clock is back at its idle level (bench.py main: clock
        warm-up).  Spinning the STEP kernel instead would age the list: its skin check is disabled here, and a pruned
        list made from a stale full list is short and fast."""
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(200):
                force_launch()
            torch.cuda.synchronize()

    def timed(fn, count, before_each=None):
        """MEDIAN of seven event-timed repetitions of `count` launches each, and all seven (round-2 review: the headline
        statistic is the median; a disturbed sample -- a clock dip, a host that launches in bursts slower than the
        kernel runs -- shows in the list beside it)."""
        for _ in range(10):
            fn()
        samples = []
        for _ in range(7):
            if before_each is not None:
                before_each()
            ops.steps_join()  # (what before_each left on the step queue is not part of the sample)
            torch.cuda.synchronize()
            start = torch.cuda.Event(enable_timing=True)
            stop = torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(count):
                fn()
            # (the step launches run on the step queue paired with the stream -- chx_steps_join, include/chx.h -- which an
            # event on the stream does not see: the closing event is recorded once they have drained, so the bracket
            # holds the launches plus one hand-over each way, ~5 us over 640 launches)
            ops.steps_join()
            stop.record()
            torch.cuda.synchronize()
            samples.append(start.elapsed_time(stop) * 1e-3 / count)
        return sorted(samples)[len(samples) // 2], [round(1e6 * v, 3) for v in samples]

    spin(spin_s)
    t_force, force_samples = timed(force_launch, reps)
    if force_only:
        p_nl = int(torch.minimum(nbr.n_neighbors, torch.tensor(nbr.n_max_neighbors, device=x4.device)).sum().item())
        return None, t_force, p_nl, fl

    # the fused step kernel on the same state; a huge skin keeps the check from ever firing so that
    # every launch does the full work (the comparison itself still runs)
    half_dt, a, b = langevin_constants(integ.timestep, integ.collision_rate)
    mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
    v = state.velocities if state.velocities is not None else torch.zeros((n, 3), device=x4.device)
    st = _RunState(integ, pot, nbr, state.positions, v, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "fused")
    st.params.skin = 1.0e6
    step = [0]

    per_call = 32

    def one_chunk():
        # 32 launches per C call (exactly how the integrator enqueues them), so the Python call rate
        # does not bound the measurement; the inner-list refresh every 32 steps is part of the per-step figure
        even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
        ops.langevin_fused_chunk(st.params, even, odd, step[0], per_call, 2, 1)
        step[0] += per_call

    def fresh_state():
        # every sample starts from the state the list was built from: with its skin check off the list is never
        # rebuilt, and a pruned list made from an ever staler full list grows short and fast (N = 262 144: 12.9 us
        # in the first 640 launches, 10.9 in the seventh).  Two chunks put the first refresh of the inner list and
        # the filling of the launch queue in front of the timed launches.
        st._load(state.positions, v, 0)
        st.params.skin = 1.0e6
        step[0] = 0
        one_chunk()
        one_chunk()

    spin(0.5 * spin_s)
    # (20 chunks = 640 launches per sample at the default --kernel-reps: long enough that filling and draining the
    # launch queue does not show, short enough that the list does not age)
    t_step, step_samples = timed(one_chunk, max(2, reps // 10), before_each=fresh_state)
    t_step /= per_call
    p_nl = int(torch.minimum(nbr.n_neighbors, torch.tensor(nbr.n_max_neighbors, device=x4.device)).sum().item())
    fl.timing_samples_us = {"step": [round(v / per_call, 3) for v in step_samples], "force": force_samples,
                            "statistic": "median of 7 event-timed samples (per-launch averages over "
                                         f"{max(2, reps // 10) * per_call} step / {reps} force launches each"
                                         + ("; the step launches run on the step queue paired with the stream: the closing event "
                                            "is recorded after chx_steps_join, so a sample also holds one hand-over each way"
                                            if ops.step_queue_on() else "") + ")"}
    return t_step, t_force, p_nl, fl


def c2_block(chunk: int, steps: int = 20000):
    """BASELINE.json's C2 ("report for C2-C4 on 1 GPU"): LJ fluid N = 512 at rho* = 0.1 with the all-pairs
    structure (PairListNsqrd, PBC, chiron/neighbors.py:910-1289; workload of Examples/LJ_langevin.py:80-91) through
    LangevinIntegrator.run -- one k_ap_step launch per MD step.  Compute-bound by construction (24 N algorithmic
    bytes for N(N-1)/2 pair evaluations), so the kernel is quoted in pairs/s, not GB/s."""
    import torch

    from chiron_amd import ops, unit
    from chiron_amd._device import as_tensor
    from chiron_amd.integrators import _RunState, langevin_constants
    from chiron_amd.neighbors import OrthogonalPeriodicSpace, PairListNsqrd
    from chiron_amd.utils import get_list_of_mass

    fluid, pot, state, thermo, _, integ = make_system("c2", 300.0, chunk, "cell")
    n = fluid.n_particles
    pl = PairListNsqrd(OrthogonalPeriodicSpace(), cutoff=3.0 * SIGMA_NM * unit.nanometer)
    state, pl = integ.run(state, thermo, number_of_steps=2000, nbr_list=pl)  # warm-up, velocities, scratch
    torch.cuda.synchronize()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        state, pl = integ.run(state, thermo, number_of_steps=steps, nbr_list=pl)
        torch.cuda.synchronize()
        rates.append(steps / (time.perf_counter() - t0))
    # the kernel alone: back-to-back launches, 64 per C call, HIP events on the launch stream
    half_dt, a, b = langevin_constants(integ.timestep, integ.collision_rate)
    mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
    st = _RunState(integ, pot, pl, state.positions, state.velocities, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "pairs")
    step = [0]

    def one_chunk():
        even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
        ops.langevin_fused_chunk(st.params, even, odd, step[0], 64, 2, 1)
        step[0] += 64

    for _ in range(10):
        one_chunk()
    samples = []
    for _ in range(7):
        ops.steps_join()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            one_chunk()
        ops.steps_join()  # (see time_kernels)
        e1.record()
        torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) * 1e-3 / (20 * 64))
    t_k = sorted(samples)[3]
    pairs = n * (n - 1) // 2
    return {"workload": f"LJ fluid N={n} rho*=0.1, PairListNsqrd (all pairs, PBC) + LangevinIntegrator, one k_ap_step launch per step",
            "n_particles": n, "steps_per_s": round(sorted(rates)[1], 2), "steps_per_s_windows": [round(r, 2) for r in rates],
            "timed_steps_per_window": steps, "kernel": "k_ap_step<64> (all-pairs force + BAOAB + wrap, a wavefront per particle)",
            "kernel_us": round(1e6 * t_k, 3), "kernel_us_samples": [round(1e6 * v, 3) for v in samples],
            "pairs_per_launch": pairs, "pairs_per_s": round(pairs / t_k, 1), "algorithmic_bytes": (24 + 72) * n,
            "bound": "launch latency / fp32 VALU (compute-bound by construction: report pairs/s, SURVEY 8d)"}


def short_run_block(integ, thermo, state, nbr, k: int, calls: int = 200):
    """Consecutive run(k) calls, each continuing from the state the previous one returned: what a short call costs with
    the reference's list build on entry (chiron/integrators.py:169; what `value` is measured with) and with the list
    carried over from the previous call (the integrator's default for runs this short: the build on entry shrinks to
    "reference positions := x0", the caller's list is rebuilt only if it is read)."""
    import torch

    out = {"steps_per_call": k, "calls": calls, "statistic": "median wall time of one call between two device synchronisations"}
    keep = integ.list_carry
    try:
        for name, carry in (("list_build_on_entry", False), ("list_carried_over", True)):
            integ.list_carry = carry
            integ.n_carried_runs = 0
            state, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=nbr)
            torch.cuda.synchronize()
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                state, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=nbr)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            med = sorted(ts)[len(ts) // 2]
            out[name] = {"us_per_call": round(1e6 * med, 1), "steps_per_s": round(k / med, 1),
                         "calls_carried": int(getattr(integ, "n_carried_runs", 0))}
    finally:
        integ.list_carry = keep
    return state, nbr, out


def mc_block(args, trials: int = 4096):
    """The Metropolis half of the north star at the metric's configuration: MonteCarloDisplacementMove on the C3 fluid
    (chiron/mcmc.py:243-306, :357-463, :680-787) through the device-resident trial loop -- propose + skin check, energy
    over the force list, next proposal -- one launch per trial (k_mc_trial), no host round trip.
    The displacement is small enough (0.001 nm) that most trials are accepted and the list lasts; rebuild trials count."""
    import torch

    from chiron_amd import unit
    from chiron_amd.mcmc import MonteCarloDisplacementMove

    fluid, pot, state, thermo, nbr, integ = make_system(args.workload, 300.0, args.chunk, args.build)
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)  # off the lattice
    move = MonteCarloDisplacementMove(displacement_sigma=0.001 * unit.nanometer, number_of_moves=trials, report_interval=10 ** 9)
    nbr.build_from_state(state)
    state, thermo, nbr = move.update(state, thermo, nbr)  # warm-up: buffers, first chunk
    torch.cuda.synchronize()
    rates = []
    for _ in range(3):
        builds0 = nbr.n_builds
        t0 = time.perf_counter()
        state, thermo, nbr = move.update(state, thermo, nbr)
        torch.cuda.synchronize()
        rates.append(trials / (time.perf_counter() - t0))
    st = move.statistics
    return {"move": "MonteCarloDisplacementMove, sigma = 0.001 nm, all particles, device-resident trial loop (one launch per trial)",
            "n_particles": fluid.n_particles, "trials_per_s": round(sorted(rates)[1], 1), "trials_per_s_windows": [round(r, 1) for r in rates],
            "trials_per_window": trials, "acceptance": round(st["n_accepted"] / max(1, st["n_proposed"]), 4),
            "list_rebuilds_last_window": nbr.n_builds - builds0}


def roofline_entry(kind: str, n: int, p_nl: int, seconds: float, extra=None):
    """Algorithmic bytes per launch: pair force 24 N + 16 P_nl; the fused step adds the
    BAOAB traffic B_lang = 60 N (x, v, F in; x, v out) + 12 N (reference positions of the skin check)."""
    force_bytes = 24 * n + 16 * p_nl
    note = None
    if kind == "step":
        algorithmic = force_bytes + 72 * n
        name = "k_lj_step (fused pair force + BAOAB + wrap + skin check; one launch per MD step)"
        note = ("the launch walks the dynamically pruned inner list, while the algorithmic bytes are defined on the reference's "
                "list (SURVEY 8d: P_nl = its non-padding entries): read `frac` as useful work per second against the HBM "
                "peak, `hbm_GBps_measured` as the bandwidth the launch really draws")
    else:
        algorithmic = force_bytes
        name = "k_lj_fl (pair force over the neighbor list)"
    achieved = algorithmic / seconds / 1e9
    d = {
        "bound": "hbm",
        "kernel": name,
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": None,
        "hbm_frac_measured": None,  # (real HBM traffic / time / peak; filled when the PMC passes of this state are on file)
        "algorithmic_bytes": algorithmic,
        "kernel_us": round(seconds * 1e6, 3),
        "n_particles": n,
        "list_pairs": p_nl,
    }
    if note:
        d["note"] = note
    if extra:
        d.update(extra)
    # HBM traffic per launch from the committed PMC passes (profiles/pmc_traffic.json: separate rocprofv3 --pmc runs of
    # scripts/pmc_collect.sh, 2 x FETCH_SIZE + WRITE_SIZE with the gfx950 correction).  One system state per entry
    # (round-2 review): the passes run on the fluid equilibrated for `equilibration_steps` steps and record its P_nl;
    # the counters are quoted only when this run's list holds the same number of pairs to 2 %.
    wl = {32768: "c3", 262144: "c4"}.get(n)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh).get(f"{wl}:{'k_lj_step' if kind == 'step' else 'k_lj_fl'}")
        if pmc:
            pmc_pairs = pmc.get("list_pairs")
            same_layout = extra is None or extra.get("lanes_per_particle") in (None, pmc.get("lanes_per_particle"))
            if pmc_pairs and abs(pmc_pairs - p_nl) <= 0.02 * p_nl and same_layout:
                d["traffic"] = int(pmc["traffic_bytes"])
                d["hbm_GBps_measured"] = round(pmc["traffic_bytes"] / seconds / 1e9, 2)
                d["hbm_frac_measured"] = round(pmc["traffic_bytes"] / seconds / 1e9 / HBM_PEAK_GBS, 5)
                d["traffic_source"] = (f"{pmc.get('source', 'profiles/pmc_traffic.json')}: separate rocprofv3 --pmc passes over this "
                                       f"kernel on the same box geometry equilibrated for {pmc.get('equilibration_steps')} steps "
                                       f"(P_nl = {pmc_pairs}, this run {p_nl}); not collected in this run")
            else:
                d["traffic_note"] = (f"profiles/pmc_traffic.json was collected on a state with P_nl = {pmc_pairs} and "
                                     f"{pmc.get('lanes_per_particle')} lane(s) per particle; this run's list holds {p_nl} pairs"
                                     f"{'' if same_layout else ' in another lane layout'}: not the same state, traffic withheld")
    except (OSError, ValueError):
        pass
    return d


def secondary_block(args, wl: str, equilibration: int, window_steps: int, lattice_entry: bool = False):
    """Everything quoted on a secondary workload, on ONE system state: the fluid of `wl` propagated for `equilibration`
    steps from the jittered lattice (the lattice start has 17 % fewer listed pairs at rho* = 0.1 and runs 10 % faster:
    round-2 review), then the pair-force and fused-step kernels event-timed on the list it ended with, then three
    windows of `window_steps` Langevin steps including their list rebuilds."""
    import torch

    out = {}
    fluid, pot, state, thermo, nbr, integ = make_system(wl, 300.0, args.chunk, "cell")
    n = fluid.n_particles
    key = {"c4": "262k"}.get(wl)
    if lattice_entry:
        # (continuity with rounds 1-2, which quoted this kernel on the lattice start; named as such)
        nbr.build_from_state(state)
        _, tf0, p0, fl0 = time_kernels(fluid, pot, state, nbr, integ, args.kernel_reps, force_only=True)
        out[f"roofline_{key}_lattice_start"] = roofline_entry(
            "force", n, p0, tf0, {"lanes_per_particle": fl0.g, "system_state": "jittered-lattice start (what rounds 1-2 quoted; NOT the "
                                  "target figure: that one is taken on the equilibrated fluid)"})
    state, nbr = integ.run(state, thermo, number_of_steps=equilibration, nbr_list=nbr)
    torch.cuda.synchronize()
    ts, tf, p_nl, fl = time_kernels(fluid, pot, state, nbr, integ, args.kernel_reps)
    info = {"lanes_per_particle": fl.g, "max_degree": fl.maxdeg, "equilibration_steps": equilibration,
            "reduced_density": WORKLOADS[wl][1]}
    smp = getattr(fl, "timing_samples_us", None) or {}
    force = roofline_entry("force", n, p_nl, tf, dict(info, kernel_us_samples=smp.get("force"), statistic=smp.get("statistic")))
    step = roofline_entry("step", n, p_nl, ts, dict(info, kernel_us_samples=smp.get("step"), statistic=smp.get("statistic")))
    rates, rb = [], []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=window_steps, nbr_list=nbr)
        torch.cuda.synchronize()
        rates.append(round(window_steps / (time.perf_counter() - t0), 2))
        rb.append(integ.n_rebuilds)
    mid = sorted(rates)[1]
    if key:  # C4: the keys rounds 2-3 used, kept for continuity
        out[f"roofline_{key}"] = force
        out[f"roofline_{key}_step"] = step
        out[f"steps_per_s_{key}"] = mid
        out[f"steps_per_s_{key}_windows"] = rates
        out[f"rebuilds_{key}"] = rb[rates.index(mid)]
        return out
    return {"workload": f"LJ fluid N={n} rho*={WORKLOADS[wl][1]}, equilibrated {equilibration} steps", "pair_force": force, "step": step,
            "steps_per_s": mid, "steps_per_s_windows": rates, "timed_steps_per_window": window_steps,
            "rebuilds": rb[rates.index(mid)]}


def nest_rooflines(out: dict) -> None:
    """The driver's record keeps the scalar members of `roofline` and drops every other top-level key (round-3
    review: the north star's own target -- the N = 262 144 pair-force fraction -- was not in BENCH_r03.json's parsed
    record).  So every secondary kernel figure is ALSO stored inside `roofline`: as a nested entry and, because nested
    objects may be dropped too, as flat scalars `<entry>_frac / _kernel_us / _algorithmic_bytes / _list_pairs /
    _achieved`; likewise the headline rates inside `config`."""
    roof = out.get("roofline")
    if not isinstance(roof, dict):
        return
    entries = {"c3_pair_force": out.get("roofline_pair_force"), "c4_pair_force": out.get("roofline_262k"),
               "c4_step": out.get("roofline_262k_step"), "c4_pair_force_lattice_start": out.get("roofline_262k_lattice_start")}
    for wl, blk in (out.get("liquid_density") or {}).items():
        if isinstance(blk, dict) and "pair_force" in blk:
            entries[f"{wl}_pair_force"] = blk["pair_force"]
            entries[f"{wl}_step"] = blk["step"]
    for name, e in entries.items():
        if not isinstance(e, dict) or "frac" not in e:
            continue
        roof[name] = {k: e[k] for k in ("kernel", "frac", "achieved", "kernel_us", "algorithmic_bytes", "list_pairs", "n_particles",
                                        "traffic", "hbm_GBps_measured", "hbm_frac_measured", "lanes_per_particle",
                                        "equilibration_steps") if k in e}
        for k in ("frac", "achieved", "kernel_us", "algorithmic_bytes", "list_pairs", "hbm_frac_measured"):
            roof[f"{name}_{k}"] = e.get(k)
    c5 = out.get("c5_one_gpu")
    if isinstance(c5, dict) and "kernel_us" in c5:
        roof["c5_batched_step"] = {k: c5[k] for k in ("kernel", "frac", "achieved", "kernel_us", "algorithmic_bytes", "replicas",
                                                      "n_particles_each", "list_pairs_all_replicas")}
        for k in ("frac", "kernel_us", "achieved", "algorithmic_bytes"):
            roof[f"c5_batched_step_{k}"] = c5[k]
        out.setdefault("config", {})["c5_one_gpu_replica_steps_per_s"] = c5["aggregate_replica_steps_per_s"]
        out["config"]["c5_one_gpu_replicas"] = c5["replicas"]
        out["config"]["c5_one_gpu_collective_backend"] = c5["collective_backend"]
        out["config"]["c5_one_gpu_speedup_over_serial_loop"] = c5["speedup_over_serial_loop"]
    c2 = out.get("c2")
    if isinstance(c2, dict) and "kernel_us" in c2:
        roof["c2"] = {k: c2[k] for k in ("kernel", "kernel_us", "pairs_per_launch", "pairs_per_s", "steps_per_s", "algorithmic_bytes", "bound")}
        for k in ("kernel_us", "pairs_per_s", "steps_per_s"):
            roof[f"c2_{k}"] = c2[k]
    cfg = out.get("config", {})
    if isinstance(out.get("steady_state"), dict):
        cfg["steady_state_steps_per_s"] = out["steady_state"]["steps_per_s"]
        cfg["steady_state_rebuilds"] = out["steady_state"]["rebuilds"]
    if "steps_per_s_262k" in out:
        cfg["c4_steps_per_s"] = out["steps_per_s_262k"]
    for wl, blk in (out.get("liquid_density") or {}).items():
        if isinstance(blk, dict) and "steps_per_s" in blk:
            cfg[f"{wl}_steps_per_s"] = blk["steps_per_s"]
    sr = out.get("short_run")
    if isinstance(sr, dict) and "list_build_on_entry" in sr:
        cfg["short_run_steps_per_call"] = sr["steps_per_call"]
        cfg["short_run_us_build_on_entry"] = sr["list_build_on_entry"]["us_per_call"]
        cfg["short_run_us_list_carried"] = sr["list_carried_over"]["us_per_call"]
    mc = out.get("mc_displacement")
    if isinstance(mc, dict) and "trials_per_s" in mc:
        cfg["mc_trials_per_s"] = mc["trials_per_s"]


def cpu_baseline(workload: str, steps: int):
    """CPU restatement of the reference algorithm on this box's host cores, two variants:
    (i) faithful: O(N^2) build, distances over the padded (N,M) list, scattered pair forces, un-fused BAOAB,
        one thread (oracle/oracle.c: orc_langevin_run) -- N <= 32768 only;
    (ii) cell-list build + OpenMP over particles, bit-identical trajectories (orc_langevin_run_cell_omp),
        at the benched workload and at C4.  `value` is variant (ii) at the benched workload.
    steps <= 0: automatic sample sizes (a few seconds of CPU work each)."""
    from oracle import c_oracle as CO
    from oracle import chiron_oracle as O

    eps = EPS_KCAL * 4.184
    cores = min(CO.effective_cores(), 64)

    def system(wl):
        n, rho = WORKLOADS[wl]
        x, box = O.lj_lattice(n, rho, SIGMA_NM, seed=1234)
        mass = np.full(n, 39.948, np.float32)
        kT, a, b = O.langevin_constants(300.0, 0.001, 1.0)
        v0 = O.initial_velocities(mass, kT, O.philox_normals3(n, 0, O.STREAM_VELOCITY, (1234, 0)))
        return n, rho, x, box, mass, kT, a, b, v0

    def cell_omp(wl, n_steps):
        n, rho, x, box, mass, kT, a, b, v0 = system(wl)
        t0 = time.perf_counter()
        _, _, st = CO.langevin_run_cell_omp(x, v0, box, mass, SIGMA_NM, eps, 3 * SIGMA_NM, SKIN_NM, 0.001, a, b, kT, n_steps,
                                            None, (1234, 1), threads=cores)
        total = time.perf_counter() - t0
        return {"cores": st["threads"], "steps_per_s": round(n_steps / max(total - st["build_seconds"], 1e-9), 3),
                "steps": n_steps, "n_particles": n, "list_build_s": round(st["build_seconds"] / (1 + st["rebuilds"]), 4),
                "steps_per_s_including_builds": round(n_steps / total, 3)}

    n, rho = WORKLOADS[workload]
    out = {"unit": "steps/s", "kind": "port", "host_cores_visible": len(os.sched_getaffinity(0)), "host_cores_effective": cores}
    big = n > 32768
    main_steps = steps if steps > 0 else (60 if big else 300)
    out["cell_omp"] = cell_omp(workload, main_steps)
    out["value"] = out["cell_omp"]["steps_per_s"]
    out["cores"] = out["cell_omp"]["cores"]
    if not big:
        f_steps = steps if steps > 0 else 30
        _, _, x, box, mass, kT, a, b, v0 = system(workload)
        t0 = time.perf_counter()
        _, _, st = CO.langevin_run(x, v0, box, mass, SIGMA_NM, eps, 3 * SIGMA_NM, SKIN_NM, 180, 0.001, a, b, kT, f_steps,
                                   None, (1234, 1))
        total = time.perf_counter() - t0
        out["faithful_1t"] = {"cores": 1, "steps_per_s": round(f_steps / max(total - st["build_seconds"], 1e-9), 4),
                              "steps": f_steps, "list_build_s": round(st["build_seconds"] / (1 + st["rebuilds"]), 3)}
    else:
        out["faithful_1t"] = "infeasible: the O(N^2) build is 6.9e10 distance evaluations and the reference's own (N,N) mask 68.7 GB"
    if workload != "c4":
        out["c4_cell_omp"] = cell_omp("c4", steps if steps > 0 else 40)
    out["sample"] = (f"(ii) cell-list + OpenMP port, {out['cell_omp']['cores']} threads: {main_steps} BAOAB steps of the same N={n} "
                     f"rho*={rho} box after one O(N) list build (excluded from value, {out['cell_omp']['list_build_s']} s); "
                     "(i) faithful 1-thread port (O(N^2) build excluded) and the N=262144 figure are reported beside it; "
                     "C restatement of the reference algorithm (oracle/oracle.c), gcc -O2 -fopenmp -ffp-contract=off")
    return out


def dry_run(args, rank: int, world: int) -> None:
    """Launcher / process-group rehearsal on CPU (`--dry-run-backend gloo`): no kernel runs and nothing is
    measured; the ranks form the group, take the barrier + max-over-ranks timing path of the real run and push
    synthetic energies through the replica-exchange collective.  The printed line says `"dry_run": true`."""
    import torch
    import torch.distributed as dist

    from chiron_amd import unit
    from chiron_amd.multistate import ReplicaExchange
    from chiron_amd.rng import PRNGKey
    from chiron_amd.states import ThermodynamicState

    torch.set_num_threads(1)

    def fake_rate():
        # rehearsal of the requalification protocol: CHX_BENCH_FAKE_RATES="rank:attempt=relative;..." (default 1.0)
        table = dict(item.split("=") for item in os.environ.get("CHX_BENCH_FAKE_RATES", "").split(";") if "=" in item)
        return 100000.0 * float(table.get(f"{rank}:{os.environ.get('CHX_BENCH_ATTEMPT', '0')}", "1.0"))

    qual = qualification_phase(fake_rate)  # (before the process group exists: a replaced rank never joined anything)
    use_dist = world > 1 or "MASTER_ADDR" in os.environ
    if use_dist:
        dist.init_process_group(args.dry_run_backend, rank=rank, world_size=world)
    from chiron_amd.potential import IdealGasPotential
    from chiron_amd.topology import Topology

    pot = IdealGasPotential(Topology.homogeneous(4))
    ladder = [ThermodynamicState(pot, temperature=300.0 * (1.05 ** r) * unit.kelvin) for r in range(world)]
    rex = ReplicaExchange(ladder, PRNGKey(2024, 7), rank=rank, world_size=world)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    done = exchanges = 0
    while done < args.steps:
        seg = min(args.exchange_interval, args.steps - done)
        done += seg
        energy = torch.tensor([-1000.0 - 40.0 * rex.my_state_index + 7.0 * ((done // seg + rank) % 3)], dtype=torch.float64)
        rex.exchange(energy)
        exchanges += 1
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks = [rank]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, rank)
        ranks = sorted(gathered)
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": None, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "dry_run": True, "backend": args.dry_run_backend,
            "config": {"workload": "dry run: no kernels, synthetic energies", "replicas": world,
                       "group_world_size": dist.get_world_size() if use_dist else 1, "ranks": ranks,
                       "replica_exchanges": exchanges, "exchange_interval_steps": args.exchange_interval,
                       "state_of_replica": [int(v) for v in rex.state_of_replica],
                       "attempt": int(os.environ.get("CHX_BENCH_ATTEMPT", "0")), "requalification": qual},
        }), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        if under_a_profiler():
            # (the tool library the profiler preloads has already initialised the GPU in THIS process: it must not become
            # a launcher that starts child interpreters -- see under_a_profiler.  Round-4 review)
            print("bench.py: --gpus N > 1 under a profiler: this process has the GPU initialised by the tool's preloaded "
                  "library and may not start the ranks.  Profile one rank (`rocprofv3 ... -- python3 bench.py --gpus 1`), or "
                  "start the ranks with an outer launcher (torch.distributed.run) and put the profiler inside it.",
                  file=sys.stderr, flush=True)
            raise SystemExit(2)
        # started by hand: become the launcher BEFORE anything can touch the GPU
        raise SystemExit(launch_ranks(args))
    if ("RANK" not in os.environ and args.gpus == 1 and args.requalify and os.environ.get("CHX_BENCH_CHILD") != "1"
            and not under_a_profiler()):
        code = launch_single(args)  # (one GPU: the process that runs the bench is chosen among fresh candidates)
        if code is not None:
            raise SystemExit(code)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run_backend:
        return dry_run(args, rank, world)
    import torch

    # a few neighbouring CPUs for this process and the HIP runtime threads it is about to get (before the first GPU
    # call): chiron_amd._device.pin_process_cpus has the why; CHX_PIN_CPUS=0 turns it off
    from chiron_amd._device import pin_process_cpus

    all_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None  # (the CPU baseline gets them back)
    pinned_cpus = pin_process_cpus()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: no CUDA/ROCm device visible (there is no CPU fallback)")
    if args.share_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # The host side of this path only launches kernels.  torch's intra-op CPU pool (one thread per
    # visible core) must not wake up inside a CPU-quota'd container: the burst gets the whole
    # process throttled for tens of milliseconds (seen as 40-70 ms holes in the launch stream).
    torch.set_num_threads(1)
    import torch.distributed as dist

    # Requalification (launcher of this script only; `requalify`): this rank's stand-alone rate, measured BEFORE it
    # joins the process group -- a rank that has drawn the runtime's slow launch regime is replaced by a fresh process
    # by the parent while it is parked here, having joined nothing.
    prebuilt, qual = None, None
    if os.environ.get("CHX_BENCH_QUALIFY_FDS"):
        from chiron_amd import _lib as _lib0

        _lib0.load()
        prebuilt = list(make_system(args.workload, 300.0 * (1.05 ** rank), args.chunk, args.build))

        def measure():
            fluid0, pot0, st0, th0, nb0, ig0 = prebuilt
            ig0.list_carry = False
            t_q = time.perf_counter()
            while time.perf_counter() - t_q < args.qualify_seconds:
                st0, nb0 = ig0.run(st0, th0, number_of_steps=2000, nbr_list=nb0)
                torch.cuda.synchronize()
            rates_q = []
            for _ in range(3):
                t_w = time.perf_counter()
                st0, nb0 = ig0.run(st0, th0, number_of_steps=10000, nbr_list=nb0)
                torch.cuda.synchronize()
                rates_q.append(10000.0 / (time.perf_counter() - t_w))
            # ... and the driver's own figure in THIS process -- W warm-up steps, K timed steps, median of five -- so that the
            # line can say what the candidate that was not chosen would have reported (launcher of --gpus 1 only)
            vals_q = []
            if os.environ.get("CHX_BENCH_CHILD") == "1" and args.steps <= 1000:
                for _ in range(5):
                    st0, nb0 = ig0.run(st0, th0, number_of_steps=args.warmup, nbr_list=nb0)
                    torch.cuda.synchronize()
                    t_w = time.perf_counter()
                    st0, nb0 = ig0.run(st0, th0, number_of_steps=args.steps, nbr_list=nb0)
                    torch.cuda.synchronize()
                    vals_q.append(args.steps / (time.perf_counter() - t_w))
            prebuilt[2], prebuilt[4] = st0, nb0
            return (sorted(rates_q)[1], sorted(vals_q)[2]) if vals_q else sorted(rates_q)[1]

        qual = qualification_phase(measure)

    # under a launcher (RANK set) the process group is always created, also for one rank, so that the
    # RCCL path (init, barrier, all-reduce, all-gather) can be exercised on a single-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"  # where the timing collectives live
    exchange_on = world > 1 or (use_dist and os.environ.get("CHX_BENCH_EXCHANGE", "0") == "1")

    from chiron_amd import _lib, ops, unit
    from chiron_amd.multistate import ReplicaExchange
    from chiron_amd.rng import PRNGKey
    from chiron_amd.states import ThermodynamicState

    _lib.load()
    temperature = 300.0 * (1.05 ** rank)
    fluid, pot, state, thermo, nbr, integ = prebuilt or make_system(args.workload, temperature, args.chunk, args.build)
    n = fluid.n_particles
    ladder = [ThermodynamicState(pot, temperature=300.0 * (1.05 ** r) * unit.kelvin) for r in range(world)]
    rex = ReplicaExchange(ladder, PRNGKey(2024, 7), rank=rank, world_size=world)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    since_exchange = [0]  # steps since the last exchange attempt: the cadence carries across propagate() calls

    def propagate(st, n_steps):
        """n_steps Langevin steps; with several replicas, one exchange attempt every --exchange-interval steps of
        the replica's life (not one per call: a 20-step timed region must not be charged a whole exchange)."""
        nb = nbr
        done = 0
        exchanges = 0
        while done < n_steps:
            seg = min(args.exchange_interval - since_exchange[0], n_steps - done) if exchange_on else n_steps - done
            st, nb = integ.run(st, rex.my_state, number_of_steps=seg, nbr_list=nb)
            done += seg
            since_exchange[0] += seg
            if exchange_on and since_exchange[0] >= args.exchange_interval:
                energy = pot.compute_energy(st.positions, nb)
                _, vscale = rex.exchange(energy)
                if vscale != 1.0:
                    st.velocities = ops.scale(st.velocities, vscale)
                exchanges += 1
                since_exchange[0] = 0
        return st, exchanges

    def timed(st, n_steps):
        """barrier + sync, n_steps, sync [clock], barrier; elapsed = max over ranks of the per-rank times (gathered).
        The closing barrier brackets the region but is not charged to it: every rank stops its clock when ITS
        device has drained, and the slowest rank's time is the job's time."""
        builds0 = nbr.n_builds
        barrier()
        t0 = time.perf_counter()
        st, exchanges = propagate(st, n_steps)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        per_rank = [mine]
        if use_dist:
            dist.barrier()
            t = torch.tensor([mine], dtype=torch.float64, device=coll_dev)
            out = torch.empty(world, dtype=torch.float64, device=coll_dev)
            dist.all_gather_into_tensor(out, t)
            per_rank = [float(v) for v in out.cpu()]
        return st, max(per_rank), per_rank, exchanges, nbr.n_builds - builds0

    if use_dist:
        # communicator set-up and the first use of each collective are one-off costs (hundreds of ms with RCCL):
        # pay them here, never inside a timed region
        dist.barrier()
        warm = torch.zeros(1, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(warm, op=dist.ReduceOp.MAX)
        dist.all_gather_into_tensor(torch.empty(world, dtype=torch.float64, device=coll_dev), warm)
        rex.gather_energies(torch.zeros(1, dtype=torch.float64, device="cuda"))
        torch.cuda.synchronize()
    # `value`, the warm-up in front of it and the steady-state windows all run with the reference's list build on every
    # run() entry (chiron/integrators.py:169), as in rounds 1-2: the integrator's list carry-over for short calls is
    # measured separately (`short_run`)
    integ.list_carry = False
    # set-up (no steps): first list build, pinned read-back slots, device scratch
    integ.run(state, rex.my_state, number_of_steps=0, nbr_list=nbr)
    if exchange_on:
        # one complete exchange cycle on a throw-away copy of the exchange state: energy kernel, read-back,
        # all-gather, decisions -- every first-use cost of that path is paid here, and `rex` itself is untouched
        t_w = time.perf_counter()
        rex_warm = ReplicaExchange(ladder, PRNGKey(2024, 7), rank=rank, world_size=world)
        for _ in range(2):
            rex_warm.exchange(pot.compute_energy(state.positions, nbr))
        torch.cuda.synchronize()
        if os.environ.get("CHX_BENCH_DEBUG"):
            print(f"[bench] exchange warm-up: {1e3 * (time.perf_counter() - t_w):.2f} ms", file=sys.stderr, flush=True)
    # clock warm-up (untimed, no exchanges: a time-based loop must not contain collectives).  A process finds the card
    # at its idle graphics clock (rocm-smi: 2.1-2.2 GHz against 2.4 under load) and the same loop runs 152-159 k
    # steps/s instead of 175-177 k until the clock has come up, which takes anything from no time to 2 s of load
    # (scripts/mode_probe.py; round 1 and most of round 2 called this the box's "slow mode").  Production runs last
    # hours, so the figure of merit is the one at the loaded clock; the W warm-up steps follow as the contract says.
    clock_warm_s = 0.0
    if args.clock_warmup_s > 0:
        # at least --clock-warmup-s seconds, then until three consecutive 5000-step windows agree to 1.5 % and sit
        # within 3 % of the best window seen (the governor also sags for a few hundred ms now and then), at most 3x as long
        t_c = time.perf_counter()
        rates = []
        while True:
            t_w = time.perf_counter()
            state, _ = integ.run(state, rex.my_state, number_of_steps=5000, nbr_list=nbr)
            torch.cuda.synchronize()
            rates.append(5000.0 / (time.perf_counter() - t_w))
            spent = time.perf_counter() - t_c
            if spent < args.clock_warmup_s:
                continue
            last = rates[-3:]
            settled = len(last) == 3 and max(last) / min(last) < 1.015 and min(last) > 0.97 * max(rates)
            if settled or spent > 3.0 * args.clock_warmup_s:
                break
        clock_warm_s = time.perf_counter() - t_c
    # The contract's measurement -- W untimed warm-up steps, then exactly K steps between barrier + synchronize, max over
    # ranks -- `repeats` times over; the line reports the MEDIAN repetition (its time, its rebuild / refresh counts) and
    # every repetition's rate beside it (config.value_samples_steps_per_s, first sample first).
    repeats = args.repeats if args.repeats > 0 else (5 if args.steps <= 1000 else 1)
    reps_ = []
    for _ in range(repeats):
        state, _ = propagate(state, args.warmup)  # warm-up (untimed)
        prune_before = (getattr(integ, "prune_passes", 0), getattr(integ, "full_list_launches", 0))
        state, elapsed, per_rank, exchanges, rebuilds = timed(state, args.steps)
        prune_after = (getattr(integ, "prune_passes", 0), getattr(integ, "full_list_launches", 0))
        reps_.append((elapsed, per_rank, exchanges, rebuilds, prune_before, prune_after))
    value_samples = [round(world * args.steps / r[0], 2) for r in reps_]
    elapsed, per_rank, exchanges, rebuilds, prune_before, prune_after = sorted(reps_, key=lambda r: r[0])[len(reps_) // 2]
    value = world * args.steps / elapsed
    sclk = {}  # (read only after the steady-state windows: a rocm-smi call idles the card for half a second)

    # Steady-state figure (>= 1000 timed steps after >= 100 warm-up steps, rebuilds included): the
    # driver's `--steps` may be too short to hold a representative share of list rebuilds and inner-list
    # refreshes, so the same loop is timed once more over --steady-steps steps in this process.
    steady = None
    if args.steady_steps > 0:
        # (the median of three back-to-back windows, all three reported: before the clock warm-up above existed the
        # same binary showed 137-164 k steps/s from one process to the next)
        extra_warm = max(0, args.steady_warmup - args.warmup - args.steps)
        state, _ = propagate(state, extra_warm)
        windows = []
        for _ in range(3):
            state, s_el, _, s_ex, s_rb = timed(state, args.steady_steps)
            windows.append({"steps_per_s": round(world * args.steady_steps / s_el, 2), "rebuilds": s_rb,
                            "seconds": round(s_el, 6), "replica_exchanges": s_ex})
        mid = sorted(windows, key=lambda w: w["steps_per_s"])[1]
        steady = {"steps": args.steady_steps, "warmup": args.warmup + args.steps + extra_warm,
                  "steps_per_s": mid["steps_per_s"], "rebuilds": mid["rebuilds"], "seconds": mid["seconds"],
                  "windows": windows, "statistic": "median of 3 windows"}

    if rank == 0:
        sclk["after_steady_state"] = read_sclk_mhz(local_rank)
        # hot kernels, timed live on the same device / list state
        t_step, t_force, p_nl, fl = time_kernels(fluid, pot, state, nbr, integ, args.kernel_reps)
        sclk["after_kernel_timing"] = read_sclk_mhz(local_rank)
        info = {"lanes_per_particle": fl.g, "max_degree": fl.maxdeg,
                "system_state": "the replica as the timed regions left it (equilibrated; list from its last rebuild)"}
        ts_ = getattr(fl, "timing_samples_us", None) or {}
        roof = roofline_entry("step", n, p_nl, t_step, dict(info, kernel_us_samples=ts_.get("step"), statistic=ts_.get("statistic")))
        roof_force = roofline_entry("force", n, p_nl, t_force, dict(info, kernel_us_samples=ts_.get("force"), statistic=ts_.get("statistic")))
        out = {
            "metric": METRIC,
            "value": round(value, 2),
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"LJ fluid N={n} rho*={WORKLOADS[args.workload][1]} (jittered lattice, seed 1234), "
                            f"sigma=0.34 nm eps=0.238 kcal/mol cutoff=1.02 nm skin=0.5 nm, "
                            f"NeighborList{'Cell' if args.build == 'cell' else 'Nsqrd'} + LangevinIntegrator "
                            f"(BAOAB, dt=1 fs, gamma=1/ps, 300 K)",
                "n_particles": n,
                "replicas": world,
                "parallelism": f"{world} independent replica(s), one per GPU" + (
                    f", label exchange every {args.exchange_interval} steps via one "
                    f"{'RCCL' if args.backend == 'nccl' else args.backend} all-gather" if world > 1 else ""),
                "group_world_size": dist.get_world_size() if use_dist else 1,
                "collective_backend": (dist.get_backend() if use_dist else None),
                "shared_gpu": bool(args.share_gpu),
                "per_rank_steps_per_s": [round(args.steps / t, 2) for t in per_rank],
                "chunk_steps": args.chunk,
                "launch_window": integ.launch_window,
                "pinned_cpus": pinned_cpus,
                "clock_warmup_s": round(clock_warm_s, 2),
                "sclk_mhz": sclk,
                "list_rebuilds_in_timed_region": rebuilds,
                # dynamically pruned inner list (chiron_amd/integrators.py): same interacting pairs
                "pruning_passes_in_timed_region": prune_after[0] - prune_before[0],
                "launches_on_full_list_in_timed_region": prune_after[1] - prune_before[1],
                "replica_exchanges": exchanges,
                "exchange_interval_steps": (args.exchange_interval if exchange_on else None),
                "swap_acceptance": (float(rex.n_accepted.sum() / max(1, rex.n_proposed.sum())) if world > 1 else None),
            },
            "steady_state": steady,
            "roofline": roof,
            "roofline_pair_force": roof_force,
        }
        out["config"]["list_carry_over"] = False
        out["config"]["value_statistic"] = (f"median of {repeats} repetitions of [{args.warmup} warm-up steps, then {args.steps} timed "
                                            "steps between barrier + synchronize]" if repeats > 1 else "one timed region")
        out["config"]["value_samples_steps_per_s"] = value_samples
        out["config"]["value_first_sample_steps_per_s"] = value_samples[0]
        try:
            sq = ops.step_queue_info()
            # (how the step launches reach the device: include/chx.h, "The step queue")
            out["config"]["step_queue"] = ("AQL packets on a private HSA queue, steady-state packets without cache fences"
                                           if sq["on"] and sq["hot_release"] == "none" else
                                           ("AQL packets on a private HSA queue, agent-scope fences" if sq["on"] else
                                            "off: hipLaunchKernelGGL (" + (sq["status"] or "CHX_AQL=0") + ")"))
        except Exception as exc:  # pragma: no cover - diagnostics only
            out["config"]["step_queue"] = repr(exc)
        from chiron_amd import integrators as _integ_mod

        # (what a list build inside the timed run() writes: the force list the step kernels read, every time; the
        # reference-format half rows on demand -- their counts always, the rows when `neighbor_list` is first read)
        out["config"]["half_rows_inside_run"] = "written" if _integ_mod._RUN_HALF_ROWS else "counted; built on first read of neighbor_list"
        out["config"]["requalification"] = qual
        if isinstance(qual, dict) and "unselected_rate_steps_per_s" in qual:
            # (scalars: the driver's record keeps scalar members only)
            un_r = [v for v in qual["unselected_rate_steps_per_s"] if v]
            un_v = [v for v in qual["unselected_value_k_steps_per_s"] if v]
            out["config"]["unselected_candidate_rate_steps_per_s"] = max(un_r) if un_r else None
            out["config"]["unselected_candidate_value_steps_per_s"] = max(un_v) if un_v else None
            out["config"]["chosen_candidate_rate_steps_per_s"] = qual.get("chosen_rate_steps_per_s")
            out["config"]["chosen_candidate_prelim_value_steps_per_s"] = qual.get("chosen_value_k_steps_per_s")
        if world == 1:
            try:
                state, nbr, out["short_run"] = short_run_block(integ, rex.my_state, state, nbr, args.steps if args.steps <= 64 else 20)
            except Exception as exc:  # pragma: no cover - diagnostics only
                out["short_run"] = {"error": repr(exc)}
        if world == 1 and args.extra_overlap:
            try:
                out["two_replicas_one_gpu"] = two_replica_overlap(args, args.steps if args.steps >= 2000 else 20000)
            except Exception as exc:  # pragma: no cover - diagnostics only
                out["two_replicas_one_gpu"] = {"error": repr(exc)}
        if world == 1 and args.extra_c5 and not use_dist:
            try:
                out["c5_one_gpu"] = c5_one_gpu_block(args)
            except Exception as exc:  # pragma: no cover - diagnostics only
                out["c5_one_gpu"] = {"error": repr(exc)}
        if world == 1 and args.extra_mc:
            try:
                out["mc_displacement"] = mc_block(args)
            except Exception as exc:  # pragma: no cover - diagnostics only
                out["mc_displacement"] = {"error": repr(exc)}
        if world == 1 and args.extra_c2 and args.workload != "c2":
            try:
                out["c2"] = c2_block(args.chunk)
            except Exception as exc:  # pragma: no cover - diagnostics only
                out["c2"] = {"error": repr(exc)}
        if world == 1 and args.extra_c4 and args.workload != "c4":
            try:
                out.update(secondary_block(args, "c4", args.c4_equilibration, 4000, lattice_entry=True))
            except Exception as exc:  # pragma: no cover - diagnostics only
                out["roofline_262k"] = {"error": repr(exc)}
        if world == 1 and args.extra_liquid:
            # SURVEY 8(d)'s secondary workload: the reference's own default fluid density (rho* = 0.8,
            # chiron/tests/conftest.py:48; tests/test_testsystems.py:168-219), ~150 half-list entries per particle --
            # the one state where the list kernels are throughput-bound and the roofline fraction means bandwidth
            liquid = {}
            for wl, eq, window in (("c3_liquid", args.liquid_equilibration, 4000), ("c4_liquid", max(500, args.liquid_equilibration // 4), 800)):
                try:
                    liquid[wl] = secondary_block(args, wl, eq, window)
                except Exception as exc:  # pragma: no cover - diagnostics only
                    liquid[wl] = {"error": repr(exc)}
            out["liquid_density"] = liquid
        nest_rooflines(out)
        if world == 1 and args.cpu_steps != 0:
            # the CPU restatement runs on every core the process was given, not on the few the GPU loop is pinned to:
            # its OpenMP threads are created inside this call and inherit the mask in force here
            mask_now = os.sched_getaffinity(0) if all_cpus is not None else None
            try:
                if all_cpus is not None and pinned_cpus:
                    os.sched_setaffinity(0, all_cpus)
                out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_steps)
            except Exception as exc:  # pragma: no cover
                out["cpu_baseline"] = {"error": repr(exc)}
            finally:
                if mask_now is not None:
                    try:
                        os.sched_setaffinity(0, mask_now)
                    except OSError:  # pragma: no cover
                        pass
        try:
            # the assumption the fence-free packets rest on, checked again now that the process has shared the card with
            # other legs' queues for minutes (the queue's first use checked it once): stale words read, 0 = holds
            out["config"]["fence_selfcheck_at_end"] = ops.step_queue_selfcheck()
            out["config"]["step_queue_status_at_end"] = ops.step_queue_info()["status"] or "ok"
        except Exception as exc:  # pragma: no cover - diagnostics only
            out["config"]["fence_selfcheck_at_end"] = repr(exc)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
