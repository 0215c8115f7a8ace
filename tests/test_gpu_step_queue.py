"""The step queue (include/chx.h: chx_steps_join; csrc/chx_aql.hip): the fused step launches dispatched as AQL packets
on a private HSA queue, steady-state packets without cache fences.  It must not change one bit of a trajectory."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_TRAJECTORY = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import torch
import bench
from chiron_amd import ops
fluid, pot, state, thermo, nbr, integ = bench.make_system(sys.argv[1], 300.0, 32, "cell")
rebuilds = 0
for k in (int(sys.argv[2]), 20, 7, 100, 1):   # a long run with rebuilds and refreshes, then short ones (carried lists)
    state, nbr = integ.run(state, thermo, number_of_steps=k, nbr_list=nbr)
    rebuilds += integ.n_rebuilds
torch.cuda.synchronize()
h = hashlib.sha256(state.positions.cpu().numpy().tobytes() + state.velocities.cpu().numpy().tobytes()
                   + nbr.ref_positions.cpu().numpy().tobytes()).hexdigest()
print("RESULT " + json.dumps({"hash": h, "rebuilds": rebuilds, "queue": ops.step_queue_info(),
                              "carried": getattr(integ, "n_carried_runs", 0), "idle_packets": ops.queue_debug_spins(),
                              "fence_checks": int(ops._lib.load().chx_step_queue_selfcheck_runs()),
                              "status": ops.step_queue_info()["status"]}))
""" % ROOT


def _run(env_extra, workload, steps):
    env = dict(os.environ)
    for k in ("CHX_AQL", "CHX_AQL_FENCE", "CHX_BUILD_SUB", "CHX_QUEUE_DEBUG", "CHX_AQL_RECHECK_EVERY"):
        env.pop(k, None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", _TRAJECTORY, workload, str(steps)], capture_output=True, text=True, timeout=600,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("workload,steps", [("c3", 3000), ("c4", 600)], ids=["c3-32768", "c4-262144"])
def test_step_queue_and_its_fence_modes_leave_every_bit_of_the_trajectory(workload, steps):
    """(Four fresh processes per workload.)  One trajectory from the bench's own start -- thousands of steps with list
    rebuilds and inner-list refreshes, then short runs on carried-over lists -- in fresh processes: plain HIP launches
    (CHX_AQL=0), the step queue with HIP's agent-scope fences on every packet (CHX_AQL_FENCE=agent), and the default
    (steady-state packets carry no fences: their outputs are write-through stores, and what they read was written through
    or released).  A single stale read anywhere -- an L1, scalar-cache or L2 line that survived a dispatch, a store still
    dirty in one XCD's L2 -- changes the hash.  (Round 6 retired the queue's narrower modes -- steps only, gate at the first
    step, the join that always sends its own barrier packet -- together with their switches.)"""
    hip = _run({"CHX_AQL": "0"}, workload, steps)
    fenced = _run({"CHX_AQL_FENCE": "agent"}, workload, steps)
    bare = _run({}, workload, steps)
    # ... and -- not a queue mode, but the same question -- the rebuilds' general row kernels instead of the filtered
    # sixteen-lanes-per-row one: every force list of the trajectory, on the clustered fluid, must come out the same bits or
    # the summation order of some force changes
    old_rows = _run({"CHX_BUILD_SUB": "0"}, workload, steps)
    assert old_rows["hash"] == bare["hash"], (old_rows, bare)
    assert old_rows["rebuilds"] == bare["rebuilds"]
    opened = lambda r: r["queue"]["gates"] + r["queue"]["idle_opens"]  # (through a marker launch / on an idle stream)
    assert not hip["queue"]["on"] and hip["queue"]["dispatched"] == 0
    assert fenced["queue"]["on"] and fenced["queue"]["hot_release"] == "agent"
    assert bare["queue"]["on"] and bare["queue"]["hot_release"] == "none", bare["queue"]
    total = steps + 20 + 7 + 100 + 1
    assert bare["queue"]["dispatched"] >= total and opened(bare) >= 5 and not bare["queue"]["open"]
    assert hip["rebuilds"] > 0 or workload == "c4"
    assert hip["hash"] == fenced["hash"] == bare["hash"], (hip, fenced, bare)
    assert hip["rebuilds"] == fenced["rebuilds"] == bare["rebuilds"] and hip["carried"] == bare["carried"]


def test_step_queue_orders_with_the_stream_it_is_paired_with():
    """chx_steps_join / the gate: a kernel enqueued on the stream BEFORE step launches is seen by them, and work the
    caller enqueues AFTER a join sees what they wrote -- with torch's own ops on both sides, which never pass through
    libchx (include/chx.h: only the caller's own work needs an explicit join)."""
    sys.path.insert(0, ROOT)
    import bench
    from chiron_amd import ops

    fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
    if not ops.step_queue_on():
        pytest.skip("the step queue is switched off (CHX_AQL=0)")
    state, nbr = integ.run(state, thermo, number_of_steps=50, nbr_list=nbr)
    info0 = ops.step_queue_info()
    # a run() returns with the queue drained: reading its results needs no join
    assert not info0["open"] and info0["dispatched"] >= 51
    x1 = state.positions.clone()
    # torch work in front (a big fill the steps' inputs do not depend on, then the inputs themselves rewritten): the
    # gate must hold the first step packet until both have run
    junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    for k in range(8):  # (a millisecond of work on the stream: the run() below finds it busy and must put a gate up)
        junk.fill_(float(k))
    state2, nbr = integ.run(state, thermo, number_of_steps=30, nbr_list=nbr)
    assert float(junk[-1].item()) == 7.0
    info1 = ops.step_queue_info()
    assert info1["gates"] > info0["gates"] and info1["joins"] > info0["joins"] and not info1["open"]
    # ... and a run() that finds the stream idle opens the queue without a marker launch
    torch.cuda.synchronize()
    state3, nbr = integ.run(state2, thermo, number_of_steps=30, nbr_list=nbr)
    info2 = ops.step_queue_info()
    assert info2["idle_opens"] > info1["idle_opens"] and info2["gates"] == info1["gates"]
    assert torch.isfinite(state3.positions).all()
    assert torch.isfinite(state2.positions).all() and not torch.equal(state2.positions, x1)


def test_runs_without_a_verlet_list_return_with_the_step_queue_drained():
    """The all-pairs paths (nbr_list=None, PairListNsqrd: k_ap_step) ride on the step queue as well; torch reads the
    returned state on its own stream, which no longer orders behind them by itself: run() joins before it returns
    (include/chx.h: the step queue).  Round 5 shipped this path without the join for a while: the suite passed on
    timing, until a faster build kernel elsewhere shifted it."""
    sys.path.insert(0, ROOT)
    import bench
    from chiron_amd import ops, neighbors, unit as u

    fluid, pot, state, thermo, nbr, integ = bench.make_system("c2", 300.0, 32, "cell")
    if not ops.step_queue_on():
        pytest.skip("the step queue is switched off (CHX_AQL=0)")
    for lst in (None, neighbors.PairListNsqrd(neighbors.OrthogonalPeriodicSpace(), cutoff=3.0 * bench.SIGMA_NM * u.nanometer)):
        for steps in (1, 40):
            before = ops.step_queue_info()
            out, _ = integ.run(state, thermo, number_of_steps=steps, nbr_list=lst)
            info = ops.step_queue_info()
            assert not info["open"], (lst, steps, info)
            assert info["dispatched"] > before["dispatched"] and info["joins"] > before["joins"]
            x = out.positions.cpu()
            assert torch.isfinite(x).all()
            if lst is not None:
                box = state.box_vectors.cpu()
                assert (x >= 0).all() and (x <= box[0, 0]).all()  # (wrapped every step: the final copy has run)


def test_a_run_state_dies_with_its_call():
    """A run()'s internal state (force list, position buffers' views, closures handed to it for the host work it does
    while launches are queued) must be freed by reference counting when the call returns -- not wait for the cycle
    collector: at N = 262 144 and liquid density a force list is half a gigabyte, and four of them parked in cycles sent
    torch's allocator back to hipMalloc (8 ms each) for every rebuild of the next calls (round 5, found by bench.py's
    liquid leg reading 13.5 k steps/s in two windows of three and 19 k in the third)."""
    import gc

    sys.path.insert(0, ROOT)
    import bench
    from chiron_amd import integrators as I

    fluid, pot, state, thermo, nbr, integ = bench.make_system("c2", 300.0, 32, "cell")
    state, nbr = integ.run(state, thermo, number_of_steps=10, nbr_list=nbr)
    gc.collect()
    gc.disable()
    try:
        for steps in (20, 1, 300, 20):  # (with and without rebuilds inside; a carried and a fresh entry)
            state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
            alive = [o for o in gc.get_objects() if type(o) is I._RunState]
            assert not alive, (steps, len(alive))
    finally:
        gc.enable()


# ---------------------------------------------------------------------------------------------------------------------
# The queue's debug mode (include/chx.h: chx_queue_debug; chiron_amd/ops.py: fresh): a missing join must fail EVERY time
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("workload,steps", [("c3", 1500)], ids=["c3-32768"])
def test_debug_delays_do_not_change_a_bit(workload, steps):
    """CHX_QUEUE_DEBUG=250: a 250 us idle packet in front of every launch whose output leaves the queue, poison in every
    such output before it is written -- same hash as the plain run, and the idle packets were really dispatched."""
    plain = _run({}, workload, steps)
    slow = _run({"CHX_QUEUE_DEBUG": "250"}, workload, steps)
    assert slow["hash"] == plain["hash"] and slow["rebuilds"] == plain["rebuilds"], (slow, plain)
    # (five runs with a gate and a final launch each; the rebuild chains add six more each when they ride on the queue --
    # not with synchronous rebuilds, CHX_SYNC_REBUILD=1, nor behind a join, which an unthrottled window ends most segments with)
    assert plain["idle_packets"] == 0 and slow["idle_packets"] >= 10, (slow, plain)


@pytest.mark.parametrize("path", ["verlet", "verlet_build_on_entry", "all_pairs"])
def test_a_missing_join_reads_poison_every_time(path):
    """The bug class of round 5 (a run() that returned without joining the step queue and passed on timing), made
    deterministic: with the debug mode on, a run() whose final join is deliberately skipped (test hook) hands back a state
    that still holds the poison -- ten times out of ten -- and the same run() with its join hands back finite numbers."""
    sys.path.insert(0, ROOT)
    import bench
    from chiron_amd import neighbors, ops, unit as u

    if path.startswith("verlet"):
        fluid, pot, state, thermo, lst, integ = bench.make_system("c3", 300.0, 32, "cell")
    else:
        fluid, pot, state, thermo, _, integ = bench.make_system("c2", 300.0, 32, "cell")
        lst = neighbors.PairListNsqrd(neighbors.OrthogonalPeriodicSpace(), cutoff=3.0 * bench.SIGMA_NM * u.nanometer)
    if not ops.step_queue_on():
        pytest.skip("the step queue is switched off (CHX_AQL=0)")
    # (Verlet list, two ways: runs that continue each other on a carried-over list, and runs that build their list on entry
    # -- those confirm the build's status block before they return, which until round 6 waited for the whole queue when
    # the block had not arrived and so, under the debug delays, hid the very join under test: the confirmation now waits
    # for the block alone, NeighborListNsqrd._wait_for_status)
    integ.list_carry = path == "verlet"
    state, lst = integ.run(state, thermo, number_of_steps=10, nbr_list=lst)  # velocities, buffers, code paths warm
    state, lst = integ.run(state, thermo, number_of_steps=10, nbr_list=lst)
    ops.is_poison(state.positions)  # (torch's own kernels of the check loaded before anything is timed against them)
    # (10 ms of delay: the consumer's read below must come before the delayed launch writes, whatever the host is doing)
    old = ops.queue_debug(10000)
    try:
        spins0 = ops.queue_debug_spins()
        caught = 0
        for _ in range(10):
            ops._debug_skip_final_join = 1
            try:
                out, lst2 = integ.run(state, thermo, number_of_steps=12, nbr_list=lst)
                # the consumer's view, on the torch stream, without a join: the run's final launch has not written yet
                poisoned = ops.is_poison(out.positions) or ops.is_poison(out.velocities)
            finally:
                ops._debug_skip_final_join = 0
                ops.steps_join()  # (the join the run() skipped)
                torch.cuda.synchronize()
            caught += int(poisoned)
            # ... and behind the join the same arrays hold the run's result
            assert torch.isfinite(out.positions).all() and torch.isfinite(out.velocities).all()
            if path == "verlet":
                state, lst = out, lst2
        assert caught == 10, f"the missing join was caught {caught} times out of 10"
        assert ops.queue_debug_spins() > spins0
        # the intact run(): never poison
        for _ in range(3):
            out, lst2 = integ.run(state, thermo, number_of_steps=12, nbr_list=lst)
            assert not ops.is_poison(out.positions) and not ops.is_poison(out.velocities)
            if path == "verlet":
                state, lst = out, lst2
        if path == "verlet":
            assert integ.n_carried_runs >= 10  # (the runs under test did start on carried lists)
    finally:
        ops.queue_debug(old)


def test_the_fence_free_assumption_is_checked_again_while_a_process_lives():
    """The steady-state packets carry no cache fences on the strength of a check the process's first queue runs once
    (csrc/chx_aql.hip: selfcheck_fences).  Round-5 review: nothing re-checked later -- under another process on the card,
    after a clock or firmware state change.  Every queue now repeats the check every 2^20 packets, at a join; with the
    period shortened to 25 packets the trajectory's five run() calls run it again at three of their joins (the 3000-, the
    20- and the 100-step call), still report no stale word, keep their fence-free packets and produce the plain run's bits."""
    plain = _run({}, "c3", 3000)
    often = _run({"CHX_AQL_RECHECK_EVERY": "25"}, "c3", 3000)
    assert plain["fence_checks"] == 1, plain
    assert often["fence_checks"] >= 3, often
    assert often["status"] == "" and often["queue"]["hot_release"] == "none", often
    assert often["hash"] == plain["hash"]


def test_a_missing_join_in_the_batched_loop_reads_poison_every_time():
    """The same detector on ``integrators.run_batch`` (several replicas per launch): its one join -- behind the batch of
    trailing kicks -- skipped on purpose: every replica's returned state is still poison, ten times out of ten."""
    sys.path.insert(0, ROOT)
    import bench
    from chiron_amd import ops
    from chiron_amd.integrators import run_batch
    from chiron_amd.rng import PRNGKey

    reps = []
    for k in range(3):
        fluid, pot, state, thermo, nbr, integ = bench.make_system("n8k", 300.0 * 1.05 ** k, 32, "cell")
        integ.list_carry = False
        state.current_PRNG_key = PRNGKey(5, k)
        reps.append([integ, state, thermo, nbr])
    if not ops.step_queue_on():
        pytest.skip("the step queue is switched off (CHX_AQL=0)")

    def go(n):
        return run_batch([r[0] for r in reps], [r[1] for r in reps], [r[2] for r in reps], n, [r[3] for r in reps])

    for r, (s, l) in zip(reps, go(10)):
        r[1], r[3] = s, l
    ops.is_poison(reps[0][1].positions)  # (torch's kernels of the check loaded before anything is timed against them)
    old = ops.queue_debug(10000)
    try:
        caught = 0
        for _ in range(10):
            ops._debug_skip_final_join = 1
            try:
                out = go(12)
                poisoned = all(ops.is_poison(s.positions) or ops.is_poison(s.velocities) for s, _ in out)
            finally:
                ops._debug_skip_final_join = 0
                ops.steps_join()
                torch.cuda.synchronize()
            caught += int(poisoned)
            assert all(torch.isfinite(s.positions).all() and torch.isfinite(s.velocities).all() for s, _ in out)
        assert caught == 10, f"the missing join was caught {caught} times out of 10"
        out = go(12)
        assert not any(ops.is_poison(s.positions) or ops.is_poison(s.velocities) for s, _ in out)
    finally:
        ops.queue_debug(old)
