"""Micro-benchmark of the pair-force / fused-step kernels (diagnostic, GPU box only).
usage: python scripts/kernel_bench.py [workload ...]   env CHX_FL_G overrides lanes/particle"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from chiron_amd import _lib, ops
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd import unit


def ev_time(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    ops.steps_join()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    ops.steps_join()  # (step launches run on the step queue, which an event on the stream does not see: include/chx.h)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps  # us


for wl in sys.argv[1:] or ["c3", "c4"]:
    for g in os.environ.get("GS", "4,8,16").split(","):
        if g == "auto":
            os.environ.pop("CHX_FL_G", None)
        else:
            os.environ["CHX_FL_G"] = g
        fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
        warm = int(os.environ.get("KB_WARM", "0"))  # measure on an equilibrated state instead of the lattice start
        snap_dir = os.environ.get("KB_SNAPSHOT_DIR")  # ... kept on disk, so that profiled passes need not re-run (and trace) it
        snap = os.path.join(snap_dir, f"{wl}_{warm}.npz") if (snap_dir and warm) else None
        if snap and os.path.exists(snap):
            import numpy as np

            d = np.load(snap)
            state.positions = torch.as_tensor(d["x"]).cuda()
            state.velocities = torch.as_tensor(d["v"]).cuda()
        elif warm:
            state, nbr = integ.run(state, thermo, number_of_steps=warm, nbr_list=nbr)
            if snap:
                import numpy as np

                os.makedirs(snap_dir, exist_ok=True)
                np.savez(snap, x=state.positions.cpu().numpy(), v=state.velocities.cpu().numpy())
        if snap or not (warm and os.environ.get("KB_KEEP_LIST")):  # KB_KEEP_LIST: time on the list the run ended with
            nbr.build_from_state(state)
        fl = nbr.force_list
        n = fl.n
        p_nl = int(nbr.n_neighbors.sum())
        alg = 24 * n + 16 * p_nl
        x4 = fl.pack(state.positions, 4)
        lib = _lib.load()
        f3 = torch.empty((n, 3), dtype=torch.float32, device="cuda")
        box = nbr._box_h
        force_launch = lambda: lib.chx_lj_fl_energy_force(ops._ptr(x4), ops._ptr(fl.fl), ops._ptr(fl.deg), n, fl.g, fl.chunks, box,
                                                          C.c_float(pot.sigma), C.c_float(pot.epsilon), C.c_float(pot.cutoff),
                                                          ops._ptr(f3), None, ops._ptr(fl.ws), ops._stream())
        # fused step kernel (skin check can never fire: huge skin)
        half_dt, a, b = langevin_constants(1.0 * unit.femtoseconds, 1.0 / unit.picoseconds)
        from chiron_amd.utils import get_list_of_mass
        from chiron_amd._device import as_tensor
        mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
        v0 = state.velocities if state.velocities is not None else torch.zeros((n, 3), device="cuda")
        st = _RunState(integ, pot, nbr, state.positions, v0, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "fused")
        st.params.skin = 1e6
        step = [0]

        def one():
            even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
            ops.langevin_fused_chunk(st.params, even, odd, step[0], CH, 2, 1)
            step[0] += CH
        CH = 64  # launches per C call: a single launch per ctypes call is host-bound at ~5 us
        if os.environ.get("KB_FULL_LIST"):
            st.params.prune_every = 0  # walk the full list (default: production pruning, a pass every 32 steps included)
        # clock warm-up: the card needs 0-2 s of load to reach 2.4 GHz.  CAVEAT: the skin check is disabled here, so the
        # list is never rebuilt while this spins; after ~10^5 steps the pruned list holds only the few pairs of the stale
        # full list that are still close and the step launch reads 4.0 us instead of the 5.1 us of a live run
        # (scripts/chunk_age_probe.py measures the real loop).  Use it for A/B decompositions, not for absolute numbers.
        spin = float(os.environ.get("KB_SPIN_S", "0"))
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < spin:
            # (the state-neutral pair-force kernel, as bench.py's time_kernels does: the step kernel would age the list,
            # and under a profiler its intercepted queue does not keep the card at its loaded clock)
            for _ in range(200):
                force_launch()
            torch.cuda.synchronize()
        # the pair-force kernel is timed AFTER the clock warm-up above (which runs the STEP kernel on the state's own
        # buffers: x4 / fl, which the force launches read, are untouched): at the card's idle clock level the same
        # launch takes 14 % longer (round 3), and a rocprofv3 --stats pass over this script then averages exactly the
        # launches of this one state at the loaded clock
        t_f = ev_time(force_launch)
        t_s = ev_time(one, reps=int(os.environ.get("KB_REPS", "20")), warm=3) / CH
        print(f"{wl} g={fl.g} maxdeg={fl.maxdeg} P_nl={p_nl} alg={alg/1e6:.2f}MB  force {t_f:7.2f} us ({alg/t_f/1e3:7.1f} GB/s, {alg/t_f/8e6*100:5.1f}%)   fused step {t_s:7.2f} us")
