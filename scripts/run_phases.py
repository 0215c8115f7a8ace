"""Host time line inside a short LangevinIntegrator.run(K) (GPU box only): medians over many calls of the time from entry
to each mark.  usage: python scripts/run_phases.py [K] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(1)
from chiron_amd._device import pin_process_cpus
pin_process_cpus()
import bench
from chiron_amd import integrators as I, neighbors as N

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
marks = {}

def wrap(obj, name, tag, when="after"):
    orig = getattr(obj, name)
    def f(*a, **k):
        if when == "before":
            marks.setdefault(tag, time.perf_counter())
        r = orig(*a, **k)
        if when == "after":
            marks[tag] = time.perf_counter()
        return r
    setattr(obj, name, f)

from chiron_amd import ops as OPS, utils as U
wrap(U, "get_list_of_mass", "0a masses listed")
wrap(N.NeighborListNsqrd, "_build_speculative", "0c _build_speculative entered", "before")
wrap(OPS, "cell_grid", "0d cell grid known")
wrap(OPS, "nl_build_cell_fl", "0e build call entered", "before")
wrap(N.NeighborListNsqrd, "_build_speculative", "1 build enqueued")
wrap(I.LangevinIntegrator, "_take_carry", "0 carry decided")
wrap(OPS, "steps_open", "0b queue being opened", "before")
wrap(OPS, "steps_open", "0b' queue opened")
wrap(OPS, "steps_join", "3d join returned")
_lib_obj = OPS._lib.load()
for _name, _tag in (("chx_cell_bin_fl", "0f binning call"), ("chx_nl_rows_cell_fl", "0g rows call")):
    _orig = getattr(_lib_obj, _name)
    def _mk(orig, tag):
        def f(*a):
            marks.setdefault(tag + " entered", time.perf_counter())
            r = orig(*a)
            marks.setdefault(tag + " returned", time.perf_counter())
            return r
        return f
    setattr(_lib_obj, _name, _mk(_orig, _tag))
wrap(I._RunState, "_setup_carry", "1a carry set up", "before")
wrap(I._RunState, "_load", "1b _load entered", "before")
wrap(I._RunState, "_open_generation", "1c generation opened")
wrap(I._RunState, "__init__", "2 state packed (pack enqueued)")
wrap(I._RunState, "_enqueue_window", "3 first steps enqueued", "before")
wrap(I._RunState, "_enqueue_window", "3a first window call returned")
wrap(I._RunState, "_enqueue_kick", "3b kick being enqueued", "before")
wrap(torch.cuda.Event, "synchronize", "3c waiting for the device", "before")
wrap(I._RunState, "final_kick", "4 kick done (device drained)")
wrap(I._RunState, "state_user", "5 state unpacked (enqueued)")
wrap(I._RunState, "finish", "5a finish() done")
wrap(I._RunState, "offer_carry", "5c carry offered")
from chiron_amd import states as S
wrap(S.SamplerState, "_copy_with", "5b state copied")
if len(sys.argv) > 3:
    integ.list_carry = sys.argv[3] != "0"
state, nbr = integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
rows = {}
for _ in range(400):
    torch.cuda.synchronize()
    marks.clear()
    ta = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=K, nbr_list=nbr)
    tb = time.perf_counter()
    torch.cuda.synchronize()
    tc = time.perf_counter()
    for k, v in list(marks.items()) + [("6 run() returned", tb), ("7 device idle", tc)]:
        rows.setdefault(k, []).append(v - ta)
for k in sorted(rows):  # (a mark that only some calls pass -- a list renewal -- shows with its count)
    print(f"  {k:36s} {1e6 * np.median(rows[k]):8.1f} us   (n = {len(rows[k])})")
