"""EXPERIMENT: pair-force kernel with a block's rows handed out in order of descending degree (chx_exp_lj_fl_ordered)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from chiron_amd import _lib, ops

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
warm = int(os.environ.get("KB_WARM", "20000"))
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
state, nbr = integ.run(state, thermo, number_of_steps=warm, nbr_list=nbr)
nbr.build_from_state(state)
fl = nbr.force_list
n, g = fl.n, fl.g
x4 = fl.pack(state.positions, 4)
lib = C.CDLL(_lib.LIB_PATH)
_lib.load()
f_ref = torch.empty((n, 3), dtype=torch.float32, device="cuda")
f_ord = torch.empty((n, 3), dtype=torch.float32, device="cuda")
box = nbr._box_h
args = (ops._ptr(x4), ops._ptr(fl.fl), ops._ptr(fl.deg), n, g, fl.chunks, box, C.c_float(pot.sigma), C.c_float(pot.epsilon), C.c_float(pot.cutoff))
base = lambda: _lib.load().chx_lj_fl_energy_force(*args, ops._ptr(f_ref), None, ops._ptr(fl.ws), ops._stream())
lib.chx_exp_lj_fl_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _lib.Box, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]


def timed(fn, reps=300):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(out)[2]


for _ in range(3000):
    base()
torch.cuda.synchronize()
print(f"{wl} g={g} n={n} maxdeg={fl.maxdeg} mean deg {float(fl.deg.float().mean()):.1f}")
print(f"  cell order           : {timed(base):7.2f} us")
for group in (256 // g, 2 * 256 // g, 4 * 256 // g, 16 * 256 // g):
    if n % group:
        continue
    d = fl.deg.view(-1, group)
    order = (d.argsort(dim=1, descending=True, stable=True) + torch.arange(0, n, group, device="cuda").view(-1, 1)).int().contiguous().view(-1)
    fn = lambda: lib.chx_exp_lj_fl_ordered(ops._ptr(x4), ops._ptr(fl.fl), ops._ptr(fl.deg), ops._ptr(order), n, g, fl.chunks, box,
                                           C.c_float(pot.sigma), C.c_float(pot.epsilon), C.c_float(pot.cutoff), ops._ptr(f_ord), ops._stream())
    t = timed(fn)
    base(); fn(); torch.cuda.synchronize()
    print(f"  sorted within {group:5d} rows: {t:7.2f} us   forces identical: {torch.equal(f_ref, f_ord)}")
ident = torch.arange(n, device="cuda").int()
fn = lambda: lib.chx_exp_lj_fl_ordered(ops._ptr(x4), ops._ptr(fl.fl), ops._ptr(fl.deg), ops._ptr(ident), n, g, fl.chunks, box,
                                       C.c_float(pot.sigma), C.c_float(pot.epsilon), C.c_float(pot.cutoff), ops._ptr(f_ord), ops._stream())
print(f"  identity order array : {timed(fn):7.2f} us")
