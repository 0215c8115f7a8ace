"""In-kernel time stamps of the fused step kernel (diagnostic build -DCHX_EXP_STAMPS, GPU box only):
where does a block's time go, how skewed are block starts, how long is the gap between launches?
usage: CHX_ALLOW_STALE_LIB=1 python scripts/exp_stamps.py [c3|c4]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from chiron_amd import _lib, ops, unit
from chiron_amd._device import as_tensor
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd.utils import get_list_of_mass

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)  # equilibrated-ish state
lib = _lib.load()
fl = nbr.force_list
n = fl.n
half_dt, a, b = langevin_constants(1.0 * unit.femtoseconds, 1.0 / unit.picoseconds)
mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
st = _RunState(integ, pot, nbr, state.positions, state.velocities, mass, 2.494, half_dt, a, b, state.current_PRNG_key, None, "fused")
st.params.skin = 1e6
step = 0
for _ in range(6):  # 6 chunks of 32: the last four launches are steps 188..191 (none of them a refresh: 160 is)
    even, odd = (st.x_a, st.x_b) if (step & 1) == 0 else (st.x_b, st.x_a)
    ops.langevin_fused_chunk(st.params, even, odd, step, 32, 2, 1)
    step += 32
torch.cuda.synchronize()
raw = np.zeros((4, 4096, 8), dtype=np.uint64)
rc = C.CDLL(_lib.LIB_PATH).chx_debug_read_stamps(raw.ctypes.data_as(C.c_void_p))
assert rc == 0
nb = (n * fl.g + 255) // 256
s = raw[:, :nb, :].astype(np.int64)
clk = 2.4e3  # shader cycles per us (approximate; the ratios are what matters)
names = ["entry -> row loads back", "row loads back -> Philox done", "Philox done -> gathers back", "gathers back -> force loop done",
         "force loop done -> after reduce+barrier", "barrier -> end (integrate, stores issued)"]
idx = [(0, 1), (1, 7), (7, 2), (2, 3), (3, 4), (4, 5)]
print(f"{wl}: N={n} G={fl.g} blocks={nb}; wave 0 of every block, medians over blocks and the last 4 launches")
tot = 0.0
for nm_, (i0, i1) in zip(names, idx):
    d = (s[:, :, i1] - s[:, :, i0]) / clk
    print(f"  {nm_:44s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}")
    tot += np.median(d)
d = (s[:, :, 5] - s[:, :, 0]) / clk
print(f"  {'entry -> end':44s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}")
rt0 = (s[:, :, 6] & 0xFFFFFFFF).astype(np.float64) / 100.0  # us (100 MHz), low 32 bits
rt1 = ((s[:, :, 6] >> 32) & 0xFFFFFFFF).astype(np.float64) / 100.0
order = np.argsort(rt0.min(axis=1))
for k in order:
    print(f"  launch slot {k}: first block entry at {rt0[k].min() - rt0[order[0]].min():8.2f} us, last block entry +{rt0[k].max() - rt0[k].min():5.2f}, "
          f"first block end +{rt1[k].min() - rt0[k].min():5.2f}, last block end +{rt1[k].max() - rt0[k].min():5.2f}")
for a_, b_ in zip(order[:-1], order[1:]):
    print(f"  gap: last end of slot {a_} -> first entry of slot {b_}: {rt0[b_].min() - rt1[a_].max():5.2f} us;  start-to-start {rt0[b_].min() - rt0[a_].min():5.2f} us")

# ---- where are the slow blocks?  (logical block index = spatial slab order; 1/8 of the range per XCD)
body = ((s[:, :, 5] - s[:, :, 0]) / clk).mean(axis=0)
loop = ((s[:, :, 3] - s[:, :, 2]) / clk).mean(axis=0)
rowl = ((s[:, :, 1] - s[:, :, 0]) / clk).mean(axis=0)
degi = st.params.deg_inner
import ctypes
deg_t = None
for name in ("deg_inner",):
    pass
grp = max(1, nb // 32)
print("  logical blocks in 32 groups: mean body / row-load / force-loop time (us)")
for gi in range(0, nb, grp):
    sl = slice(gi, min(nb, gi + grp))
    print(f"    blocks {gi:5d}-{min(nb, gi + grp) - 1:5d}: body {body[sl].mean():6.2f} (max {body[sl].max():6.2f})  row loads {rowl[sl].mean():5.2f}  loop {loop[sl].mean():5.2f}")
worst = np.argsort(body)[-12:]
print("  slowest blocks:", [(int(b_), round(float(body[b_]), 2)) for b_ in worst])

# ---- shader clock seen by the kernel: s_memtime cycles per s_memrealtime tick (100 MHz)
cyc = (s[:, :, 5] - s[:, :, 0]).astype(np.float64)
drt = (rt1 - rt0) * 100.0  # ticks
ok = drt > 50
print(f"  shader clock (median over blocks): {np.median(cyc[ok] / drt[ok]) * 0.1:.3f} GHz;  start-to-start {np.median(np.diff(np.sort(rt0.min(axis=1)))):.2f} us")
