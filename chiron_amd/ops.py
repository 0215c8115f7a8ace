"""Tensor-level entry points: thin wrappers that hand device pointers to libchx.

Every function here launches hand-written HIP kernels through the C ABI on the
caller's current HIP stream.  Inputs must be ROCm device tensors; there is no CPU
implementation behind these calls (a CPU tensor or a missing ``libchx.so`` raises).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import Box, LangevinParams, check
from .rng import PRNGKey


# ---------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------
def _require_device(*tensors) -> None:
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise _lib.ChxError(
                "chiron_amd kernels run on the ROCm device only: got a "
                f"{type(t).__name__} on {getattr(t, 'device', 'host')}. There is no CPU fallback."
            )
        if not t.is_contiguous():
            raise _lib.ChxError("chiron_amd kernels need contiguous tensors")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current HIP stream as a raw handle.  ``torch.cuda.current_stream()`` builds a Python
    Stream object (~9 us, a dozen times per integrator call); the raw accessor is a plain C call."""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_CUR_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    return x.contiguous()


def make_box(box_vectors, periodic: bool = True) -> Box:
    """Host-side ``chx_box_t`` from a (3,3) box (tensor / ndarray) -- diag only (neighbors.py:74-76)."""
    if box_vectors is None or not periodic:
        return Box(0, 0.0, 0.0, 0.0)
    if isinstance(box_vectors, torch.Tensor):
        b = box_vectors.detach().to("cpu", torch.float32).numpy()
    else:
        b = np.asarray(box_vectors, dtype=np.float32)
    if b.shape != (3, 3):
        raise ValueError(f"box_vectors should be a 3x3 array, shape provided: {b.shape}")
    return Box(1, float(b[0, 0]), float(b[1, 1]), float(b[2, 2]))


def cutoff_threshold(c: float) -> float:
    return float(_lib.load().chx_cutoff_threshold_h(C.c_float(c)))


# ---------------------------------------------------------------------------
# spaces
# ---------------------------------------------------------------------------
def displacement(x1: torch.Tensor, x2: torch.Tensor, box: Box) -> Tuple[torch.Tensor, torch.Tensor]:
    x1, x2 = torch.broadcast_tensors(_f32(x1), _f32(x2))
    x1, x2 = x1.contiguous(), x2.contiguous()
    _require_device(x1, x2)
    if x1.shape[-1] != 3:
        raise ValueError(f"last axis must have length 3, got {tuple(x1.shape)}")
    r = torch.empty_like(x1)
    d = torch.empty(x1.shape[:-1], dtype=torch.float32, device=x1.device)
    n = d.numel()
    check(_lib.load().chx_displacement(_ptr(x1), _ptr(x2), n, box, _ptr(r), _ptr(d), _stream()), "chx_displacement")
    return r, d


def wrap(x: torch.Tensor, box: Box) -> torch.Tensor:
    x = _f32(x)
    _require_device(x)
    out = torch.empty_like(x)
    check(_lib.load().chx_wrap(_ptr(x), _ptr(out), x.numel() // 3, box, _stream()), "chx_wrap")
    return out


def scale(x: torch.Tensor, s: float) -> torch.Tensor:
    x = _f32(x)
    _require_device(x)
    out = torch.empty_like(x)
    check(_lib.load().chx_scale(_ptr(x), _ptr(out), x.numel(), C.c_float(s), _stream()), "chx_scale")
    return out


# ---------------------------------------------------------------------------
# neighbor list
# ---------------------------------------------------------------------------
def cell_grid(box: Box, cutoff_plus_skin: float, n: int):
    nc = (C.c_int32 * 3)()
    total = _lib.load().chx_cell_grid_h(box, C.c_float(cutoff_plus_skin), n, nc)
    return int(total), nc


def nl_build_nsq(xyz: torch.Tensor, box: Box, cutoff_plus_skin: float, m: int):
    xyz = _f32(xyz)
    _require_device(xyz)
    n = xyz.shape[0]
    dev = xyz.device
    nl = torch.empty((n, m), dtype=torch.int32, device=dev)  # uint32 payload
    mask = torch.empty((n, m), dtype=torch.int32, device=dev)
    n_nbr = torch.empty((n,), dtype=torch.int32, device=dev)
    check(
        _lib.load().chx_nl_build_nsq(_ptr(xyz), n, m, box, C.c_float(cutoff_plus_skin), _ptr(nl), _ptr(mask), _ptr(n_nbr), _stream()),
        "chx_nl_build_nsq",
    )
    return nl, mask, n_nbr


def nl_build_cell(xyz: torch.Tensor, box: Box, cutoff_plus_skin: float, m: int, ncell: int, nc):
    """Returns (nl, mask, n_nbr, sorted_ids, status)."""
    xyz = _f32(xyz)
    _require_device(xyz)
    n = xyz.shape[0]
    dev = xyz.device
    nl = torch.empty((n, m), dtype=torch.int32, device=dev)
    mask = torch.empty((n, m), dtype=torch.int32, device=dev)
    n_nbr = torch.empty((n,), dtype=torch.int32, device=dev)
    cell_of = torch.empty((n,), dtype=torch.int32, device=dev)
    cell_count = torch.zeros((ncell + 1,), dtype=torch.int32, device=dev)  # (zero on entry: include/chx.h)
    cell_start = torch.empty((ncell + 1,), dtype=torch.int32, device=dev)
    sorted_ids = torch.empty((n,), dtype=torch.int32, device=dev)
    status = torch.empty((4,), dtype=torch.int32, device=dev)
    check(
        _lib.load().chx_nl_build_cell(
            _ptr(xyz), n, m, box, C.c_float(cutoff_plus_skin), nc, _ptr(cell_of), _ptr(cell_count),
            _ptr(cell_start), _ptr(sorted_ids), _ptr(status), _ptr(nl), _ptr(mask), _ptr(n_nbr), _stream(),
        ),
        "chx_nl_build_cell",
    )
    return nl, mask, n_nbr, sorted_ids, status


_BUILD_LISTS = os.environ.get("CHX_BUILD_LISTS", "1") != "0"


def nl_build_cell_fl(xyz: torch.Tensor, box: Box, cutoff_plus_skin: float, m: int, ncell: int, nc, g: int, chunks: int,
                     scratch: Optional[dict] = None, riders: Optional[dict] = None, half_rows: bool = True,
                     checked: bool = False):
    """One-pass cell build (half-list rows without padding + force list).

    Returns (nl_rows, first, n_nbr, sorted_ids, fl, deg, status); nothing is synchronised here.
    ``half_rows=False``: the half rows are counted (n_nbr, which the capacity checks need) but not written or sorted
    -- nl_rows and first come back as None (the integrator's rebuilds: the step kernels read the force list only).
    ``scratch`` (a dict the caller keeps between builds) lets the kernel-internal work arrays be
    reused: the arrays that become part of the list are always fresh.
    ``riders`` (include/chx.h: chx_build_riders_t): ``copy`` -- a (n,3) tensor that receives a copy of ``xyz``;
    ``pack`` -- (v3, mass, ref3 or None, x4_out, v_out, mass_out, ref4_out or None): the Langevin state packed into
    the new internal order; ``ctrl`` / ``ctrl_keep`` -- control block to reset; ``publish`` -- pinned host tensor that
    receives status[0..7] from the chain's last launch.
    ``checked``: the caller has verified that ``xyz`` and the tensors of the ``copy`` and ``pack`` riders are contiguous
    fp32 device tensors (NeighborListNsqrd._build_speculative, LangevinIntegrator.run: on the path to a run()'s first
    launch every check made twice is device idle time).
    """
    if not checked:
        xyz = _f32(xyz)
        _require_device(xyz)
    n = xyz.shape[0]
    dev = xyz.device
    i32 = dict(dtype=torch.int32, device=dev)
    key = (n, ncell, dev)
    # (the internal order: a fresh array per build -- it becomes part of the list --, made ahead of time when the caller
    # saw this build coming: neighbors.NeighborListNsqrd._prepare_next_build)
    sorted_ids = scratch.pop("next_ids", None) if scratch is not None else None
    if sorted_ids is None or sorted_ids.shape[0] != n or sorted_ids.device != dev:
        sorted_ids = torch.empty((n,), **i32)
    if scratch is not None and scratch.get("key") == key:
        cell_of, cell_count, cell_start, status, xs4 = scratch["arrays"]
        cand = scratch.get("cand")
        slots = scratch.get("slots")
    else:
        steps_join()  # (the zero fill below is a torch kernel on the stream: not beside launches riding on the step queue)
        cell_of = torch.empty((n,), **i32)
        cell_count = torch.zeros((ncell + 1,), **i32)  # (zero on entry, left zeroed by every build: include/chx.h)
        cell_start = torch.empty((ncell + 1,), **i32)
        status = torch.empty((_lib.BUILD_STATUS_WORDS,), **i32)
        xs4 = torch.empty((n, 4), dtype=torch.float32, device=dev)
        cand = None  # (per-cell candidate index lists: made below, the first time a build wants them)
        # per-cell id slots (include/chx.h: chx_build_riders_t::cell_slots): the chain needs no scatter launch.  Twice the
        # mean occupancy + 32; a cell beyond that voids the build (status[5]) and the caller turns the slots off
        slots = None
        slot_cap = int(os.environ.get("CHX_BUILD_SLOT_CAP", "0")) or (int(2.0 * n / ncell) + 32 + 63) // 64 * 64  # (env: tests)
        if slot_cap <= 1024 and os.environ.get("CHX_BUILD_SLOTS", "1") != "0" and not (scratch or {}).get("no_slots"):
            slots = (torch.empty((ncell, slot_cap), **i32), slot_cap)
        if scratch is not None:
            scratch["key"], scratch["arrays"], scratch["cand"] = key, (cell_of, cell_count, cell_start, status, xs4), cand
            scratch["slots"] = slots
    if slots is not None and scratch is not None and scratch.get("no_slots"):
        slots = scratch["slots"] = None
    # Per-cell candidate index lists (include/chx.h: chx_build_riders_t::cand_list) for the row kernels that read them: the
    # per-particle kernel (grids below 8192 cells: beyond that the cell-wise kernel shares the neighbourhood scan among a
    # cell's rows itself) and, at any size, the filtered kernel of the integrator's rebuilds (half_rows=False:
    # k_nl_build_sub_fl).  Capacity = 1.5 x the mean neighbourhood; a longer one keeps the range search.  A build that
    # will not read them does not have them written.
    if (ncell < 8192 or not half_rows) and _BUILD_LISTS:
        if cand is None:
            cap = min(8192, (int(1.5 * 27.0 * n / ncell) + 64 + 63) // 64 * 64)
            if ncell * cap * 4 < (1 << 30):  # (the kernels address a list with 32-bit byte offsets)
                cand = (torch.empty((ncell, cap), **i32), torch.empty((ncell,), **i32), cap)
                if scratch is not None:
                    scratch["cand"] = cand
    else:
        cand = None
    rd = None
    publish = None
    if riders or cand is not None or slots is not None:
        riders = riders or {}
        cp = riders.get("copy")
        pk = riders.get("pack")
        ct = riders.get("ctrl")
        pb = riders.get("publish")
        # The rider block is kept with the work arrays: its members that point at things which outlive a build -- the
        # candidate lists and id slots (this scratch), the integrator's own buffers (pack outputs, control block), the
        # list's pinned status block -- are filled once and reused while they are the same objects (references held
        # here, so an id cannot be recycled); what changes from build to build (the copy of the positions, the velocities
        # to pack, ctrl_keep) is set every time.  ~4 us less Python in front of a run()'s first launch.
        stable = (cand[0] if cand is not None else None, slots[0] if slots is not None else None,
                  pk[1] if pk is not None else None, pk[3] if pk is not None else None, pk[4] if pk is not None else None,
                  pk[5] if pk is not None else None, pk[6] if pk is not None else None, ct, pb)
        kept = scratch.get("rd") if scratch is not None else None
        same = kept is not None
        if same:
            for u, w in zip(kept[0], stable):
                if u is not w:
                    same = False
                    break
        if same:
            rd, publish = kept[1], kept[2]
        else:
            rd = _lib.BuildRiders()
            if cand is not None:
                rd.cand_list, rd.cand_len, rd.cand_cap = cand[0].data_ptr(), cand[1].data_ptr(), cand[2]
            if slots is not None:
                rd.cell_slots, rd.cell_slot_cap = slots[0].data_ptr(), slots[1]
            if pk is not None:
                _, mass, _, x4_out, v_out, mass_out, ref4_out = pk
                _require_device(mass, x4_out, v_out, mass_out)
                if mass.dtype != torch.float32 or not mass.is_contiguous():
                    raise _lib.ChxError("pack rider: velocities and masses must be contiguous fp32 tensors")
                rd.mass = mass.data_ptr()
                rd.x4_out, rd.v_out, rd.mass_out = x4_out.data_ptr(), v_out.data_ptr(), mass_out.data_ptr()
                rd.ref4_out = ref4_out.data_ptr() if ref4_out is not None else None
            if ct is not None:
                rd.ctrl_reset = ct.data_ptr()
            if pb is not None:
                if not pb.is_pinned():
                    raise _lib.ChxError("publish needs a pinned (host-mapped) destination tensor")
                publish = C.c_void_p(pb.data_ptr())
            if scratch is not None:
                scratch["rd"] = (stable, rd, publish)
        if cp is not None:
            if not checked:
                _require_device(cp)
            rd.xyz_copy_out = cp.data_ptr()
        else:
            rd.xyz_copy_out = None
        if pk is not None:
            v3, ref3 = pk[0], pk[2]
            if not checked:
                _require_device(v3)
                if v3.dtype != torch.float32 or not v3.is_contiguous():
                    raise _lib.ChxError("pack rider: velocities and masses must be contiguous fp32 tensors")
            rd.v3 = v3.data_ptr()
            rd.ref3 = ref3.data_ptr() if ref3 is not None else None
        rd.ctrl_keep = int(riders.get("ctrl_keep", 0)) if ct is not None else 0
    lib = _lib.load()
    stream = _stream()
    cs = float(cutoff_plus_skin)  # (argtypes are declared: a float and the integer of a data_ptr() convert in C, without an object each)
    # the binning launches go to the device FIRST; the arrays the rows go into are allocated while they run (at the entry
    # of a short run() the device idles until the first launch is made: include/chx.h, chx_cell_bin_fl)
    # (the work arrays' pointers as ctypes objects, made once per set of arrays)
    wp = scratch.get("wptr") if scratch is not None else None
    if wp is None or wp[0] is not cell_of or wp[1] is not (cand[0] if cand is not None else None):
        wp = (cell_of, cand[0] if cand is not None else None, _ptr(cell_of), _ptr(cell_count), _ptr(cell_start), _ptr(status),
              _ptr(xs4), _ptr(cand[0]) if cand is not None else None, _ptr(cand[1]) if cand is not None else None)
        if scratch is not None:
            scratch["wptr"] = wp
    p_cell_of, p_cell_count, p_cell_start, p_status, p_xs4, p_cand0, p_cand1 = wp[2:]
    p_sorted = sorted_ids.data_ptr()
    check(
        lib.chx_cell_bin_fl(xyz.data_ptr(), n, box, cs, nc, p_cell_of, p_cell_count, p_cell_start, p_sorted,
                            p_status, p_xs4, C.byref(rd) if rd is not None else None, stream),
        "chx_cell_bin_fl",
    )
    # (`fresh`: what the row launches write is read from outside the queue -- by the force kernels of compute_energy, by
    # whoever looks at the list; poison in the queue's debug mode)
    nl = fresh((n, m), **i32) if half_rows else None
    first = fresh((n,), **i32) if half_rows else None
    n_nbr = fresh((n,), **i32)
    fl = fresh((chunks, n, FL_QUAD * g), **i32)
    deg = fresh((n,), **i32)
    check(
        lib.chx_nl_rows_cell_fl(n, m, box, cs, nc, p_cell_of, p_cell_start, p_sorted, p_status, p_xs4,
                                _ptr(nl) if half_rows else None, _ptr(first) if half_rows else None, _ptr(n_nbr), g, chunks, _ptr(fl), _ptr(deg), publish,
                                p_cand0, p_cand1, cand[2] if cand is not None else 0, stream),
        "chx_nl_rows_cell_fl",
    )
    return nl, first, n_nbr, sorted_ids, fl, deg, status


def nl_finish_padding(nl: torch.Tensor, first: torch.Tensor, n_nbr: torch.Tensor) -> torch.Tensor:
    """Pad the rows in place (reference fill rule) and return the (N,M) list mask."""
    _require_device(nl, first, n_nbr)
    n, m = nl.shape
    mask = torch.empty((n, m), dtype=torch.int32, device=nl.device)
    check(_lib.load().chx_nl_finish_padding(_ptr(nl), _ptr(first), _ptr(n_nbr), n, m, _ptr(mask), _stream()), "chx_nl_finish_padding")
    return mask


def pack_langevin(x3, v3, mass, ref3, perm, want_ref: bool, out=None, ctrl=None, publish=None, ctrl_keep: int = 0,
                  validate=None):
    """User-order state -> internal order in one launch: (x4, v, mass_internal, ref4 or None).
    ``out`` = a previous result of the same shapes to write into.  Riders on the same launch: ``ctrl`` = control
    block to reset (``ctrl_reset``; ``ctrl_keep`` bit 0 / 1: leave the reference-event counters / booked events alone),
    ``publish`` = (device block of 8 ints, pinned host block) to copy (``publish``); ``validate`` = (snap4, box, skin, step,
    host_ctrl tensor or None, sync2): the launch also answers chx_validate_snap's question and leaves the answer in the
    reset control block's halt word (include/chx.h: chx_pack_validate_langevin; needs ``ctrl``)."""
    x3, v3, mass = _f32(x3), _f32(v3), _f32(mass)
    _require_device(x3, v3, mass, perm)
    n = x3.shape[0]
    dev = x3.device
    if out is not None and out[0].shape[0] == n and (out[3] is not None) == bool(want_ref):
        x4, v, m_int, ref4 = out
    else:
        x4 = torch.empty((n, 4), dtype=torch.float32, device=dev)
        v = torch.empty((n, 3), dtype=torch.float32, device=dev)
        m_int = torch.empty((n,), dtype=torch.float32, device=dev)
        ref4 = torch.empty((n, 4), dtype=torch.float32, device=dev) if want_ref else None
    if want_ref:
        ref3 = _f32(ref3)
        _require_device(ref3)
    if validate is not None:
        snap4, box, skin, step, host_ctrl, sync2 = validate
        _require_device(snap4, sync2, ctrl)
        if publish:
            raise _lib.ChxError("pack_langevin: validate and publish do not combine")
        check(
            _lib.load().chx_pack_validate_langevin(
                box, C.c_float(skin), int(step), _ptr(snap4), C.c_void_p(host_ctrl.data_ptr()) if host_ctrl is not None else None,
                _ptr(sync2), _ptr(x3), _ptr(v3), _ptr(mass), _ptr(ref3) if want_ref else None, _ptr(perm), n, _ptr(x4), _ptr(v),
                _ptr(m_int), _ptr(ref4), _ptr(ctrl), int(ctrl_keep), _stream()),
            "chx_pack_validate_langevin",
        )
        return x4, v, m_int, ref4
    check(
        _lib.load().chx_pack_langevin(_ptr(x3), _ptr(v3), _ptr(mass), _ptr(ref3) if want_ref else None, _ptr(perm), n,
                                      _ptr(x4), _ptr(v), _ptr(m_int), _ptr(ref4), _ptr(ctrl), int(ctrl_keep),
                                      _ptr(publish[0]) if publish else None,
                                      C.c_void_p(publish[1].data_ptr()) if publish else None, _stream()),
        "chx_pack_langevin",
    )
    return x4, v, m_int, ref4


def validate_snap(x4, snap4, box: Box, skin: float, step: int, ctrl, host_ctrl=None) -> None:
    """Halt launch ``step`` unless every particle is within skin/2 of the positions its force list was built from
    (include/chx.h: chx_validate_snap)."""
    _require_device(x4, snap4, ctrl)
    check(
        _lib.load().chx_validate_snap(_ptr(x4), _ptr(snap4), x4.shape[0], box, C.c_float(skin), int(step), _ptr(ctrl),
                                      C.c_void_p(host_ctrl.data_ptr()) if host_ctrl is not None else None, _stream()),
        "chx_validate_snap",
    )


def nl_calculate(xyz, nl, nl_mask, box: Box, cutoff: float):
    xyz = _f32(xyz)
    _require_device(xyz, nl, nl_mask)
    n, m = nl.shape
    if xyz.shape[0] != n:
        raise ValueError(f"positions have {xyz.shape[0]} rows but the neighbor list was built for {n}")
    dev = xyz.device
    n_pairs = torch.empty((n,), dtype=torch.int32, device=dev)
    mask = torch.empty((n, m), dtype=torch.int32, device=dev)
    dist = torch.empty((n, m), dtype=torch.float32, device=dev)
    rij = torch.empty((n, m, 3), dtype=torch.float32, device=dev)
    check(
        _lib.load().chx_nl_calculate(_ptr(xyz), _ptr(nl), _ptr(nl_mask), n, m, box, C.c_float(cutoff), _ptr(n_pairs), _ptr(mask), _ptr(dist), _ptr(rij), _stream()),
        "chx_nl_calculate",
    )
    return n_pairs, mask, dist, rij


def nl_check(xyz, ref, box: Box, skin: float) -> torch.Tensor:
    xyz, ref = _f32(xyz), _f32(ref)
    _require_device(xyz, ref)
    flag = torch.empty((1,), dtype=torch.int32, device=xyz.device)
    check(_lib.load().chx_nl_check(_ptr(xyz), _ptr(ref), xyz.shape[0], box, C.c_float(skin), _ptr(flag), _stream()), "chx_nl_check")
    return flag


def pl_build(n: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    w = max(n - 1, 0)
    all_pairs = torch.empty((n, w), dtype=torch.int32, device=device)
    red = torch.empty((n, w), dtype=torch.uint8, device=device)
    _require_device(all_pairs)
    check(_lib.load().chx_pl_build(n, _ptr(all_pairs), _ptr(red), _stream()), "chx_pl_build")
    return all_pairs, red.to(torch.bool)


def pl_calculate(xyz, box: Box, cutoff: Optional[float]):
    xyz = _f32(xyz)
    _require_device(xyz)
    n = xyz.shape[0]
    w = max(n - 1, 0)
    dev = xyz.device
    n_pairs = torch.empty((n,), dtype=torch.int32, device=dev)
    mask = torch.empty((n, w), dtype=torch.int32, device=dev)
    dist = torch.empty((n, w), dtype=torch.float32, device=dev)
    rij = torch.empty((n, w, 3), dtype=torch.float32, device=dev)
    c = -1.0 if cutoff is None else float(cutoff)
    check(_lib.load().chx_pl_calculate(_ptr(xyz), n, box, C.c_float(c), _ptr(n_pairs), _ptr(mask), _ptr(dist), _ptr(rij), _stream()), "chx_pl_calculate")
    return n_pairs, mask, dist, rij


# ---------------------------------------------------------------------------
# force list + LJ kernels
# ---------------------------------------------------------------------------
FL_QUAD = 4  # entries per lane per chunk: a force-list row is stored in chunks of FL_QUAD * g entries


def fl_chunks_for(maxdeg: int, g: int) -> int:
    return max(1, -(-int(maxdeg) // (FL_QUAD * g)))


def _choose_g(maxdeg: int, n: int) -> int:
    """Lanes that cooperate on one particle in the list kernels.

    Measured on MI355X at the loaded clock (round-2 in-process A/B, Langevin steps/s on the rho* = 0.1 fluid; best G and
    the runners-up): N = 8192: 8 (16: -10 %, 32: -17 %); 16384: 8 (4: -4 %, 16: -7 %); 32768: 4 (8: -2 %, 2: -15 %);
    65536: 4 (2: -7 %, 8: -15 %); 131072: 2 (4: -4 %, 1: -11 %); 262144: 1 (2: equal, 4: -4 %).  I.e. about 2^18 lanes
    in flight for the large systems, 2^17 below N = 2^16 and never more than 8 lanes per particle for the sake of the
    lane count alone: a row of the pruned list has ~14 entries and a chunk 4 G slots, so beyond G = 4..8 the extra
    lanes mostly evaluate padding, pay a longer reduction and double the workgroups to dispatch.  No more than ~96
    listed entries per lane (the step kernel mostly walks the pruned list, about half of that).
    """
    env = os.environ.get("CHX_FL_G")
    if env:
        return int(env)
    g = 1  # one lane per particle once that alone fills the chip twice over (N >= 2^19)
    # (dense rows keep the 2^18 target at every size: rho* = 0.8, N = 32768: 8 lanes 82 k steps/s, 4: 77 k, 16: 73 k)
    # N = 262144: round 2 tuned on the lattice start, where one and two lanes were equal; on the EQUILIBRATED fluid (rows
    # of 10..64 entries instead of a narrow band around 31: a wave of 64 one-lane rows waits for its longest) two lanes
    # per particle are worth -5 % on the pair-force kernel and +1 % steps/s (round 3: 58.0-58.1 k against 57.4-57.6 k)
    lanes = (1 << 19) if n >= (1 << 18) else ((1 << 18) if (n >= (1 << 16) or maxdeg > 128) else (1 << 17))
    while g < 8 and n * g < lanes:
        g *= 2
    while g < 8 and maxdeg > 96 * g:  # (96: at rho* = 0.8, N = 262144 four lanes per particle beat eight by 6 %)
        g *= 2  # (never beyond 8: the fused step and Monte Carlo trial kernels are instantiated for 1, 2, 4, 8 -- round 6)
    while g > 1 and g > max(1, maxdeg):
        g //= 2
    return g


class ForceList:
    """Internal full neighbor table (see include/chx.h).

    ``ForceList(nl, n_nbr, perm)`` derives it from a public half list (any builder, exactly the
    listed pairs); ``ForceList.prebuilt(...)`` wraps the table the one-pass cell build emitted.
    """

    @classmethod
    def prebuilt(cls, fl: torch.Tensor, deg: torch.Tensor, perm: torch.Tensor, g: int, chunks: int, maxdeg: int):
        self = cls.__new__(cls)
        self.n = deg.shape[0]
        self.perm, self.inv = perm, None
        self.fl, self.deg, self.g, self.chunks, self.maxdeg = fl, deg, g, chunks, maxdeg
        self.sorted = True
        self.ws = torch.empty((int(_lib.load().chx_lj_ws_count(self.n)),), dtype=torch.float64, device=deg.device)
        return self

    def __init__(self, nl: torch.Tensor, n_nbr: torch.Tensor, perm: Optional[torch.Tensor] = None):
        _require_device(nl, n_nbr, perm)
        lib = _lib.load()
        n, m = nl.shape
        dev = nl.device
        self.n = n
        self.perm = perm
        self.inv = None
        if perm is not None:
            self.inv = torch.empty_like(perm)
            check(lib.chx_invert_perm(_ptr(perm), n, _ptr(self.inv), _stream()), "chx_invert_perm")
        self.deg = torch.empty((n,), dtype=torch.int32, device=dev)
        maxdeg_t = torch.empty((1,), dtype=torch.int32, device=dev)
        check(lib.chx_fl_count(_ptr(nl), _ptr(n_nbr), n, m, _ptr(self.inv), _ptr(self.deg), _ptr(maxdeg_t), _stream()), "chx_fl_count")
        self.maxdeg = int(maxdeg_t.item())  # build-time sync (the reference syncs here too, neighbors.py:709)
        self.g = _choose_g(self.maxdeg, n)
        chunks = fl_chunks_for(self.maxdeg, self.g)
        self.chunks = chunks
        self.fl = torch.zeros((chunks, n, FL_QUAD * self.g), dtype=torch.int32, device=dev)  # a valid index everywhere; chx_fl_fill pads rows with their own index
        cursor = torch.empty((n,), dtype=torch.int32, device=dev)
        check(lib.chx_fl_fill(_ptr(nl), _ptr(n_nbr), n, m, _ptr(self.inv), self.g, _ptr(cursor), _ptr(self.fl), _stream()), "chx_fl_fill")
        rc = lib.chx_fl_sort(_ptr(self.fl), _ptr(self.deg), n, self.g, self.maxdeg, _stream())
        self.sorted = rc == 0
        if rc not in (0, -3):
            check(rc, "chx_fl_sort")
        self.ws = torch.empty((int(lib.chx_lj_ws_count(n)),), dtype=torch.float64, device=dev)

    def pack(self, x3: torch.Tensor, stride: int = 4) -> torch.Tensor:
        """User-order (n,3) -> internal order (n,stride)."""
        x3 = _f32(x3)
        _require_device(x3)
        out = torch.empty((self.n, stride), dtype=torch.float32, device=x3.device)
        check(_lib.load().chx_permute_pack(_ptr(x3), _ptr(self.perm), self.n, _ptr(out), stride, _stream()), "chx_permute_pack")
        return out

    def unpack(self, x: torch.Tensor) -> torch.Tensor:
        """Internal order (n,3|4) -> user order (n,3)."""
        _require_device(x)
        out = torch.empty((self.n, 3), dtype=torch.float32, device=x.device)
        check(_lib.load().chx_permute_unpack(_ptr(x), x.shape[1], _ptr(self.perm), self.n, _ptr(out), _stream()), "chx_permute_unpack")
        return out

    def energy_force(self, x4: torch.Tensor, box: Box, sigma: float, epsilon: float, cutoff: float,
                     want_force: bool = True, want_energy: bool = True):
        """Run the list kernel on internal-order positions; returns (force3_internal, energy[1])."""
        _require_device(x4)
        dev = x4.device
        f3 = torch.empty((self.n, 3), dtype=torch.float32, device=dev) if want_force else None
        e = torch.empty((1,), dtype=torch.float32, device=dev) if want_energy else None
        check(
            _lib.load().chx_lj_fl_energy_force(
                _ptr(x4), _ptr(self.fl), _ptr(self.deg), self.n, self.g, self.chunks, box, C.c_float(sigma),
                C.c_float(epsilon), C.c_float(cutoff), _ptr(f3), _ptr(e), _ptr(self.ws), _stream(),
            ),
            "chx_lj_fl_energy_force",
        )
        return f3, e


def lj_nl_energy_force(xyz: torch.Tensor, nl: torch.Tensor, n_nbr: torch.Tensor, box: Box, sigma: float,
                       epsilon: float, cutoff: float, want_force: bool = True, want_energy: bool = True):
    """Energy/force straight from the public half list (no force list; atomics for the reaction)."""
    xyz = _f32(xyz)
    _require_device(xyz, nl, n_nbr)
    lib = _lib.load()
    n, m = nl.shape
    dev = xyz.device
    f = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_force else None
    e = torch.empty((1,), dtype=torch.float32, device=dev) if want_energy else None
    ws = torch.empty((int(lib.chx_lj_ws_count(n)),), dtype=torch.float64, device=dev)
    check(
        lib.chx_lj_nl_energy_force(_ptr(xyz), _ptr(nl), _ptr(n_nbr), n, m, box, C.c_float(sigma), C.c_float(epsilon),
                                   C.c_float(cutoff), _ptr(e), _ptr(f), _ptr(ws), _stream()),
        "chx_lj_nl_energy_force",
    )
    return f, e


def baoab_step(x: torch.Tensor, v: torch.Tensor, f: torch.Tensor, mass: torch.Tensor, half_dt: float, a: float,
               b: float, kT: float, box: Box, wrap: bool, ref: Optional[torch.Tensor], skin: float, key: PRNGKey,
               step: int, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One in-place BAOAB update on user-order arrays; returns the device rebuild flag (int32[1])."""
    _require_device(x, v, f, mass, ref, noise)
    flag = torch.empty((1,), dtype=torch.int32, device=x.device)
    check(
        _lib.load().chx_baoab_step(_ptr(x), _ptr(v), _ptr(f), _ptr(mass), x.shape[0], box, int(bool(wrap)),
                                   C.c_float(half_dt), C.c_float(a), C.c_float(b), C.c_float(kT), _ptr(ref), C.c_float(skin),
                                   key[0], key[1], int(step) & 0xFFFFFFFF, _ptr(noise), _ptr(flag), _stream()),
        "chx_baoab_step",
    )
    return flag


def lj_allpairs(xyz: torch.Tensor, box: Box, sigma: float, epsilon: float, cutoff: Optional[float],
                want_force: bool = True, want_energy: bool = True):
    xyz = _f32(xyz)
    _require_device(xyz)
    lib = _lib.load()
    n = xyz.shape[0]
    dev = xyz.device
    f = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_force else None
    e = torch.empty((1,), dtype=torch.float32, device=dev) if want_energy else None
    ws = torch.empty((int(lib.chx_lj_ws_count(n)),), dtype=torch.float64, device=dev)
    c = -1.0 if cutoff is None else float(cutoff)
    check(
        lib.chx_lj_allpairs_energy_force(_ptr(xyz), n, box, C.c_float(sigma), C.c_float(epsilon), C.c_float(c), _ptr(f), _ptr(e), _ptr(ws), _stream()),
        "chx_lj_allpairs_energy_force",
    )
    return f, e


# ---------------------------------------------------------------------------
# Langevin
# ---------------------------------------------------------------------------
def init_velocities(mass: torch.Tensor, kT: float, key: PRNGKey, normals: Optional[torch.Tensor] = None) -> torch.Tensor:
    mass = _f32(mass)
    _require_device(mass, normals)
    n = mass.shape[0]
    v = torch.empty((n, 3), dtype=torch.float32, device=mass.device)
    check(_lib.load().chx_init_velocities(_ptr(v), _ptr(mass), n, C.c_float(kT), key[0], key[1], _ptr(normals), _stream()), "chx_init_velocities")
    return v


def jax_normal(key, shape, device=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``jax.random.normal(key, shape)`` for a threefry key (chiron_amd.rng.JaxKey), generated on the device."""
    count = 1
    for d in shape:
        count *= int(d)
    if out is None:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=device or "cuda")
    _require_device(out)
    if out.numel() != count or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("jax_normal: `out` must be a contiguous fp32 tensor of the requested shape")
    check(_lib.load().chx_jax_normal(int(key[0]), int(key[1]), count, _ptr(out), _stream()), "chx_jax_normal")
    return out


def kick(v: torch.Tensor, f: torch.Tensor, mass: torch.Tensor, half_dt: float) -> None:
    _require_device(v, f, mass)
    check(_lib.load().chx_kick(_ptr(v), _ptr(f), _ptr(mass), v.shape[0], C.c_float(half_dt), _stream()), "chx_kick")


def langevin_chunk(params: LangevinParams, step0: int, n_steps: int, kicks_first: int) -> None:
    check(_lib.load().chx_langevin_chunk(C.byref(params), step0, n_steps, kicks_first, _stream()), "chx_langevin_chunk")


def langevin_fused_chunk(params: LangevinParams, x_even: torch.Tensor, x_odd: torch.Tensor, step0: int,
                         n_steps: int, kicks_first: int, do_aoa: int = 1) -> None:
    _require_device(x_even, x_odd)
    check(
        _lib.load().chx_langevin_fused_chunk(C.byref(params), _ptr(x_even), _ptr(x_odd), step0, n_steps, kicks_first, do_aoa, _stream()),
        "chx_langevin_fused_chunk",
    )


_tls = __import__("threading").local()  # (MultiStateSampler may drive two replicas from two threads)


def langevin_fused_run(params: LangevinParams, x_even: torch.Tensor, x_odd: torch.Tensor, step0: int,
                       n_steps: int, kicks_first: int, window: int) -> int:
    """``langevin_fused_chunk`` with a bounded launch queue (include/chx.h: chx_langevin_fused_run).
    Returns the number of launches made: fewer than ``n_steps`` when the host saw the halt word."""
    _require_device(x_even, x_odd)
    out = getattr(_tls, "n_enqueued", None)
    if out is None:
        out = _tls.n_enqueued = C.c_int32(0)
    check(
        _lib.load().chx_langevin_fused_run(C.byref(params), _ptr(x_even), _ptr(x_odd), step0, n_steps, kicks_first, window,
                                           C.byref(out), _stream()),
        "chx_langevin_fused_run",
    )
    return out.value


def _batch_items(entries):
    """ctypes array of chx_batch_item_t for ``entries`` = [(params, x_even, x_odd, step0, n_steps, kicks_first), ...]"""
    arr = (_lib.BatchItem * len(entries))()
    for it, (params, x_even, x_odd, step0, n_steps, kicks_first) in zip(arr, entries):
        it.p = C.pointer(params)
        it.x_even, it.x_odd = x_even.data_ptr(), x_odd.data_ptr()
        it.step0, it.n_steps, it.kicks_first, it.n_enqueued = step0, n_steps, kicks_first, 0
    return arr


def langevin_batch_run(entries, window: int):
    """One launch per step for SEVERAL replicas (include/chx.h: chx_langevin_batch_run).  ``entries``: one
    (params, x_even, x_odd, step0, n_steps, kicks_first) per replica.  Returns the launches each replica took part in --
    fewer than asked for when the host saw some replica's halt word."""
    arr = _batch_items(entries)
    check(_lib.load().chx_langevin_batch_run(arr, len(entries), window, _stream()), "chx_langevin_batch_run")
    return [int(it.n_enqueued) for it in arr]


def langevin_batch_kick(entries) -> None:
    """The force-only launch that ends a run, for several replicas in one dispatch (chx_langevin_batch_kick);
    ``entries`` as for ``langevin_batch_run`` (n_steps is ignored)."""
    arr = _batch_items(entries)
    check(_lib.load().chx_langevin_batch_kick(arr, len(entries), _stream()), "chx_langevin_batch_kick")


# ---------------------------------------------------------------------------
# Debug mode of the step queue (include/chx.h: chx_queue_debug): the deterministic detector for a missing join.
# The C side delays the launches whose outputs are read from outside the queue; this side fills those outputs with
# poison (NaN / -1) when it allocates them, on the stream, and waits for the fill -- so the poison is in place before the
# packet that overwrites it is even written.  A reader that skipped steps_join() then reads poison every time.
# CHX_QUEUE_DEBUG=<microseconds> switches both halves on for a whole process; queue_debug(us) at run time.
# ---------------------------------------------------------------------------
_debug_us = int(__import__("os").environ.get("CHX_QUEUE_DEBUG", "0") or 0)
_debug_skip_final_join = 0  # test hook: that many of the joins that end a run() are skipped (tests/test_gpu_step_queue.py)


def queue_debug(delay_us: Optional[int] = None) -> int:
    """Set (or, with None, read) the step queue's debug delay in microseconds; 0 = off.  Returns the previous value."""
    global _debug_us
    old = _debug_us
    if delay_us is not None:
        _debug_us = max(0, int(delay_us))
        _lib.load().chx_queue_debug(_debug_us)
    return old


def queue_debug_spins() -> int:
    return int(_lib.load().chx_queue_debug_spins())


def fresh(shape, dtype=torch.float32, device=None) -> torch.Tensor:
    """``torch.empty`` for an array that a launch on the step queue is about to write and somebody OUTSIDE the queue will
    read (the state a run() returns, a rebuilt list's arrays).  In debug mode the array starts out as poison."""
    t = torch.empty(shape, dtype=dtype, device=device)
    if _debug_us > 0 and t.is_cuda:
        t.fill_(float("nan") if t.is_floating_point() else -1)
        torch.cuda.current_stream(t.device).synchronize()
    return t


def is_poison(t: torch.Tensor) -> bool:
    """Does ``t`` still hold (any of) the poison ``fresh`` put there?  (Reads on the current stream, no join.)"""
    return bool((torch.isnan(t) if t.is_floating_point() else (t == -1)).any().item())


def steps_join() -> None:
    """Wait on the host until every step launch dispatched for the current stream has completed (include/chx.h:
    chx_steps_join; returns at once when none is outstanding).  The step launches run on a queue of their own, so a
    wait for the STREAM -- an event, ``stream.synchronize()``, a ``.cpu()`` -- does not cover them: whoever is about to
    touch what they write with anything but a libchx call joins first (libchx calls join by themselves)."""
    check(_lib.load().chx_steps_join(_stream()), "chx_steps_join")


def steps_open() -> None:
    """Put the gate up now (include/chx.h: chx_steps_open): the list build and the pack launches that follow ride on the
    step queue with the steps, behind everything the stream holds at this point."""
    check(_lib.load().chx_steps_open(_stream()), "chx_steps_open")


_queue_on = {}


def step_queue_on() -> bool:
    """Do the fused step launches of the calling thread go through the step queue?  The queue belongs to the device it was
    opened on, so the answer is a function of the thread's current device: asked once per device (a ctypes call and a
    lock), then remembered -- run() asks behind its final join, on the path to its return."""
    dev = torch.cuda.current_device()
    on = _queue_on.get(dev)
    if on is None:
        out = (C.c_int64 * 8)()
        check(_lib.load().chx_step_queue_info(out, None), "chx_step_queue_info")
        on = _queue_on[dev] = bool(out[0])
    return on


def step_queue_info() -> dict:
    """Diagnostics of the step queue paired with the current stream (include/chx.h: chx_step_queue_info)."""
    out = (C.c_int64 * 8)()
    lib = _lib.load()
    check(lib.chx_step_queue_info(out, _stream()), "chx_step_queue_info")
    status = lib.chx_step_queue_status()
    return {"on": bool(out[0]), "hot_release": {0: "none", 1: "agent"}.get(int(out[1]), "off"), "kernels": int(out[2]),
            "dispatched": int(out[3]), "joins": int(out[4]), "gates": int(out[5]), "open": bool(out[6]),
            "idle_opens": int(out[7]),
            "status": status.decode() if status else ""}


def step_queue_selfcheck() -> int:
    """Repeat the step queue's start-up check of its fence-free packets now (include/chx.h: chx_step_queue_selfcheck):
    stale words read (0 = fine), -1 = not applicable."""
    return int(_lib.load().chx_step_queue_selfcheck(_stream()))


def ctrl_reset(ctrl: torch.Tensor) -> None:
    _require_device(ctrl)
    check(_lib.load().chx_ctrl_reset(_ptr(ctrl), _stream()), "chx_ctrl_reset")


def publish(src: torch.Tensor, pinned_dst: torch.Tensor, n: int) -> None:
    """Device ints -> host-mapped pinned tensor by a kernel on the current stream (no copy engine)."""
    _require_device(src)
    if not pinned_dst.is_pinned():
        raise _lib.ChxError("publish needs a pinned (host-mapped) destination tensor")
    check(_lib.load().chx_publish(_ptr(src), C.c_void_p(pinned_dst.data_ptr()), n, _stream()), "chx_publish")


def langevin_resume(params: LangevinParams) -> None:
    check(_lib.load().chx_langevin_resume(C.byref(params), _stream()), "chx_langevin_resume")


# ---------------------------------------------------------------------------
# Monte Carlo
# ---------------------------------------------------------------------------
def mc_displace(x: torch.Tensor, sigma_d: float, box: Box, key: PRNGKey, ctr: int = 0,
                subset_mask: Optional[torch.Tensor] = None, normals: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _f32(x)
    _require_device(x, subset_mask, normals)
    out = torch.empty_like(x)
    check(
        _lib.load().chx_mc_displace(_ptr(x), _ptr(out), _ptr(subset_mask), x.shape[0], C.c_float(sigma_d), box, key[0], key[1], ctr, _ptr(normals), _stream()),
        "chx_mc_displace",
    )
    return out


def mc_displace_chunk(params: "_lib.McParams", trial0: int, n_trials: int) -> None:
    """Run ``n_trials`` device-resident displacement trials (include/chx.h: chx_mc_displace_chunk) and wait for them: the
    launches ride on the step queue, and the caller reads the chunk's state and logs with torch (``.cpu()``) next."""
    check(_lib.load().chx_mc_displace_chunk(C.byref(params), trial0, n_trials, _stream()), "chx_mc_displace_chunk")
    steps_join()


def metropolis_select(x_cur, x_prop, log_ratio, u_cur, u_prop, key: PRNGKey, ctr: int, accepted, counters, uniform: float = -1.0) -> None:
    _require_device(x_cur, x_prop, log_ratio, u_cur, u_prop, accepted, counters)
    check(
        _lib.load().chx_metropolis_select(
            _ptr(x_cur), _ptr(x_prop), x_cur.numel(), _ptr(log_ratio), _ptr(u_cur), _ptr(u_prop),
            key[0], key[1], ctr, C.c_float(uniform), _ptr(accepted), _ptr(counters), _stream(),
        ),
        "chx_metropolis_select",
    )
