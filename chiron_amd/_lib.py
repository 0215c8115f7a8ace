"""ctypes binding of ``libchx.so`` (the C ABI declared in ``include/chx.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``chiron_amd.build``.
There is no fallback: if the shared object is missing or a symbol cannot be
resolved, importing the binding raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHX_LIB") or os.path.join(_HERE, "libchx.so")  # CHX_LIB: A/B experiments against another build


class Box(C.Structure):
    _fields_ = [("periodic", C.c_int32), ("lx", C.c_float), ("ly", C.c_float), ("lz", C.c_float)]


class LangevinParams(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("g", C.c_int32),
        ("x4", C.c_void_p),
        ("v", C.c_void_p),
        ("f", C.c_void_p),
        ("ref4", C.c_void_p),
        ("mass", C.c_void_p),
        ("perm", C.c_void_p),
        ("fl", C.c_void_p),
        ("deg", C.c_void_p),
        ("box", Box),
        ("wrap", C.c_int32),
        ("skin", C.c_float),
        ("half_dt", C.c_float),
        ("a", C.c_float),
        ("b", C.c_float),
        ("kT", C.c_float),
        ("sigma", C.c_float),
        ("epsilon", C.c_float),
        ("cutoff", C.c_float),
        ("k0", C.c_uint32),
        ("k1", C.c_uint32),
        ("noise", C.c_void_p),
        ("noise_step0", C.c_int64),
        ("ctrl", C.c_void_p),
        ("ws", C.c_void_p),
        ("external_force", C.c_int32),
        ("fl_chunks", C.c_int32),
        ("fl_inner", C.c_void_p),
        ("deg_inner", C.c_void_p),
        ("ref_inner4", C.c_void_p),
        ("inner_skin", C.c_float),
        ("prune_every", C.c_int32),
        ("prune_phase", C.c_int64),
        ("host_ctrl", C.c_void_p),
        ("snap4", C.c_void_p),
        ("ev_log_cap", C.c_int32),
        ("ev_log", C.c_void_p),
        ("final_x3", C.c_void_p),
        ("final_v3", C.c_void_p),
        ("final_ref3", C.c_void_p),
    ]


CTRL_WORDS = 16  # include/chx.h: CHX_CTRL_WORDS


class McParams(C.Structure):
    """``chx_mc_t`` (include/chx.h)."""

    _fields_ = [
        ("n", C.c_int32),
        ("g", C.c_int32),
        ("fl_chunks", C.c_int32),
        ("reserved", C.c_int32),
        ("cur4", C.c_void_p),
        ("prop4", C.c_void_p),
        ("ref4", C.c_void_p),
        ("perm", C.c_void_p),
        ("subset_mask", C.c_void_p),
        ("fl", C.c_void_p),
        ("deg", C.c_void_p),
        ("box", Box),
        ("skin", C.c_float),
        ("sigma_d", C.c_float),
        ("sigma", C.c_float),
        ("epsilon", C.c_float),
        ("cutoff", C.c_float),
        ("inner_skin", C.c_float),
        ("beta", C.c_double),
        ("state", C.c_void_p),
        ("energy", C.c_void_p),
        ("accept_log", C.c_void_p),
        ("energy_log", C.c_void_p),
        ("ws", C.c_void_p),
        ("k0", C.c_uint32),
        ("k1", C.c_uint32),
        ("keytab", C.c_void_p),
        ("keytab_h", C.c_void_p),
        ("chain_out_h", C.c_void_p),
        ("fli", C.c_void_p),
        ("cnti", C.c_void_p),
        ("refi4", C.c_void_p),
    ]


# indices into the chx_mc_t state block (include/chx.h)
MC_CUR, MC_ACCEPTED, MC_PROPOSED, MC_HALT, MC_KEY0, MC_KEY1, MC_SLOT_WORDS, MC_STATE_WORDS = 0, 1, 2, 3, 4, 5, 8, 16
MC_REFRESH, MC_REFRESHES = 6, 7  # slot 0: inner-list bookkeeping of the kernels

class BuildRiders(C.Structure):
    """chx_build_riders_t (include/chx.h): work that rides on the launches of chx_nl_build_cell_fl."""

    _fields_ = [
        ("xyz_copy_out", C.c_void_p),
        ("v3", C.c_void_p),
        ("mass", C.c_void_p),
        ("ref3", C.c_void_p),
        ("x4_out", C.c_void_p),
        ("v_out", C.c_void_p),
        ("mass_out", C.c_void_p),
        ("ref4_out", C.c_void_p),
        ("ctrl_reset", C.c_void_p),
        ("ctrl_keep", C.c_int32),
        ("publish_host_dst8", C.c_void_p),
        ("cand_list", C.c_void_p),
        ("cand_len", C.c_void_p),
        ("cand_cap", C.c_int32),
        ("cell_slots", C.c_void_p),
        ("cell_slot_cap", C.c_int32),
    ]


class BatchItem(C.Structure):
    """include/chx.h: chx_batch_item_t"""

    _fields_ = [
        ("p", C.POINTER(LangevinParams)),
        ("x_even", C.c_void_p),
        ("x_odd", C.c_void_p),
        ("step0", C.c_int64),
        ("n_steps", C.c_int32),
        ("kicks_first", C.c_int32),
        ("n_enqueued", C.c_int32),
        ("reserved", C.c_int32),
    ]


BUILD_STATUS_WORDS = 8  # include/chx.h: chx_nl_build_cell_fl, status8

_P = C.c_void_p
_I32 = C.c_int32
_I64 = C.c_int64
_U32 = C.c_uint32
_F = C.c_float

# name -> (restype, argtypes); must list every symbol include/chx.h declares
SIGNATURES = {
    "chx_version": (C.c_int, []),
    "chx_last_error": (C.c_char_p, []),
    "chx_source_hash": (C.c_char_p, []),
    "chx_sizeof_langevin_t": (_I64, []),
    "chx_sizeof_mc_t": (_I64, []),
    "chx_sizeof_build_riders_t": (_I64, []),
    "chx_cutoff_threshold_h": (_F, [_F]),
    "chx_philox4x32_h": (None, [_U32, _U32, _U32, _U32, _U32, _U32, C.POINTER(_U32)]),
    "chx_displacement": (C.c_int, [_P, _P, _I64, Box, _P, _P, _P]),
    "chx_wrap": (C.c_int, [_P, _P, _I64, Box, _P]),
    "chx_nl_build_nsq": (C.c_int, [_P, _I32, _I32, Box, _F, _P, _P, _P, _P]),
    "chx_cell_grid_h": (_I64, [Box, _F, _I32, C.POINTER(_I32)]),
    "chx_nl_build_cell": (
        C.c_int,
        [_P, _I32, _I32, Box, _F, C.POINTER(_I32), _P, _P, _P, _P, _P, _P, _P, _P, _P],
    ),
    "chx_nl_build_cell_fl": (
        C.c_int,
        [_P, _I32, _I32, Box, _F, C.POINTER(_I32), _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _P, _P, C.POINTER(BuildRiders), _P],
    ),
    "chx_cell_bin_fl": (C.c_int, [_P, _I32, Box, _F, C.POINTER(_I32), _P, _P, _P, _P, _P, _P, C.POINTER(BuildRiders), _P]),
    "chx_nl_rows_cell_fl": (
        C.c_int,
        [_I32, _I32, Box, _F, C.POINTER(_I32), _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P],
    ),
    "chx_nl_finish_padding": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "chx_pack_langevin": (C.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _I32, _P, _P, _P]),
    "chx_validate_snap": (C.c_int, [_P, _P, _I32, Box, _F, _I32, _P, _P, _P]),
    "chx_pack_validate_langevin": (C.c_int, [Box, _F, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "chx_nl_calculate": (C.c_int, [_P, _P, _P, _I32, _I32, Box, _F, _P, _P, _P, _P, _P]),
    "chx_nl_check": (C.c_int, [_P, _P, _I32, Box, _F, _P, _P]),
    "chx_pl_build": (C.c_int, [_I32, _P, _P, _P]),
    "chx_pl_calculate": (C.c_int, [_P, _I32, Box, _F, _P, _P, _P, _P, _P]),
    "chx_fl_count": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P, _P]),
    "chx_fl_fill": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _P, _P, _P]),
    "chx_fl_sort": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "chx_permute_pack": (C.c_int, [_P, _P, _I32, _P, _I32, _P]),
    "chx_permute_unpack": (C.c_int, [_P, _I32, _P, _I32, _P, _P]),
    "chx_unpack_langevin": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P]),
    "chx_invert_perm": (C.c_int, [_P, _I32, _P, _P]),
    "chx_lj_ws_count": (_I64, [_I32]),
    "chx_lj_fl_energy_force": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, Box, _F, _F, _F, _P, _P, _P, _P]),
    "chx_lj_allpairs_energy_force": (C.c_int, [_P, _I32, Box, _F, _F, _F, _P, _P, _P, _P]),
    "chx_init_velocities": (C.c_int, [_P, _P, _I32, _F, _U32, _U32, _P, _P]),
    "chx_jax_normal": (C.c_int, [_U32, _U32, _I64, _P, _P]),
    "chx_kick": (C.c_int, [_P, _P, _P, _I32, _F, _P]),
    "chx_langevin_chunk": (C.c_int, [C.POINTER(LangevinParams), _I64, _I32, _I32, _P]),
    "chx_langevin_resume": (C.c_int, [C.POINTER(LangevinParams), _P]),
    "chx_langevin_fused_chunk": (C.c_int, [C.POINTER(LangevinParams), _P, _P, _I64, _I32, _I32, _I32, _P]),
    "chx_langevin_fused_run": (C.c_int, [C.POINTER(LangevinParams), _P, _P, _I64, _I32, _I32, _I32, C.POINTER(C.c_int32), _P]),
    "chx_langevin_batch_run": (C.c_int, [C.POINTER(BatchItem), _I32, _I32, _P]),
    "chx_langevin_batch_kick": (C.c_int, [C.POINTER(BatchItem), _I32, _P]),
    "chx_sizeof_batch_item_t": (_I64, []),
    "chx_set_poll_sleep_ns": (_I64, [_I64]),
    "chx_poll_stats": (C.c_int, [C.POINTER(_I64), _I32]),
    "chx_steps_join": (C.c_int, [_P]),
    "chx_steps_open": (C.c_int, [_P]),
    "chx_step_queue_info": (C.c_int, [C.POINTER(_I64), _P]),
    "chx_step_queue_status": (C.c_char_p, []),
    "chx_step_queue_selfcheck": (C.c_int, [_P]),
    "chx_step_queue_selfcheck_runs": (_I64, []),
    "chx_queue_debug": (C.c_int, [_I32]),
    "chx_queue_debug_spins": (_I64, []),
    "chx_ctrl_reset": (C.c_int, [_P, _P]),
    "chx_publish": (C.c_int, [_P, _P, _I32, _P]),
    "chx_mc_displace": (C.c_int, [_P, _P, _P, _I32, _F, Box, _U32, _U32, _U32, _P, _P]),
    "chx_scale": (C.c_int, [_P, _P, _I64, _F, _P]),
    "chx_metropolis_select": (
        C.c_int,
        [_P, _P, _I64, _P, _P, _P, _U32, _U32, _U32, _F, _P, _P, _P],
    ),
    "chx_mc_key_chain_h": (C.c_int, [_U32, _U32, _I32, C.POINTER(_U32), C.POINTER(_U32)]),
    "chx_mc_displace_chunk": (C.c_int, [_P, _I64, _I32, _P]),
    "chx_lj_nl_energy_force": (C.c_int, [_P, _P, _P, _I32, _I32, Box, _F, _F, _F, _P, _P, _P, _P]),
    "chx_baoab_step": (
        C.c_int,
        [_P, _P, _P, _P, _I32, Box, _I32, _F, _F, _F, _F, _P, _F, _U32, _U32, _U32, _P, _P, _P],
    ),
}

ABI_VERSION = 5  # include/chx.h: CHX_ABI_VERSION (history there)

_lib: Optional[C.CDLL] = None


class ChxError(RuntimeError):
    pass


def _share_torch_hip_runtime() -> None:
    """libchx must drive the SAME HIP runtime instance that owns torch's streams and allocations.

    The PyTorch-ROCm wheel bundles its own libamdhip64.so while libchx.so is linked against the
    system one (/opt/rocm): with both in the process, libchx's HIP calls have to bind to the copy
    torch uses, or the stream handles it is given belong to another runtime.  Putting torch's
    copy in the global symbol scope and initialising torch's device context BEFORE libchx (and its
    registration of the gfx950 code objects) is loaded guarantees that, whatever the import order
    of the caller (found with build()+smoke() in one process: hipMemsetAsync failed).
    """
    try:
        import torch
    except ImportError:  # pragma: no cover - torch is the memory/stream provider of this package
        return
    if getattr(torch.version, "hip", None):
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    if torch.cuda.is_available():
        torch.cuda.init()


def load() -> C.CDLL:
    """Load libchx.so and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ChxError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.chx_version() != ABI_VERSION:
        raise ChxError(f"libchx ABI version {lib.chx_version()} != {ABI_VERSION} (include/chx.h: CHX_ABI_VERSION)")
    # the structs are mirrored by hand above: a layout drift would corrupt kernel arguments silently
    for name, mirror in (("chx_sizeof_langevin_t", LangevinParams), ("chx_sizeof_mc_t", McParams),
                         ("chx_sizeof_build_riders_t", BuildRiders), ("chx_sizeof_batch_item_t", BatchItem)):
        if getattr(lib, name)() != C.sizeof(mirror):
            raise ChxError(f"{name}() = {getattr(lib, name)()} but ctypes mirror {mirror.__name__} is {C.sizeof(mirror)} bytes")
    # a library built from other sources than the ones beside it is refused (CHX_LIB: deliberate A/B runs).  A
    # binary-only install -- libchx.so shipped without csrc/ -- has nothing to compare against and loads as it is;
    # the comparison is made once per process (8 ranks: 8 x 13 small files, not 8 x every load() call).
    if not os.environ.get("CHX_LIB") and os.environ.get("CHX_ALLOW_STALE_LIB", "0") != "1":
        from . import build as _build

        if _build.sources():
            have, want = lib.chx_source_hash().decode(), _build.source_hash()
            if have != want:
                raise ChxError(f"{LIB_PATH} was built from other sources (hash {have[:16]}, tree {want[:16]}): "
                               "run `python -c 'import __graft_entry__ as g; g.build()'`")
            # the code objects the step queue dispatches from (libchx_step.hsaco, libchx_k_*.hsaco) belong to the same
            # build: without them the library would quietly fall back to HIP launches (include/chx.h, "The step queue")
            if os.environ.get("CHX_AQL", "1") != "0" and not _build.step_object_current():
                raise ChxError(f"{_build.STEP_OUT} (the step queue's code objects) is missing or was built from other sources: "
                               "run `python -c 'import __graft_entry__ as g; g.build()'` (CHX_AQL=0 runs without the queue)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().chx_last_error()
        raise ChxError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
