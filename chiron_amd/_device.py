"""Device/tensor plumbing: where arrays live and how host inputs become tensors.

PyTorch-ROCm is used only for device memory, streams and ``torch.distributed``;
every number on the hot path is produced by the HIP kernels in ``csrc/``.

Host-side conversions are done with NumPy on purpose: a ``torch`` CPU op on a few
hundred thousand elements wakes torch's whole intra-op thread pool (one thread per
visible core), and in a CPU-quota'd container that burst gets the process throttled
by the scheduler for tens of milliseconds -- long enough to starve the launch loop.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

_FORCED_DEVICE: Optional[torch.device] = None

_NP_DTYPE = {
    torch.float32: np.float32,
    torch.float64: np.float64,
    torch.int32: np.int32,
    torch.int64: np.int64,
    torch.uint8: np.uint8,
    torch.bool: np.bool_,
}


def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(device_index: int, sysfs: str = "/sys"):
    """CPUs of the NUMA node GPU ``device_index`` hangs off, read from sysfs without touching the GPU: the KFD
    topology lists the compute nodes this process may use in the order the runtime enumerates them (plain integer
    lists in ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES are applied on top), each with its PCI location, and the PCI
    device knows its ``local_cpulist``.  None when any of that is not there or not understood."""
    import os

    try:
        base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
        gpus = []
        for name in sorted((d for d in os.listdir(base) if d.isdigit()), key=int):
            try:
                with open(os.path.join(base, name, "properties")) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
            v = os.environ.get(var, "").strip()
            if v:
                idx = [t.strip() for t in v.split(",")]
                if not all(t.isdigit() for t in idx):
                    return None
                gpus = [gpus[int(t)] for t in idx if int(t) < len(gpus)]
        if not 0 <= device_index < len(gpus):
            return None
        loc, dom = int(gpus[device_index]["location_id"]), int(gpus[device_index].get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        with open(os.path.join(sysfs, "bus/pci/devices", bdf, "local_cpulist")) as fh:
            cpus = _parse_cpulist(fh.read())
        return cpus or None
    except (OSError, ValueError, KeyError, IndexError):
        return None


def pin_process_cpus(width: int = 4, slot: Optional[int] = None):
    """Restrict this process -- and every thread created from here on, the HIP runtime's helpers included -- to
    ``width`` neighbouring CPUs around the one it is running on.  Call it before the first GPU call.

    Why: the Langevin loop issues one kernel launch every ~5 us from one thread, and the HIP runtime's launch path
    shares a lock with one of its helper threads.  On a many-core host (256 hardware threads on the MI355X boxes)
    the scheduler is free to put the two on far-apart cores, and one process in three then launches in a slower
    regime (thousands of futex waits per 100 000 launches, 3-12 % fewer steps per second).  A
    bare HIP launch loop pinned to two CPUs was fast in 8 processes of 8, unpinned in 4 of 8
    (round-2 sweep of a bare HIP launch probe).  ``CHX_PIN_CPUS=0`` turns it off, ``CHX_PIN_CPUS=<n>`` sets the width.
    ``slot`` (default: ``LOCAL_RANK``, else 0) = the GPU this process drives: the group is taken from the CPUs of that
    GPU's NUMA node when sysfs tells (``gpu_local_cpus``), and the ranks of one node get different groups.
    Returns the CPU list now in force (None if nothing was changed)."""
    import os

    req = os.environ.get("CHX_PIN_CPUS", "")
    if req == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    if req.isdigit():
        width = int(req)
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) <= width:
        return None
    try:
        here = os.sched_getcpu() if hasattr(os, "sched_getcpu") else allowed[0]
    except OSError:  # pragma: no cover
        here = allowed[0]
    if slot is None:
        lr = os.environ.get("LOCAL_RANK", "0")
        slot = int(lr) if lr.isdigit() else 0
    near = gpu_local_cpus(slot)
    if near:
        near = sorted(set(near) & set(allowed))
        if len(near) > width:
            allowed = near
    groups = len(allowed) // width
    if "LOCAL_RANK" in os.environ:
        # ranks of one node: group = local rank, whatever CPU each happens to start on (two ranks on one group
        # of four CPUs, each with a polling launch thread, lose a quarter of their rate)
        at = slot % groups * width
    else:
        at = ((allowed.index(here) if here in allowed else 0) // width + slot) % groups * width
    cpus = allowed[at:at + width]
    try:
        os.sched_setaffinity(0, cpus)  # (the calling thread; threads it creates from now on inherit)
    except OSError:  # a sandbox that does not allow it: run unpinned
        return None
    try:  # threads that exist already (an imported library's pools): idle, but keep the process in one place
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
    except OSError:
        pass
    return cpus


def launch_regime_probe(run_steps, launches: int = 20000, threshold: int = 150) -> dict:
    """Is this process in the HIP runtime's slow launch regime?  (one fresh process in about
    eight launches 3-12 % slower for its whole life -- its launching thread collides with a helper thread of the
    runtime, ~10 us per collision -- whatever the pinning, the poll policy or the runtime's environment knobs.)

    ``run_steps(n)`` must push ``n`` step launches through ``chx_langevin_fused_run`` from the calling thread and
    return when they are enqueued or done.  While it runs the launch window busy-waits instead of sleeping, so the only
    voluntary context switches the launching thread can make are futex waits inside the runtime: a fast process makes
    0-5 of them per 20 000 launches, a mildly slow one 400-700, a badly slow one 900-1800 (round-2 measurements).
    Returns {"voluntary_switches", "launches", "slow"}.  DIAGNOSTIC ONLY: measured over 84 processes in round 3
    (profiles/r03_launch_regime.md) the count does not predict which processes run slow under the sleeping poll, and
    nothing in this repository acts on it.  What bench.py's launcher does act on is a rank's own stand-alone RATE
    (bench.requalify, profiles/r04_rank_requalification.md): the parent replaces a slow rank by a FRESH child process
    before the process group forms.  A process that has initialised the GPU must never be re-exec'ed (os.exec*) to
    "start again": on this pool that takes the machine down; exit non-zero and let an outer launcher start a new process,
    or spawn a new child.  Side effect while it runs: the poll policy is PROCESS-GLOBAL (chx_set_poll_sleep_ns), so
    every other thread launching through chx_langevin_fused_run busy-waits too until the probe returns."""
    import threading

    from . import _lib

    lib = _lib.load()

    def switches() -> int:
        try:
            with open(f"/proc/self/task/{threading.get_native_id()}/status") as fh:
                for line in fh:
                    if line.startswith("voluntary_ctxt_switches"):
                        return int(line.split()[1])
        except (OSError, ValueError):
            pass
        return -1

    prev = lib.chx_set_poll_sleep_ns(0)
    try:
        before = switches()
        run_steps(int(launches))
        after = switches()
    finally:
        lib.chx_set_poll_sleep_ns(prev)
    count = after - before if before >= 0 and after >= 0 else -1
    return {"voluntary_switches": count, "launches": int(launches), "slow": count > threshold * launches / 20000.0}


def default_device() -> torch.device:
    """``cuda`` (ROCm) when a GPU is visible, else ``cpu`` (host logic/tests only)."""
    if _FORCED_DEVICE is not None:
        return _FORCED_DEVICE
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def set_default_device(device) -> None:
    global _FORCED_DEVICE
    _FORCED_DEVICE = None if device is None else torch.device(device)


def as_tensor(x: Any, dtype=torch.float32, device=None) -> torch.Tensor:
    """Convert lists / numpy arrays / tensors to a contiguous tensor on the device."""
    if (device is None and _FORCED_DEVICE is None and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == dtype
            and x.is_contiguous()):
        return x  # already what the kernels take (the common case on the hot path)
    if device is None:
        device = x.device if isinstance(x, torch.Tensor) and _FORCED_DEVICE is None and x.is_cuda else default_device()
    if isinstance(x, torch.Tensor):
        if x.is_cuda or x.dtype == dtype or dtype not in _NP_DTYPE:
            return x.to(device=device, dtype=dtype).contiguous()
        x = x.detach().numpy()
    arr = np.ascontiguousarray(np.asarray(x), dtype=_NP_DTYPE.get(dtype))
    return torch.from_numpy(arr).to(device).contiguous()


def is_array(x: Any) -> bool:
    return isinstance(x, (torch.Tensor, np.ndarray))
