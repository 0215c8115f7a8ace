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
