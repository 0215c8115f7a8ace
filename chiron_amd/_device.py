"""Device/tensor plumbing: where arrays live and how host inputs become tensors.

PyTorch-ROCm is used only for device memory, streams and ``torch.distributed``;
every number on the hot path is produced by the HIP kernels in ``csrc/``.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

_FORCED_DEVICE: Optional[torch.device] = None


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
    if device is None:
        device = x.device if isinstance(x, torch.Tensor) and _FORCED_DEVICE is None and x.is_cuda else default_device()
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=dtype)
    else:
        t = torch.as_tensor(np.asarray(x), dtype=dtype).to(device)
    return t.contiguous()


def is_array(x: Any) -> bool:
    return isinstance(x, (torch.Tensor, np.ndarray))
