"""In-memory reporters with the reference's ``report`` / ``flush_buffer`` /
``get_property`` surface (chiron/reporters.py:47-203).

The reference writes HDF5 (h5py) and XTC (mdtraj); neither library is available
here and file I/O is outside the hot path, so these reporters keep the same
dict-of-lists buffer and expose it through the same calls.  Device tensors are
copied to host ``numpy`` arrays when reported (that copy is the only host sync a
reporter introduces).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch


class BaseReporter:
    _directory = None

    @classmethod
    def set_directory(cls, directory: str):
        cls._directory = directory

    @classmethod
    def get_directory(cls):
        from pathlib import Path

        return Path.cwd() if cls._directory is None else Path(cls._directory)


def _to_host(value):
    if isinstance(value, torch.Tensor):
        return value.detach().cpu().numpy()
    return value


class _SimulationReporter:
    def __init__(self, file_name: str = "reporter", buffer_size: int = 10):
        self.workdir = BaseReporter.get_directory()
        self.file_path_base = self.workdir / f"{file_name}"
        self.log_file_path = self.file_path_base.with_suffix(".npz")
        self.buffer_size = buffer_size
        self.buffer: Dict[str, List] = {}
        self._store: Dict[str, List] = {}
        self._default_properties: List[str] = []

    @property
    def properties_to_report(self):
        return self._default_properties

    @properties_to_report.setter
    def properties_to_report(self, properties: List[str]):
        self._default_properties = properties

    def get_available_keys(self):
        return list(set(self._store) | set(self.buffer))

    def report(self, data_dict):
        for key, value in data_dict.items():
            self.buffer.setdefault(key, []).append(_to_host(value))
        if self.buffer and all(len(v) > self.buffer_size for v in self.buffer.values()):
            self.flush_buffer()

    def flush_buffer(self) -> None:
        for key, values in self.buffer.items():
            if values:
                self._store.setdefault(key, []).extend(values)
        self.buffer = {key: [] for key in self.buffer}

    def reset_reporter_file(self):
        self._store = {}
        self.buffer = {}

    def get_property(self, name: str) -> Optional[np.ndarray]:
        values = list(self._store.get(name, [])) + list(self.buffer.get(name, []))
        if not values:
            return None
        return np.array(values)

    def save(self, path=None) -> str:
        """Dump everything reported so far to one ``.npz``: one array per key, first axis = frames -- the layout of
        the reference's HDF5 log (one resizable dataset per key, chiron/reporters.py:133-144).  A later ``save``
        rewrites the file with every frame reported so far, so the file always equals what the reference's
        append-on-flush log would hold."""
        import os

        path = str(path or self.log_file_path)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        self.flush_buffer()
        np.savez(path, **{k: np.array(v) for k, v in self._store.items()})
        return path

    @staticmethod
    def load(path) -> Dict[str, np.ndarray]:
        """Read a file written by ``save`` back: {key: array with first axis = frames} (what reading the reference's
        log with h5py gives, chiron/reporters.py:176-203)."""
        with np.load(str(path), allow_pickle=False) as f:
            return {k: f[k] for k in f.files}


class LangevinDynamicsReporter(_SimulationReporter):
    _name = "langevin_reporter"

    def __init__(self, file_name: str = None, buffer_size: int = 1, topology=None, name: str = None):
        # (first parameter named as in chiron/reporters.py:303-308; `name=` was this package's spelling until round 3)
        super().__init__(file_name or name or self._name, buffer_size)
        self.topology = topology
        self.xtc_file_path = None
        self._default_properties = ["positions", "box_vectors", "u_kn", "state_index", "step"]

    @classmethod
    def get_name(cls):
        return cls._name


class MCReporter(_SimulationReporter):
    _name = "mc_reporter"

    def __init__(self, file_name=None, buffer_size: int = 1, name=None):
        # (chiron/reporters.py:278)
        if isinstance(file_name, int):  # the reference is also called as MCReporter(1)
            file_name, buffer_size = None, file_name
        super().__init__(file_name or name or self._name, buffer_size)

    @classmethod
    def get_name(cls):
        return cls._name


class MultistateReporter(_SimulationReporter):
    _name = "multistate_reporter"

    def __init__(self, file_name: str = None, buffer_size: int = 1, name: str = None):
        # (chiron/reporters.py:220-224)
        super().__init__(file_name or name or self._name, buffer_size)
        self._default_properties = ["positions", "box_vectors", "u_kn", "state_index", "step"]

    @classmethod
    def get_name(cls):
        return cls._name
