import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_literals.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def methane_coords():
    return np.load(os.path.join(ROOT, "tests", "golden", "methane_coords.npy"))


def cube_coordinates(n_xyz=2, scale_factor=2.0):
    """jnp.mgrid[0:n,0:n,0:n]*scale/n reshaped to (n^3, 3) -- chiron/tests/test_pairs.py:244-250."""
    mesh = np.mgrid[0:n_xyz, 0:n_xyz, 0:n_xyz] * scale_factor / n_xyz
    return np.stack(mesh.reshape(3, -1), axis=1).astype(np.float32)
