import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _two_zone(dim):
    def fn(c):
        rho = np.abs(c[:, 0]) if dim == 2 else np.hypot(c[:, 0], c[:, 1])
        z = c[:, dim - 1]
        m = np.where(z > 1.0, 2, 1)
        m[rho < 0.1] = 0
        return m.astype(np.int32)
    return fn


@pytest.fixture(scope="session")
def mesh2d():
    from remo3d_amd.meshgen import make_mesh
    return make_mesh(2, 50.0, [0.0, 0.1, -0.1], scale=2.0, material_fn=_two_zone(2), seed=0)


@pytest.fixture(scope="session")
def mesh3d():
    from remo3d_amd.meshgen import make_mesh
    return make_mesh(3, 50.0, [0.0, 0.1, -0.1], scale=8.0, material_fn=_two_zone(3), seed=0)


SIGMA3 = [1.0, 0.1, 0.02]


@pytest.fixture(scope="session")
def gpu_ctx():
    from remo3d_amd import solver
    ctx = solver.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def examples_dir():
    """Input / output DATA files of the reference's Examples (CC BY 4.0, README.md:30) kept as fixtures."""
    return os.path.join(ROOT, "tests", "golden", "examples")
