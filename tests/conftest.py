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


@pytest.fixture(scope="session")
def size_L_case():
    """Two batches of the HEADLINE workload (bench.py default: BM3 dip 30, both tools, 100 depths, mesh size L = 1.2 x the reference's
    size field: ~370 k tetrahedra, ~1.7 M unknowns) - batch 0 (lateral tool only) and batch 20 (normal and lateral) - with every one of
    their ten right-hand sides, and one extra right-hand side of the property test, going through the CPU oracle in threads started
    HERE (the C calls release the GIL): ~1 minute of a core each, so they run beside the GPU work of whichever tests come next and
    are waited for where they are compared."""
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
    import bench
    from oracle.fem_oracle import lib, solve_batch as oracle_batch
    lib()
    scale = bench.SIZES["L"]
    with ProcessPoolExecutor(max_workers=2, mp_context=multiprocessing.get_context("spawn")) as pp:      # two distinct meshes, ~20 s of a core each
        work = [f.result()[0] for f in [pp.submit(bench._build_some, (100, scale, "lattice", [bi])) for bi in (0, 20)]]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
    pool = ThreadPoolExecutor(max_workers=max(2, min(11, cores - 2)))

    def one(w, z, I, e, rtol):
        return oracle_batch(w["mesh"], w["sigma"], [0, len(z)], list(z), list(I), [0, len(e)], list(e), condense=True, rtol=rtol, maxit=20000)
    futs = {"properties": pool.submit(one, work[0], [0.0], [1.0], [0.4, 6.4], 1e-11)}
    for bi, w in enumerate(work):
        for k, ((z, I), e) in enumerate(zip(w["sources"], w["evals"])):
            futs[(bi, k)] = pool.submit(one, w, z, I, e, 1e-10)
    yield dict(work=work, futs=futs, scale=scale)
    pool.shutdown(wait=False, cancel_futures=True)
