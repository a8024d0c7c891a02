"""The CPU ORACLE against numbers the REFERENCE holds: the same `Model` sweep as the product (tool tables, batching, windowing,
conforming meshes, Ra formula) with oracle/fem_oracle.c as the solver backend (tests/oracle_backend.py), compared with the
reference's committed Example_01 log.  This is the one external pin the oracle has (NGSolve is not installable: parity with
NGSolve itself stays unpinned); the tolerance is a mesh tolerance (in-repo Delaunay mesh vs Netgen).
A sample of depths: the oracle's Jacobi-PCG needs ~3500 steps per 2D right-hand side."""
import os

import numpy as np

from oracle_backend import OracleContext


def test_oracle_reproduces_the_reference_example_01_log(examples_dir):
    from remo3d_amd.model import Model
    ex = os.path.join(examples_dir, "Example_01")
    tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
    depths = np.arange(0, 25.1, 5.0)
    gold = np.loadtxt(os.path.join(ex, "Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)
    m = Model(tools)
    m.set_model_parameters(os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"))
    m.initialize_workers(cpu_workers=1, gpu_workers=4, context_factory=OracleContext)     # four oracle "contexts" = four host threads
    m.simulate_logs(depths, verbose=False, mesh_scale=1.0)      # the coarse size field keeps the oracle's Jacobi-PCG to seconds; the GPU tests use the default
    m.shutdown_workers()
    assert m.timing["failed_batches"] == 0 and m.timing["not_converged"] == 0
    rows = np.rint(depths / 0.1).astype(int)
    rel = np.array([np.abs(m.logs[t][:, 1] - gold[rows, 1 + i]) / gold[rows, 1 + i] for i, t in enumerate(tools)])
    print("oracle vs the reference's Example_01 log at %d points: median %.2e, max %.2e" % (rel.size, np.median(rel), rel.max()))
    assert np.all(np.isfinite(rel))
    assert np.median(rel) < 1e-3 and rel.max() < 5e-3
