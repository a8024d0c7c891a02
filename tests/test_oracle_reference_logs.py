"""The CPU ORACLE against numbers the REFERENCE holds: the same `Model` sweep as the product (tool tables, batching, windowing,
conforming meshes, Ra formula) with oracle/fem_oracle.c as the solver backend (tests/oracle_backend.py), compared with the logs
the reference committed (Example_01, Example_02, thin-bedded benchmark).  These are the external pins the oracle has (NGSolve is
not installable: parity with NGSolve itself stays unpinned); the tolerances are MESH tolerances (in-repo
Delaunay meshes vs Netgen).

Two tiers.  The default CPU suite runs ~400 points on the coarse size field (mesh_scale 1.0: the oracle's Jacobi-PCG needs
~3500 steps per 2D right-hand side, so this is what fits a few minutes on 8 cores) and asserts median / p99.  The same sweeps at the
DEFAULT mesh scale (0.35, the one `Model` uses and the GPU suite meets the logs with) take ~15 minutes of 8 cores: run with
REMO_ORACLE_FULL=1 they rewrite profiles/r03_oracle_vs_reference_logs_default_scale.json, and the default suite asserts on that
committed record."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle_backend import OracleContext

EX01_TOOLS = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
TB_TOOLS = ["A0.4M0.1N", "A1.0M0.1N", "A2.0M0.5N", "A8.0M1.0N"]
RECORD = os.path.join(ROOT, "profiles", "r03_oracle_vs_reference_logs_default_scale.json")
THREADS = max(2, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))


def _pcts(rel):
    rel = np.asarray(rel).ravel()
    return dict(points=int(rel.size), median=float(np.median(rel)), p90=float(np.percentile(rel, 90)), p99=float(np.percentile(rel, 99)), max=float(rel.max()))


def _oracle_sweep(tools, depths, gold_rows, gold_path, formation, borehole, set_kw=None, **kw):
    from remo3d_amd.model import Model
    gold = np.loadtxt(gold_path, skiprows=2)
    m = Model(tools)
    m.set_model_parameters(formation, borehole, **(set_kw or {}))
    m.initialize_workers(cpu_workers=1, gpu_workers=THREADS, context_factory=OracleContext)     # oracle "contexts" = host threads
    m.simulate_logs(depths, verbose=False, **kw)
    m.shutdown_workers()
    assert m.timing["failed_batches"] == 0 and m.timing["not_converged"] == 0, m.timing
    rel = np.array([np.abs(m.logs[t][:, 1] - gold[gold_rows, 1 + i]) / gold[gold_rows, 1 + i] for i, t in enumerate(tools)])
    assert np.all(np.isfinite(rel))
    return rel


def _cases(examples_dir):
    ex1, ex2 = os.path.join(examples_dir, "Example_01"), os.path.join(examples_dir, "Example_02")
    tb = os.path.join(examples_dir, "Benchmark models", "Thin-bedded model")
    d26 = np.arange(0, 25.1, 1.0)                       # 26 depths x 6 tools = 156 points
    dtb = np.arange(0, 20.01, 1.0)                      # 21 depths x 4 tools = 84 points (every fourth depth of the reference's log)
    return dict(
        example_01=dict(tools=EX01_TOOLS, depths=d26, rows=np.rint(d26 / 0.1).astype(int), gold=os.path.join(ex1, "Output/Results_2024_08_17__18_59_29/Results_1.txt"),
                        formation=os.path.join(ex1, "Input/Formation.txt"), borehole=os.path.join(ex1, "Input/Borehole.txt"), kw={}),
        # (the committed Example_02 log was made with R = 50, not the 25 of Example_02.py: DESIGN.md section 4)
        example_02=dict(tools=EX01_TOOLS, depths=d26, rows=np.rint(d26 / 0.1).astype(int), gold=os.path.join(ex2, "Output/Results_2024_08_17__19_03_42/Results_1.txt"),
                        formation=os.path.join(ex2, "Input/Formation.txt"), borehole=os.path.join(ex2, "Input/Borehole.txt"),
                        set_kw=dict(borehole_geometry_type="diameter"), kw=dict(mesh_generator="netgen", domain_radius=50, batch_size=10)),
        # (the reference's thin-bedded logs were made with R = 15, batch 10: DESIGN.md section 4)
        thin_bedded_1=dict(tools=TB_TOOLS, depths=dtb, rows=np.rint(dtb / 0.25).astype(int), gold=os.path.join(tb, "Logs", "Logs 1", "Results_1.txt"),
                           formation=os.path.join(tb, "Formation", "Formation_model_1.txt"), borehole=os.path.join(tb, "Borehole", "Borehole_model_correct_rm.txt"),
                           kw=dict(domain_radius=15, batch_size=10)))


@pytest.mark.parametrize("case,bounds", [("example_01", (1.5e-3, 1e-2, 3e-2)), ("example_02", (1.5e-3, 1e-2, 3e-2)), ("thin_bedded_1", (2e-3, 1.5e-2, 3e-2))])
def test_oracle_reproduces_the_reference_logs_on_the_coarse_size_field(case, bounds, examples_dir):
    """156 + 156 + 84 points, mesh_scale 1.0 (2.9 x the default element size): median / p99 / max against the reference's logs."""
    c = _cases(examples_dir)[case]
    rel = _oracle_sweep(c["tools"], c["depths"], c["rows"], c["gold"], c["formation"], c["borehole"], set_kw=c.get("set_kw"), mesh_scale=1.0, **c["kw"])
    p = _pcts(rel)
    print("oracle vs the reference's %s log, coarse size field: %s" % (case, p))
    assert p["points"] >= 84
    assert p["median"] < bounds[0] and p["p99"] < bounds[1] and p["max"] < bounds[2], p


@pytest.mark.skipif(os.environ.get("REMO_ORACLE_FULL") != "1", reason="~15 minutes of 8 cores: REMO_ORACLE_FULL=1 (rewrites the committed record)")
def test_oracle_at_the_default_mesh_scale_full(examples_dir):
    rec = dict(note="oracle/fem_oracle.c through Model.simulate_logs (tests/oracle_backend.py) at the DEFAULT mesh scale against the reference's committed logs; "
                    "relative differences of apparent resistivity", threads=THREADS, cases={})
    for name, c in _cases(examples_dir).items():
        rel = _oracle_sweep(c["tools"], c["depths"], c["rows"], c["gold"], c["formation"], c["borehole"], set_kw=c.get("set_kw"), **c["kw"])
        rec["cases"][name] = dict(_pcts(rel), per_tool={t: _pcts(rel[i]) for i, t in enumerate(c["tools"])}, settings={k: v for k, v in c["kw"].items()})
        print(name, rec["cases"][name])
    with open(RECORD, "w") as f:
        json.dump(rec, f, indent=1)


def test_committed_default_scale_record_meets_the_logs():
    """The record of the full-resolution oracle sweeps (test above, run in the build container) holds >= 150 points per Example
    log and a thin-bedded sample, and meets the reference's logs at p99 < 2e-3 - the tolerance the GPU path meets them with."""
    if not os.path.exists(RECORD):
        pytest.skip("no committed record yet (REMO_ORACLE_FULL=1 writes it)")
    rec = json.load(open(RECORD))
    for name in ("example_01", "example_02", "thin_bedded_1"):
        c = rec["cases"][name]
        assert c["points"] >= (150 if name.startswith("example") else 80)
        assert c["median"] < 3e-4 and c["p99"] < 2e-3 and c["max"] < 1e-2, (name, c)


def test_config1_bm1_ten_depths_through_the_oracle(examples_dir):
    """BASELINE configs[0] at its own workload: Benchmark model 1 (2D axisymmetric), one normal tool A0.4M6.0N, the
    ten depths linspace(10, 50, 10), R = 50, batch 5 => 2 batches / 10 right-hand sides, through the CPU restatement (the plumbing
    leg: no GPU).  Checked here: it runs, converges, and the readings lie between the model's smallest and largest resistivity;
    tests/test_gpu_configs.py::test_config1_bm1_hip_against_the_oracle compares the HIP path with the same two batches at 1e-10."""
    from remo3d_amd.model import Model
    ex = os.path.join(examples_dir, "Benchmark models", "Benchmark model 1")
    m = Model(["A0.4M6.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
    m.initialize_workers(cpu_workers=1, gpu_workers=2, context_factory=OracleContext)
    m.simulate_logs(np.linspace(10, 50, 10), verbose=False, domain_radius=50, batch_size=5, mesh_scale=1.0)
    m.shutdown_workers()
    assert m.timing["batches"] == 2 and m.timing["failed_batches"] == 0 and m.timing["not_converged"] == 0 and m.timing["points"] == 10
    ra = m.logs["A0.4M6.0N"][:, 1]
    rho = np.concatenate([m.formation_model[:, 2:].ravel(), m.borehole_model[:, 2]])
    rho = rho[np.isfinite(rho)]
    assert np.all(np.isfinite(ra)) and np.all(ra > 0.5 * rho.min()) and np.all(ra < 2.0 * rho.max()), (ra, rho.min(), rho.max())
