"""BASELINE.json configs and the reference's committed logs, at their own workloads, on the GPU (through the C ABI).

  * NGSolve probe: is `import ngsolve` possible on the GPU box?  If it is, the build's own harness (tests/ngsolve_harness.py)
    solves the test meshes with NGSolve and the HIP path must agree to 1e-6 on apparent resistivity (north star).
  * the reference's complete committed logs: Example_01, Example_02, thin-bedded Logs 1-4 (every point, asserted percentiles);
  * config 2: Benchmark model 1 (2D), one normal tool, 100 depths: HIP vs the CPU oracle on every right-hand side;
  * config 3: two batches of the headline sweep at size L (normal + lateral tool, ten right-hand sides) on the library's default path vs the oracle;
  * config 5: one batch of the ~5 M-dof mesh on the default (patch) operator and on the assembled matrix, mixed precision vs fp64, properties, TRUE residual;
  * dipping 3D: closed-form image solution across a plane interface inclined by 30 degrees;
  * MSH 2.2 files into the HIP path;
  * many short solves on two contexts at once (the "all columns frozen" flag must not split a workgroup).
"""
import json
import os
import time

import numpy as np
import pytest

from conftest import ROOT, SIGMA3

pytestmark = pytest.mark.gpu

EX01_TOOLS = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
TB_TOOLS = ["A0.4M6.0N", "A1.62M6.0N", "A4.0M0.5N", "A8.0M1.0N"]


def _record(name, obj):
    """Evidence written beside the test run (gpurun_out/ is merged back from the GPU box)."""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        json.dump(obj, f, indent=1)


def _pcts(rel):
    rel = np.asarray(rel).ravel()
    return dict(points=int(rel.size), nan=int(np.isnan(rel).sum()), median=float(np.nanmedian(rel)), p90=float(np.nanpercentile(rel, 90)),
                p99=float(np.nanpercentile(rel, 99)), max=float(np.nanmax(rel)))


# ---------------------------------------------------------------------------------------------------------------
def test_ngsolve_probe_and_parity_if_present(mesh2d, mesh3d, gpu_ctx):
    """SURVEY 8c / 8d Baseline A.  The outcome of the probe is recorded either way (gpurun_out/ngsolve_probe.json)."""
    import ngsolve_harness as H
    from remo3d_amd import solver
    ok, info = H.available()
    rec = dict(ngsolve_importable=bool(ok), detail=info)
    if not ok:
        _record("ngsolve_probe.json", rec)
        return        # "ngsolve absent": the oracle stays unpinned against NGSolve, said so in DESIGN.md and in the bench line
    src, ev = [([0.0], [1.0])], [[0.4, 6.4, 2.0, 2.5]]
    rec["cases"] = []
    try:
        for name, mesh in (("2d", mesh2d), ("3d", mesh3d)):
            for pre in ("local", "multigrid"):
                t0 = time.time()
                u_ng, ndof = H.solve(mesh, SIGMA3, src[0][0], src[0][1], ev[0], preconditioner=pre, condense=True, rtol=1e-12)
                t_ng = time.time() - t0
                outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, src, ev, solver.make_opts(preconditioner=pre, rtol=1e-12, maxsteps=20000))
                assert rc == 0
                u = outs[0]
                ra = lambda w: np.array([abs(4 * np.pi * 0.4 * 6.4 / 6.0 * (w[1] - w[0])), abs(4 * np.pi * 2.0 * 2.5 / 0.5 * (w[3] - w[2]))])
                d_ra = float(np.max(np.abs(ra(u) - ra(u_ng)) / np.abs(ra(u_ng))))
                rec["cases"].append(dict(mesh=name, preconditioner=pre, ndof_ngsolve=int(ndof), ndof_hip=int(st["n_dof"]), seconds_ngsolve=t_ng,
                                         max_rel_diff_potential=float(np.max(np.abs(u - u_ng) / np.abs(u_ng))), max_rel_diff_ra=d_ra))
    except AssertionError:
        raise
    except Exception as ex:       # the harness could not be exercised in the build container: an API mismatch is not a parity failure
        rec["harness_error"] = "%s: %s" % (type(ex).__name__, ex)
        _record("ngsolve_probe.json", rec)
        pytest.skip("ngsolve is importable but the harness failed: " + rec["harness_error"])
    _record("ngsolve_probe.json", rec)
    assert all(c["max_rel_diff_ra"] <= 1e-6 for c in rec["cases"]), rec


# ---------------------------------------------------------------------------------------------------------------
def _compare_with_log(tools, depths, gold_path, formation, borehole, **kw):
    from remo3d_amd.model import Model
    gold = np.loadtxt(gold_path, skiprows=2)
    t0 = time.time()
    m = Model.compute_synthetic_logs(tools, depths, formation, borehole, gpu_workers=1, verbose=False, **kw)
    rel = np.array([np.abs(m.logs[t][:, 1] - gold[:, 1 + i]) / gold[:, 1 + i] for i, t in enumerate(tools)])
    signed = np.array([(m.logs[t][:, 1] - gold[:, 1 + i]) / gold[:, 1 + i] for i, t in enumerate(tools)])
    out = _pcts(rel)
    out.update(seconds=time.time() - t0, mesh_s=m.timing["mesh_s"], solve_s=m.timing["solve_s"], failed_batches=m.timing["failed_batches"],
               per_tool={t: _pcts(rel[i]) for i, t in enumerate(tools)}, per_tool_signed_mean={t: float(np.nanmean(signed[i])) for i, t in enumerate(tools)})
    return out, rel, signed


def test_example_01_complete_log(examples_dir):
    """The reference's complete Example_01 (251 depths x 6 tools = 1506 points, default settings) against its committed log.
    The two meshes differ (in-repo conforming Delaunay mesh vs Netgen), so this is a mesh tolerance; the reference's own
    Example_01 vs Example_02 runs differ by up to 3.1e-4."""
    ex = os.path.join(examples_dir, "Example_01")
    out, rel, _ = _compare_with_log(EX01_TOOLS, np.arange(0, 25.1, 0.1), os.path.join(ex, "Output/Results_2024_08_17__18_59_29/Results_1.txt"),
                                    os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"))
    _record("example01_parity.json", out)
    print("Example_01:", {k: out[k] for k in ("median", "p90", "p99", "max", "seconds")})
    assert out["points"] == 1506 and out["nan"] == 0 and out["failed_batches"] == 0
    assert out["median"] < 3e-4 and out["p99"] < 1e-3 and out["max"] < 5e-3, out


def test_example_02_complete_log(examples_dir):
    """Example_02 of the reference: the same model, batch_size 10, Netgen-path windowing, borehole given by diameter.
    Example_02.py also passes domain_radius = 25, but the log the reference COMMITTED was not computed with it: it agrees with
    its own Example_01 log (R = 50) to median 2e-5 / max 3.1e-4, and with our runs at R = 50 to median 6e-5 / p99 8e-4, against
    4.3e-4 / 3.4e-3 at R = 25 and worse at 20 or 30 (tools/example02_settings_scan.py, profiles/r02_example02_settings_scan.log).
    Compared here at the radius the numbers were evidently made with."""
    ex = os.path.join(examples_dir, "Example_02")
    out, rel, _ = _compare_with_log(EX01_TOOLS, np.arange(0, 25.1, 0.1), os.path.join(ex, "Output/Results_2024_08_17__19_03_42/Results_1.txt"),
                                    os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"),
                                    borehole_geometry_type="diameter", dip=0, mesh_generator="netgen", domain_radius=50, batch_size=10)
    out["settings"] = dict(domain_radius=50, batch_size=10, note="Example_02.py says domain_radius=25; the committed log does not")
    _record("example02_parity.json", out)
    print("Example_02:", {k: out[k] for k in ("median", "p90", "p99", "max", "seconds")})
    assert out["points"] == 1506 and out["nan"] == 0 and out["failed_batches"] == 0
    assert out["median"] < 3e-4 and out["p99"] < 1e-3 and out["max"] < 5e-3, out


def test_example_02_at_the_settings_of_its_script(examples_dir):
    """The fit made visible: Example_02 with the domain_radius = 25 its script passes (Example_02.py:21) instead of the 50 its
    committed log was evidently made with (test above).  The looser bounds are the measured ones: median 4.3e-4, p99 3.4e-3."""
    ex = os.path.join(examples_dir, "Example_02")
    out, rel, _ = _compare_with_log(EX01_TOOLS, np.arange(0, 25.1, 0.1), os.path.join(ex, "Output/Results_2024_08_17__19_03_42/Results_1.txt"),
                                    os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"),
                                    borehole_geometry_type="diameter", dip=0, mesh_generator="netgen", domain_radius=25, batch_size=10)
    out["settings"] = dict(domain_radius=25, batch_size=10, note="the settings of Example_02.py; the committed log agrees with R = 50 instead")
    _record("example02_parity_script_settings.json", out)
    print("Example_02 at R = 25:", {k: out[k] for k in ("median", "p90", "p99", "max")})
    assert out["points"] == 1506 and out["nan"] == 0 and out["failed_batches"] == 0
    assert out["median"] < 1e-3 and out["p99"] < 6e-3 and out["max"] < 2e-2, out


def test_thin_bedded_log_at_default_settings(examples_dir):
    """The fit made visible: thin-bedded Logs 1 with the DEFAULT settings (R = 50, batch 5) instead of the R = 15 / batch 10 the
    reference's logs were evidently made with.  The offset grows with the tool length (medians 3.8e-4, 1.8e-3, 2.9e-3 for the 6.4 m
    normal, the 7.6 m normal and the 4.5 m lateral); the 9 m lateral A8.0M1.0N carries the sawtooth (median 2.5e-2, up to +4.7 %)
    that led to the settings scan."""
    base = os.path.join(examples_dir, "Benchmark models", "Thin-bedded model")
    out, rel, signed = _compare_with_log(TB_TOOLS, np.arange(0, 20.01, 0.25), os.path.join(base, "Logs", "Logs 1", "Results_1.txt"),
                                         os.path.join(base, "Formation", "Formation_model_1.txt"), os.path.join(base, "Borehole", "Borehole_model_correct_rm.txt"))
    out["settings"] = dict(domain_radius=50, batch_size=5, note="Model defaults; the reference's logs agree with R = 15, batch 10 instead")
    _record("thin_bedded_Logs_1_default_settings.json", out)
    print("thin-bedded Logs 1, defaults:", {t: (out["per_tool"][t]["median"], out["per_tool"][t]["max"]) for t in TB_TOOLS})
    assert out["nan"] == 0 and out["failed_batches"] == 0
    for t in TB_TOOLS[:3]:
        assert out["per_tool"][t]["median"] < 5e-3 and out["per_tool"][t]["max"] < 2e-2, (t, out["per_tool"][t])
    assert out["per_tool"]["A8.0M1.0N"]["median"] < 5e-2 and out["per_tool"]["A8.0M1.0N"]["max"] < 8e-2, out["per_tool"]["A8.0M1.0N"]


@pytest.mark.parametrize("logs,formation,shifted", [("Logs 1", "Formation_model_1.txt", False), ("Logs 2", "Formation_model_2.txt", False),
                                                    ("Logs 3", "Formation_model_1.txt", True), ("Logs 4", "Formation_model_2.txt", True)])
def test_thin_bedded_logs(logs, formation, shifted, examples_dir):
    """The reference's thin-bedded benchmark (140 / 201 layers of ~0.125 m, 81 depths x 4 tools incl. the 9 m lateral
    A8.0M1.0N); Logs 3 / 4 were computed at the misaligned depths of Logs_depth_shifts.txt.  The reference did not record the
    settings of these runs.  They are domain_radius = 15, batch_size = 10: with the defaults (50 / 5) the long lateral differs
    from the reference's log by a sawtooth in depth that is linear in the position of the current electrode inside a batch of
    TEN depths (the image of an off-centre source in a grounded sphere only 15 m away), and a scan over the two settings
    (tools/thin_bedded_settings_scan.py, profiles/r02_thin_bedded_settings_scan.log) takes the median difference of that tool
    from 2.5e-2 (R = 50) through 1.3e-2 (20) and 7.7e-3 (17.5) to 2.8e-4 at R = 15 - where ALL four tools agree with the
    reference's `%.4f` log to a few 1e-5."""
    base = os.path.join(examples_dir, "Benchmark models", "Thin-bedded model")
    depths = np.arange(0, 20.01, 0.25)
    if shifted:
        depths = np.loadtxt(os.path.join(base, "Logs", "Logs_depth_shifts.txt"), skiprows=2)[:, 1]
    out, rel, signed = _compare_with_log(TB_TOOLS, depths, os.path.join(base, "Logs", logs, "Results_1.txt"), os.path.join(base, "Formation", formation),
                                         os.path.join(base, "Borehole", "Borehole_model_correct_rm.txt"), domain_radius=15, batch_size=10)
    out["settings"] = dict(domain_radius=15, batch_size=10)
    out["A8.0M1.0N_signed_by_depth"] = [[float(d), float(s)] for d, s in zip(depths, signed[3])]
    _record("thin_bedded_%s.json" % logs.replace(" ", "_"), out)
    print(logs, {t: (out["per_tool"][t]["median"], out["per_tool"][t]["p99"], out["per_tool"][t]["max"]) for t in TB_TOOLS})
    assert out["nan"] == 0 and out["failed_batches"] == 0
    assert out["median"] < 2e-4 and out["p90"] < 1e-3 and out["max"] < 1e-2, out
    for t in TB_TOOLS:
        assert out["per_tool"][t]["median"] < 5e-4, (t, out["per_tool"][t])


# ---------------------------------------------------------------------------------------------------------------
def test_config2_bm1_100_depths_against_the_oracle(examples_dir, gpu_ctx):
    """BASELINE configs[1]: Benchmark model 1 (2D axisymmetric), tool A0.4M6.0N, 100 depths in [5, 55] m, R = 50,
    batch 5 => 20 batches / 100 right-hand sides.  Every right-hand side through the HIP path and through the CPU oracle on
    the same mesh; both converged to rtol 1e-13: potentials within 1e-10 relative, apparent resistivity likewise."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.fem_oracle import solve_batch as oracle_batch
    from remo3d_amd import geometry, solver, tasks
    from remo3d_amd.model import Model, default_mesh_provider
    ex = os.path.join(examples_dir, "Benchmark models", "Benchmark model 1")
    m = Model(["A0.4M6.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
    depths = np.linspace(5, 55, 100)
    sim, batches = tasks.build_batches(m.tools, m.sec, depths, 5)
    assert len(batches) == 20 and sum(len(b.solves) for b in batches) == 100
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    provider = default_mesh_provider()
    work = []
    for bi, b in enumerate(batches):
        fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50.0)
        mesh = provider(2, 50.0, b, fg, bh, 0.0)
        sources, evals, readers = tasks.batch_rhs(b, m.tools)
        work.append((mesh, sigma, sources, evals, readers))
    opts = solver.make_opts(rtol=1e-13, maxsteps=20000)
    t0 = time.time()
    gpu = [gpu_ctx.solve_batch(w[0], w[1], w[2], w[3], opts) for w in work]
    t_gpu = time.time() - t0
    assert all(rc == 0 for _, _, rc in gpu)

    def cpu(w):
        mesh, sigma, sources, evals, _ = w
        sp, sz, sI, ep, ez = [0], [], [], [0], []
        for (z, I), e in zip(sources, evals):
            sz += list(z); sI += list(I); sp.append(len(sz)); ez += list(e); ep.append(len(ez))
        out, rc, st = oracle_batch(mesh, sigma, sp, sz, sI, ep, ez, condense=True, rtol=1e-13, maxit=100000)
        assert rc == 0
        return [out[ep[k]:ep[k + 1]] for k in range(len(evals))]
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as tp:      # the C calls release the GIL
        ref = list(tp.map(cpu, work))
    t_cpu = time.time() - t0
    worst_u = worst_ra = 0.0
    for (outs, st, rc), r, w in zip(gpu, ref, work):
        for u, ur, rd in zip(outs, r, w[4]):
            worst_u = max(worst_u, float(np.max(np.abs(u - ur) / np.abs(ur))))
            for (di, ti, K, o, mm) in rd:
                a, b = tasks.apparent_resistivity(u[o:o + mm], mm, K, 2), tasks.apparent_resistivity(ur[o:o + mm], mm, K, 2)
                worst_ra = max(worst_ra, abs(a - b) / abs(b))
    _record("config2_bm1_parity.json", dict(batches=len(work), rhs=100, n_free=int(gpu[0][1]["n_free"]), max_rel_diff_potential=worst_u, max_rel_diff_ra=worst_ra,
                                           gpu_seconds=t_gpu, oracle_seconds=t_cpu, rtol=1e-13))
    print("config 2: potentials %.2e, Ra %.2e (GPU %.1f s, oracle %.1f s)" % (worst_u, worst_ra, t_gpu, t_cpu))
    assert worst_u <= 1e-10 and worst_ra <= 1e-9, (worst_u, worst_ra)


def test_config1_bm1_hip_against_the_oracle(examples_dir, gpu_ctx):
    """BASELINE configs[0] at its own workload: Benchmark model 1, tool A0.4M6.0N, depths linspace(10, 50, 10),
    R = 50, batch 5 => 2 batches / 10 right-hand sides.  The CPU suite runs this sweep through the oracle
    (tests/test_oracle_reference_logs.py::test_config1_bm1_ten_depths_through_the_oracle); here the HIP path and the oracle solve
    the same two batches on the same meshes (default mesh scale), both converged to 1e-13: potentials within 1e-10."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.fem_oracle import solve_batch as oracle_batch
    from remo3d_amd import geometry, solver, tasks
    from remo3d_amd.model import Model, default_mesh_provider
    ex = os.path.join(examples_dir, "Benchmark models", "Benchmark model 1")
    m = Model(["A0.4M6.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
    sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(10, 50, 10), 5)
    assert len(batches) == 2 and sum(len(b.solves) for b in batches) == 10
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    provider = default_mesh_provider()
    worst_u = worst_ra = 0.0
    for bi, b in enumerate(batches):
        fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50.0)
        mesh = provider(2, 50.0, b, fg, bh, 0.0)
        sources, evals, readers = tasks.batch_rhs(b, m.tools)
        outs, st, rc = gpu_ctx.solve_batch(mesh, sigma, sources, evals, solver.make_opts(rtol=1e-13, maxsteps=20000))
        assert rc == 0

        def cpu(k):
            (z, I), e = sources[k], evals[k]
            out, rc2, _ = oracle_batch(mesh, sigma, [0, len(z)], list(z), list(I), [0, len(e)], list(e), condense=True, rtol=1e-13, maxit=100000)
            assert rc2 == 0
            return out
        with ThreadPoolExecutor(max_workers=len(sources)) as tp:      # the C calls release the GIL
            ref = list(tp.map(cpu, range(len(sources))))
        for u, ur, rd in zip(outs, ref, readers):
            worst_u = max(worst_u, float(np.max(np.abs(u - ur) / np.abs(ur))))
            for (di, ti, K, o, mm) in rd:
                a, c = tasks.apparent_resistivity(u[o:o + mm], mm, K, 2), tasks.apparent_resistivity(ur[o:o + mm], mm, K, 2)
                worst_ra = max(worst_ra, abs(a - c) / abs(c))
    _record("config1_bm1_parity.json", dict(batches=2, rhs=10, max_rel_diff_potential=worst_u, max_rel_diff_ra=worst_ra, rtol=1e-13))
    print("config 1: potentials %.2e, Ra %.2e" % (worst_u, worst_ra))
    assert worst_u <= 1e-10 and worst_ra <= 1e-9, (worst_u, worst_ra)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,vertex_solver", [("fp64", "library default"), ("mixed", "library default"), ("fp64", "cycle"), ("mixed", "cycle")])
def test_config3_size_L_batches_against_the_oracle(precision, vertex_solver, gpu_ctx, size_L_case):
    """BASELINE configs[2] at its own workload and size (bench.py's headline): Benchmark model 3, dip 30, tools A0.4M6.0N
    (normal) + A2.0M0.5N (lateral), 100 depths, batch 5, meshes at size L - two of its 40 batches (batch 0: lateral tool; batch 20:
    both tools; ten right-hand sides, 1.5-1.7 M unknowns each) through the calls `Model.simulate_logs` makes per batch
    (Context.solve_batch: patch operator, P1 block + diagonal assembled; vertex-block solver = the library's default, the Chebyshev
    polynomial, or "cycle" = `coarse="amg_or_chebyshev"`, what Model and the bench's headline ask for when several contexts share the
    GPU: one multigrid cycle; tasks.apparent_resistivity) and through the CPU oracle, right-hand side by right-hand side (threads
    started by the fixture): potentials within 2e-8 (fp64) / 1e-6 (mixed), apparent resistivity within 1e-6 (the north star's tolerance)."""
    from remo3d_amd import solver, tasks
    worst_u = worst_ra = 0.0
    n_rhs = 0
    for bi, w in enumerate(size_L_case["work"]):
        cycle = vertex_solver == "cycle"
        outs, st, rc = gpu_ctx.solve_batch(w["mesh"], w["sigma"], w["sources"], w["evals"],
                                           solver.make_opts(rtol=1e-10, precision=precision, coarse="amg_or_chebyshev" if cycle else "auto"))
        assert rc == 0, st
        assert st["op_used"] == 3 and st["nnz"] == 0 and st["coarse_used"] == (2 if cycle else 1) and st["n_free"] > 1400000, st
        for k, (u, rd) in enumerate(zip(outs, w["readers"])):
            ref, rc_o, st_o = size_L_case["futs"][(bi, k)].result(timeout=1200)
            assert rc_o == 0 and st_o["n"] == st["n_free"]
            worst_u = max(worst_u, float(np.max(np.abs(u - ref) / np.abs(ref))))
            for (di, ti, K, o, mm) in rd:
                a, c = tasks.apparent_resistivity(u[o:o + mm], mm, K, 3), tasks.apparent_resistivity(np.asarray(ref)[o:o + mm], mm, K, 3)
                worst_ra = max(worst_ra, abs(a - c) / abs(c))
            n_rhs += 1
    assert n_rhs == 10
    tools_seen = sorted({ti for w in size_L_case["work"] for rd in w["readers"] for (di, ti, K, o, mm) in rd})
    assert tools_seen == [0, 1]                          # normal and lateral tool
    _record("config3_sizeL_parity_%s_%s.json" % (precision, vertex_solver.replace(" ", "_")), dict(batches=2, vertex_solver=vertex_solver, rhs=n_rhs, mesh_T=[int(w["mesh"].n_elems) for w in size_L_case["work"]],
                                                             max_rel_diff_potential=worst_u, max_rel_diff_ra=worst_ra, rtol=1e-10, precision=precision))
    print("config 3 at size L (%s, %s): potentials %.2e, Ra %.2e" % (precision, vertex_solver, worst_u, worst_ra))
    assert worst_u <= (2e-8 if precision == "fp64" else 1e-6) and worst_ra <= 1e-6, (worst_u, worst_ra)


# ---------------------------------------------------------------------------------------------------------------
def test_config5_xl_batch_mixed_precision(gpu_ctx):
    """BASELINE configs[4]: one batch of the ~5 M-dof mesh (bench size XL), fp32 PCG inside fp64 residual refinement against the
    fp64 solve, each on the operator the library picks at this size (op = "auto": the patch operator, nothing of A assembled but
    the diagonal and the P1 block) and on the assembled matrix (op = "csr", 258 M stored entries): potentials within 1e-6,
    reciprocity, linearity, and the TRUE residual f - A x of every solution (recomputed from the solution with one device product)
    at the requested tolerance."""
    import bench
    from remo3d_amd import solver
    w = bench.build_workload(0, 1, 5, bench.SIZES["XL"], max_batches=1)["work"][0]
    mesh, sigma = w["mesh"], np.asarray(w["sigma"], dtype=np.float64)
    za, zb, zc = 0.0, 0.4, 6.4
    src = [([za], [1.0]), ([zb], [1.0]), ([za, zb], [1.0, -1.0]), ([za], [2.5]), ([0.15], [1.0])]
    ev = [[zb, zc], [za, zc], [zc, 2.0], [zb, zc], [2.1, 2.6]]
    b = gpu_ctx.batch(mesh, sigma, src, ev)
    res = {}
    try:
        for variant in ("fp64", "mixed", "fp64+csr", "mixed+csr"):
            t0 = time.time()
            csr = variant.endswith("+csr")
            rc = b.run(solver.make_opts(rtol=1e-9, precision=variant.split("+")[0], maxsteps=3000, op="csr" if csr else "auto"))
            st = b.stats
            assert rc == 0, st
            assert st["op_used"] == (0 if csr else 3) and (st["nnz"] > 200000000 if csr else st["nnz"] == 0), st
            assert st["n_free"] > 4500000, st["n_free"]
            res[variant] = dict(out=[o.copy() for o in b.fetch()], steps=int(st["pcg_steps"]), ms=float(st["ms_solve"]), true_relres=b.true_relres().tolist(),
                                wall=time.time() - t0, cycles=int(st["refinement_cycles"]), nnz=int(st["nnz"]))
    finally:
        b.close()
    tol = 1e-6
    for variant, r in res.items():
        out = r["out"]
        assert abs(out[0][0] - out[1][0]) <= tol * abs(out[0][0]), variant                      # reciprocity u_a(z_b) = u_b(z_a)
        assert abs(out[2][0] - (out[0][1] - out[1][1])) <= tol * abs(out[0][1]), variant        # dipole = difference of its poles
        assert np.allclose(out[3], 2.5 * out[0], rtol=tol, atol=0), variant
        assert max(r["true_relres"]) <= 5e-9, (variant, r["true_relres"])                       # asked for 1e-9 (recurrence / refinement)
    for other in ("mixed", "fp64+csr", "mixed+csr"):
        for a, c in zip(res["fp64"]["out"], res[other]["out"]):
            assert np.allclose(a, c, rtol=tol, atol=0), other
        assert res[other]["cycles"] >= (1 if "mixed" in other else 0)
    _record("config5_xl_batch.json", dict(n_free=int(st["n_free"]), T=int(mesh.n_elems),
                                         **{p: {k: v for k, v in r.items() if k != "out"} for p, r in res.items()},
                                         max_rel_diff_mixed_vs_fp64=float(max(np.max(np.abs(a - c) / np.abs(a)) for a, c in zip(res["fp64"]["out"], res["mixed"]["out"])))))


# ---------------------------------------------------------------------------------------------------------------
def _dipping_interface_case(scale, sectors=6):
    """Point source on the axis in medium 1, plane interface z + x tan(30 deg) = b to medium 2 (what worker.py:128-131 /
    gmsh_functions.py:610-612 build for a dipping bed; no borehole).  Image solution with the source mirrored in the plane."""
    from remo3d_amd import meshgen
    R, dip, bnd, s1, s2 = 50.0, np.deg2rad(30.0), 1.5, 0.2, 0.02
    fg = np.array([[-80.0, bnd, np.nan], [bnd, 80.0, np.nan]])
    bh = np.array([[-80.0, 0.0], [80.0, 0.0]])                         # radius 0: no mud column
    zs = np.array([0.4, 1.0, -0.7, -3.0, 2.0, 4.0, 6.4])
    mesh = meshgen.make_mesh_3d_conforming(R, fg, bh, dip, sources_z=[0.0], snap_z=list(zs), scale=scale, sectors=sectors)
    assert mesh.mat.min() >= 1                                          # nothing classified as mud
    n = np.array([np.sin(dip), 0.0, np.cos(dip)])                       # unit normal of the plane n.x = h
    h = bnd * np.cos(dip)
    image = 2.0 * h * n
    k = (s1 - s2) / (s1 + s2)
    P = np.stack([np.zeros_like(zs), np.zeros_like(zs), zs], 1)
    in1 = P @ n < h
    u_full = np.where(in1, (1.0 / np.abs(zs) + k / np.linalg.norm(P - image, axis=1)) / (4 * np.pi * s1), 1.0 / (2 * np.pi * (s1 + s2) * np.abs(zs)))
    return mesh, [1.0, s1, s2], zs, 2.0 * u_full                        # half-space model: twice the potential (worker.py:129)


@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_dipping_interface_image_solution_3d(precision, gpu_ctx):
    """External pin of the dipping (dip != 0) 3D physics: differences of axis potentials on both sides of the inclined interface
    against the closed form to 5e-3 (the grounded sphere shifts all potentials alike), falling under refinement."""
    from remo3d_amd import solver
    errs = {}
    for scale in (2.0, 1.0):
        mesh, sigma, zs, exact = _dipping_interface_case(scale)
        # (the degenerate "borehole" of radius 0 leaves needle elements along the axis: ~1900 PCG steps in fp64; the fp32 inner solver
        # of the mixed mode loses conjugacy on such a system and is asked for the reference's default tolerance only)
        rtol = 1e-10 if precision == "fp64" else 1e-8
        outs, st, rc = gpu_ctx.solve_batch(mesh, sigma, [([0.0], [1.0])], [list(zs)], solver.make_opts(rtol=rtol, precision=precision, maxsteps=20000))
        assert rc == 0, (rc, st["pcg_steps"], st["relres"][:1], st["refinement_cycles"])
        got = outs[0]
        d_got, d_ex = got[:-1] - got[1:], exact[:-1] - exact[1:]
        errs[scale] = (float(np.max(np.abs(d_got - d_ex) / np.abs(d_ex))), int(mesh.n_elems), int(st["n_free"]), int(st["pcg_steps"]))
    _record("dipping_interface_%s.json" % precision, {str(k): v for k, v in errs.items()})
    print("dipping interface, max rel error of potential differences by mesh scale:", errs)
    assert errs[1.0][0] < 5e-3, errs
    assert errs[1.0][0] < errs[2.0][0], errs


# ---------------------------------------------------------------------------------------------------------------
GMSH_STYLE_3D = """$MeshFormat
2.2 0 8
$EndMeshFormat
$PhysicalNames
5
2 1 "dirichlet_boundary"
2 2 "neumann_boundary"
3 3 "vol_1"
3 4 "vol_2"
3 5 "vol_3"
$EndPhysicalNames
$Nodes
%(nn)d
%(nodes)s
$EndNodes
$Elements
%(ne)d
%(elems)s
$EndElements
"""


def _write_gmsh_style(path, mesh, tags):
    """A file laid out the way Gmsh writes the reference's 3D models (gmsh_functions.py:660-678): sparse node ids, a point
    record first, boundary triangles and tetrahedra mixed entity by entity, elementary tags out of order."""
    ids = 7 + 3 * np.arange(mesh.n_nodes)                      # not 1..n
    nodes = "\n".join("%d %.17g %.17g %.17g" % (ids[k], p[0], p[1], p[2]) for k, p in enumerate(mesh.coords))
    recs = ["1 15 2 0 1 %d" % ids[0]]
    eid = 2
    half = len(mesh.bconn) // 2
    def tri(fac, d):
        nonlocal eid
        phys = 1 if d else 2
        recs.append("%d 2 2 %d %d %s" % (eid, phys, 40 + phys, " ".join(str(ids[v]) for v in fac))); eid += 1
    def tet(t):
        nonlocal eid
        recs.append("%d 4 2 %d %d %s" % (eid, 3 + int(mesh.mat[t]), tags[int(mesh.mat[t])], " ".join(str(ids[v]) for v in mesh.conn[t]))); eid += 1
    for fac, d in zip(mesh.bconn[:half], mesh.bdirichlet[:half]):
        tri(fac, d)
    order = np.argsort(np.asarray(tags)[mesh.mat], kind="stable")          # entity by entity, in tag order (not material order)
    for t in order[: len(order) // 2]:
        tet(t)
    for fac, d in zip(mesh.bconn[half:], mesh.bdirichlet[half:]):
        tri(fac, d)
    for t in order[len(order) // 2:]:
        tet(t)
    with open(path, "w") as f:
        f.write(GMSH_STYLE_3D % dict(nn=mesh.n_nodes, nodes=nodes, ne=len(recs), elems="\n".join(recs)))


@pytest.mark.parametrize("which", ["2d", "3d", "gmsh3d"])
def test_msh_file_into_the_hip_path(which, mesh2d, mesh3d, gpu_ctx, tmp_path):
    """a mesh that went through an MSH 2.2 file (ReadGmsh's numbering: materials by first appearance of the
    elementary tag, Dirichlet by physical name, gmsh_functions.py:243-381) gives the SAME potentials as the arrays it was written
    from, bit for bit, once sigma follows the material permutation."""
    from remo3d_amd import msh_io, solver
    mesh = mesh2d if which == "2d" else mesh3d
    p = str(tmp_path / "fm_0.msh")
    if which == "gmsh3d":
        _write_gmsh_style(p, mesh, tags=[31, 12, 25])
    else:
        msh_io.write_msh(p, mesh)
    back = msh_io.read_msh(p, mesh.dim)
    old = {tuple(c): int(t) for c, t in zip(mesh.conn.tolist(), mesh.mat.tolist())}
    old_of_new = {}
    for c, t in zip(back.conn.tolist(), back.mat.tolist()):
        old_of_new[int(t)] = old[tuple(c)]
    sigma_back = [SIGMA3[old_of_new[k]] for k in range(len(old_of_new))]
    if which == "gmsh3d":
        assert [old_of_new[k] for k in range(3)] == [1, 2, 0]           # tags 12, 25, 31 appear in that order
        assert int(back.bdirichlet.sum()) == int(mesh.bdirichlet.sum())
    src, ev = [([0.0], [1.0]), ([0.1, -0.1], [1.0, -1.0])], [[0.4, 6.4], [2.0, 2.5, -1.0]]
    opts = solver.make_opts(rtol=1e-10)
    a, st_a, rc_a = gpu_ctx.solve_batch(mesh, SIGMA3, src, ev, opts)
    b, st_b, rc_b = gpu_ctx.solve_batch(back, sigma_back, src, ev, opts)
    assert rc_a == 0 and rc_b == 0 and st_a["n_free"] == st_b["n_free"] and st_a["nnz"] == st_b["nnz"]
    for x, y in zip(a, b):
        assert np.allclose(x, y, rtol=1e-9, atol=0), (x, y)

def test_many_short_solves_on_two_contexts(mesh2d, mesh3d):
    """Several contexts on one GPU start their workgroups late and unevenly: the update launch that freezes the last column
    must not let a workgroup split on the flag it raises itself (the survivors would sum LDS slots the leavers never wrote).
    Many short solves on two contexts at once; every returned solution is checked against its TRUE residual."""
    from concurrent.futures import ThreadPoolExecutor
    from remo3d_amd import solver
    src = [([0.0], [1.0]), ([0.1], [1.0]), ([-0.1, 0.1], [1.0, -1.0])]
    ev = [[0.4, 6.4], [2.1, 2.6], [0.5, 3.0]]

    def drive(j):
        worst = 0.0
        with solver.Context(0) as ctx:
            for mesh in (mesh2d, mesh3d):
                b = ctx.batch(mesh, SIGMA3, src, ev)
                try:
                    for rep in range(30):
                        rtol = (1e-4, 1e-6, 1e-8)[(rep + j) % 3]
                        rc = b.run(solver.make_opts(rtol=rtol, check_every=(1, 3, 5)[rep % 3]))
                        assert rc == 0
                        tr = b.true_relres()
                        worst = max(worst, float(np.max(tr)) / rtol)
                finally:
                    b.close()
        return worst
    with ThreadPoolExecutor(max_workers=2) as tp:
        worst = list(tp.map(drive, range(2)))
    print("true residual / requested tolerance, worst over 120 solves on two contexts:", worst)
    assert max(worst) < 3.0, worst      # the Jacobi-weighted true residual sits within a small factor of the <Cr,r> recurrence test
