"""Host bookkeeping (tool tables, batching, model loading, windowing) against golden vectors
generated from the reference's own functions (tests/golden/make_golden.py).  Bit-exact for
indices, 1e-12 for floats."""
import json
import os

import numpy as np
import pytest

from remo3d_amd import geometry, tasks, tools
from remo3d_amd.model import Model

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_tool_tables_match_reference():
    gold = load("tools.json")
    for key, case in gold.items():
        if key == "rejects":
            continue
        tables, sec = tools.tool_tables(case["names"], case["force"])
        assert sec == case["sec"], key
        assert list(tables.keys()) == case["names"]
        for name in case["names"]:
            np.testing.assert_allclose(tables[name], np.array(case["tables"][name]), rtol=1e-13, atol=1e-13, err_msg=f"{key} {name}")
    # a known value from SURVEY.md section 8c
    t, _ = tools.tool_tables(["A2.0M0.5N"])
    np.testing.assert_allclose(t["A2.0M0.5N"], [[0, 2, 2.5, 125.66370614359172], [1, 0, 0, -2.25]], rtol=1e-12)


def test_bad_tool_names_rejected_like_reference():
    for name, rejected in load("tools.json")["rejects"]:
        if rejected is True:
            with pytest.raises(ValueError):
                tools.tool_tables([name])
    with pytest.raises(ValueError):
        tools.tool_tables("A2.0M0.5N")          # not a list
    with pytest.raises(ValueError):
        tools.tool_tables(["A2.0M0.5N"], 1)     # flag not a bool


@pytest.mark.parametrize("fixture", ["tasks_example_01.json", "tasks_bm3.json", "tasks_bm1_single.json", "tasks_nonsec.json",
                                     "tasks_example_02.json", "tasks_thin_bedded.json"])
def test_batching_matches_reference(fixture):
    gold = load(fixture)
    tables, sec = tools.tool_tables(gold["names"], gold["force"])
    assert sec == gold["sec"]
    combined, batches = tasks.build_batches(tables, sec, np.array(gold["depths"]), gold["batch_size"])
    np.testing.assert_allclose(combined, gold["simulation_depths"], rtol=0, atol=1e-12)
    mine = tasks.to_reference_layout(batches)
    ref = gold["tasks"]
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        assert a[0] == b[0]
        np.testing.assert_allclose(np.array(a[1]), np.array(b[1]), rtol=0, atol=1e-12)
        assert len(a[2]) == len(b[2])
        for sa, sb in zip(a[2], b[2]):
            assert sa[0] == sb[0]
            np.testing.assert_allclose(np.array(sa[1]), np.array(sb[1]), rtol=0, atol=1e-12)
            assert len(sa[2]) == len(sb[2])
            for ra, rb in zip(sa[2], sb[2]):
                assert ra[0] == rb[0] and ra[1] == rb[1]
                assert abs(ra[2] - rb[2]) < 1e-12


def test_survey_counts():
    """Example_01 -> 164 batches / 818 solves / 1506 records; BM3 -> 40 / 200."""
    g = load("tasks_example_01.json")
    tables, sec = tools.tool_tables(g["names"], True)
    _, b = tasks.build_batches(tables, sec, np.array(g["depths"]), 5)
    assert len(b) == 164 and sum(len(x.solves) for x in b) == 818
    assert sum(len(s.records) for x in b for s in x.solves) == 1506
    g = load("tasks_bm3.json")
    tables, sec = tools.tool_tables(g["names"], True)
    _, b = tasks.build_batches(tables, sec, np.array(g["depths"]), 5)
    assert len(b) == 40 and sum(len(x.solves) for x in b) == 200


@pytest.mark.parametrize("fixture", ["windows_example_01.json", "windows_example_01_r5.json", "windows_bm2.json", "windows_bm2_r8.json",
                                     "windows_bm3_30.json", "windows_bm3_60_r6.json"])
def test_model_loading_and_windowing_match_reference(fixture, examples_dir):
    gold = load(fixture)
    m = Model(["A0.4M6.0N", "A2.0M0.5N"])
    m.set_model_parameters(os.path.join(examples_dir, gold["formation_file"]), os.path.join(examples_dir, gold["borehole_file"]),
                           dip=gold["dip_deg"])
    np.testing.assert_allclose(m.formation_model, np.array(gold["formation_model"], dtype=float), rtol=1e-14, equal_nan=True)
    np.testing.assert_allclose(m.borehole_model, np.array(gold["borehole_model_loaded"]), rtol=1e-14)
    if m.dip_deg != 0:
        m.borehole_model = m._add_points_to_borehole()
    np.testing.assert_allclose(m.borehole_model, np.array(gold["borehole_model"]), rtol=1e-13)
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    for case in gold["cases"]:
        fg, bh, sigma = geometry.select_data_range(bg, m.formation_model, m.dip_rad, case["rm"], case["depth"], gold["R"])
        np.testing.assert_allclose(fg, np.array(case["formation_geometry"], dtype=float), rtol=1e-13, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(bh, np.array(case["borehole_geometry"]), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(sigma, case["sigma"], rtol=1e-13)


def test_batch_rhs_and_ra_formula():
    tables, sec = tools.tool_tables(["A0.4M6.0N", "A2.0M0.5N"])
    _, batches = tasks.build_batches(tables, sec, np.linspace(5, 20, 100, endpoint=False), 5)
    src, ev, readers = tasks.batch_rhs(batches[3], tables)
    assert len(src) == len(batches[3].solves) == len(ev)
    for (z, I), s in zip(src, batches[3].solves):
        assert np.all(I == 1.0) and len(z) >= 1      # single-electrode mode: unit sources (remo3d.py:288-295, 659-660)
    # homogeneous full space: u = 1/(4 pi sigma r)  =>  Ra = 1/sigma exactly (remo3d.py:285-306)
    sigma = 0.25
    for name, t in tables.items():
        meas = t[0, :3][t[1, :3] == 0]
        cur = t[0, :3][t[1, :3] != 0][0]
        u = 1.0 / (4 * np.pi * sigma * np.abs(meas - cur))
        assert abs(tasks.apparent_resistivity(u, 2, t[0, 3], 2) - 1 / sigma) < 1e-12
        assert abs(tasks.apparent_resistivity(2 * u, 2, t[0, 3], 3) - 1 / sigma) < 1e-12   # half-space model, halved (worker.py:129)


def test_results_writer_layout(tmp_path):
    m = Model(["A0.4M6.0N", "A2.0M0.5N"])
    d = np.array([1.0, 1.1, 1.2])
    m.logs = {"A0.4M6.0N": np.vstack([d, [5.0, 5.1234567, 6.0]]).T, "A2.0M0.5N": np.vstack([d, [7.0, 8.0, np.nan]]).T}
    files = m.save_results(str(tmp_path))
    lines = open(files[0]).read().splitlines()
    assert lines[0] == "DEPTH\tA0.4M6.0N\tA2.0M0.5N" and lines[1] == "M\tOHMM\tOHMM"
    assert lines[3] == "1.1000\t5.1235\t8.0000" and lines[4].endswith("nan")


def test_results_picture(tmp_path, examples_dir):
    """save_results draws what the reference draws (remo3d.py:993-1147): the formation panel (a polygon per layer, one more per
    invaded zone, the borehole; coloured by resistivity) and the log tracks, as Results_plot.png beside the tables - here for
    Example_01 with the reference's own committed log as the curves.  The model and the logs are left as they were."""
    from remo3d_amd import plotting
    ex = os.path.join(examples_dir, "Example_01")
    tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
    gold = np.loadtxt(os.path.join(ex, "Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)
    m = Model(tools)
    m.set_model_parameters(os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"))
    m.logs = {t: np.column_stack([gold[:, 0], gold[:, 1 + i]]) for i, t in enumerate(tools)}
    m.logs[tools[1]][10:13, 1] = np.nan                      # a failed batch
    before_f, before_l = m.formation_model.copy(), {k: v.copy() for k, v in m.logs.items()}
    polys, res = plotting.model_polygons(m.formation_model, m.borehole_model, m.dip_deg, [0.0, 25.0], [-1.0, 1.0])
    invaded = int(np.sum(~np.isnan(m.formation_model[:, 2])))
    assert len(polys) == len(res) == m.formation_model.shape[0] + invaded + 1
    assert res[1] == 18.0 and res[2] == 3.0                  # second layer: virgin zone across the picture, flushed zone on top of it
    assert abs(res[-1] - np.mean(m.borehole_model[:, 2])) < 1e-12 and polys[-1].shape == (2 * m.borehole_model.shape[0], 2)
    files = m.save_results(str(tmp_path))
    png = [f for f in files if f.endswith("Results_plot.png")]
    assert len(png) == 1 and os.path.getsize(png[0]) > 20000 and open(png[0], "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    files = m.save_results(str(tmp_path / "two_tracks"), plot_layout=[tools[:3], tools[3:]], logs_at_nan="continue", logs_interpolation_factor=2,
                           plot_depth_lim=[5.0, 20.0], model_res_lim=[1.0, 20.0], logs_res_lim=[0.0, 30.0], logs_colours=[["r", "g", "b"], ["k", "c", "m"]])
    assert os.path.getsize([f for f in files if f.endswith(".png")][0]) > 20000
    fig = m.save_results(None, plot_layout=[tools[:2], tools[2:4], tools[4:]])
    assert len(fig.axes) >= 1 + 3 + 1                         # model panel, three tracks (+ their twinned axes), colour bar
    import matplotlib.pyplot as plt
    plt.close(fig)
    with pytest.raises(ValueError):
        m.save_results(str(tmp_path / "bad"), logs_at_nan="skip")
    assert np.array_equal(m.formation_model, before_f, equal_nan=True)
    assert all(np.array_equal(m.logs[k], before_l[k], equal_nan=True) for k in before_l)
    assert m.save_results(str(tmp_path / "tables_only"), plot=False)[-1].endswith(".txt")


@pytest.mark.parametrize("fixture", ["netgen_windows_example_01.json", "netgen_windows_example_01_r5.json", "netgen_windows_bm2.json",
                                     "netgen_windows_bm2_r8.json", "netgen_windows_thin_bedded.json"])
def test_netgen_path_windowing_matches_reference(fixture, examples_dir):
    gold = load(fixture)
    m = Model(["A0.4M6.0N", "A2.0M0.5N"])
    m.set_model_parameters(os.path.join(examples_dir, gold["formation_file"]), os.path.join(examples_dir, gold["borehole_file"]))
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    for case in gold["cases"]:
        fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, case["rm"], case["depth"], gold["R"])
        np.testing.assert_allclose(fg, np.array(case["formation_geometry"], dtype=float), rtol=1e-13, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(bh, np.array(case["borehole_geometry"]), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(sigma, case["sigma"], rtol=1e-13)


def test_mesh_workers_deliver_the_same_meshes_as_inline_generation():
    """simulate_logs(mesh_workers=N): the default mesher runs ahead of the solver in spawned processes; every
    batch must receive exactly the mesh the inline path builds (seeded mesher, same window)."""
    from remo3d_amd.model import Model

    class RecordingContext:
        def __init__(self):
            self.seen = []

        def solve_batch(self, mesh, sigma, sources, evals, opts):
            self.seen.append((mesh.dim, mesh.n_nodes, mesh.n_elems, float(mesh.coords.sum()), int(mesh.mat.sum()), len(sigma)))
            return [np.ones(len(e)) for e in evals], {}, 0

        def close(self):
            pass

    form = np.array([[0.0, 12.0, np.nan, np.nan, 7.0], [12.0, 13.5, 0.6, 2.0, 30.0], [13.5, 60.0, np.nan, np.nan, 4.0]])
    bore = np.array([[0.0, 0.2, 0.5], [60.0, 0.2, 0.5]])
    seen = {}
    for workers in (0, 2):
        m = Model(["A0.4M6.0N"])
        m.set_model_parameters(form, bore)
        m.ctx = RecordingContext()
        m.simulate_logs(np.arange(10.0, 16.0, 0.5), domain_radius=50, batch_size=4, mesh_scale=3.0, verbose=False, mesh_workers=workers)
        seen[workers] = m.ctx.seen
        assert not np.isnan(m.logs["A0.4M6.0N"][:, 1]).any()
    assert len(seen[0]) == 3 and seen[0] == seen[2]


def test_vertex_solver_rule_of_the_model():
    """model.vertex_solver_options: what `Model` asks of the library for the P1 block when the caller did not choose - 2D: nothing (the
    library's default, the multigrid cycle); 3D on its own conforming meshes: the cycle with the tuned polynomial behind it, whatever the
    number of contexts; 3D on other meshes: the cycle (with fallback) only when several contexts share the GPU.  The keywords are
    make_opts keywords."""
    from remo3d_amd import model, solver
    assert model.DEFAULT_CONTEXTS >= 2
    assert model.vertex_solver_options(2, 80000, False, 3) == {}
    assert model.vertex_solver_options(3, 83000, False, 1) == {}
    assert model.vertex_solver_options(3, 83000, False, 3) == dict(coarse="amg_or_chebyshev")
    for n_ctx in (1, 3):
        o = model.vertex_solver_options(3, 64000, True, n_ctx)
        assert o["coarse"] == "amg_or_chebyshev" and 6 <= o["coarse_degree"] <= 16 and 150 <= o["coarse_ratio"] <= 1200
        opts = solver.make_opts(**o)
        assert opts.coarse == 3 and opts.coarse_degree == o["coarse_degree"]
    with pytest.raises(ValueError):
        solver.make_opts(coarse="ilu")


def test_package_import_asks_for_eight_hardware_queues_unless_the_caller_chose():
    """remo3d_amd/__init__.py: GPU_MAX_HW_QUEUES (read by the HIP runtime at its first call) defaults to 8 - one queue per context
    stream of a Model with DEFAULT_CONTEXTS - and a caller's own value is left alone."""
    import subprocess, sys
    code = "import os, remo3d_amd, remo3d_amd.model as m; print(os.environ['GPU_MAX_HW_QUEUES'], m.DEFAULT_CONTEXTS, m.MAX_CONTEXTS)"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == 8 and int(out[1]) <= int(out[0]) and int(out[1]) <= int(out[2])
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(env, GPU_MAX_HW_QUEUES="4"), capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == 4


def test_initialize_workers_opens_the_default_number_of_contexts_and_caps_the_callers():
    """Model.initialize_workers (remo3d.py:552-599's keywords): gpu_workers = 0 opens model.DEFAULT_CONTEXTS contexts on the rank's GPU,
    gpu_workers = k opens k of them, at most model.MAX_CONTEXTS (one hardware queue each: remo3d_amd/__init__.py)."""
    from remo3d_amd import model

    class Stub:
        opened = 0

        def __init__(self, device):
            Stub.opened += 1

        def close(self):
            pass

    for asked, expect in ((0, model.DEFAULT_CONTEXTS), (1, 1), (3, 3), (50, model.MAX_CONTEXTS)):
        Stub.opened = 0
        m = Model(["A0.4M6.0N"])
        m.initialize_workers(cpu_workers=1, gpu_workers=asked, context_factory=Stub)
        assert Stub.opened == expect and 1 + len(m.extra_ctx) == expect
        m.shutdown_workers()
