"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64 everywhere): assembled CSR values 1e-12 relative to the largest entry (the two
sides integrate differently: exact reference tensors vs per-element quadrature); SpMV 1e-13
relative; potentials at axis points 1e-8 relative (both PCGs converged to rtol 1e-12, so the
difference is solver error, far below the north star's 1e-6 on apparent resistivity).
"""
import numpy as np
import pytest

from conftest import SIGMA3

pytestmark = pytest.mark.gpu

SRC = [([0.0], [1.0]), ([0.1], [1.0]), ([-0.1, 0.1], [1.0, -1.0])]
EVAL = [[0.4, 6.4, -2.0], [2.1, 2.6], [0.5, 3.0, 0.0]]


def _oracle_solve(mesh, sigma, condense):
    from oracle.fem_oracle import Oracle
    o = Oracle(mesh, sigma, condense=condense)
    outs = []
    for (z, I), ez in zip(SRC, EVAL):
        f, se, sf = o.rhs(z, I)
        u, it, rr, rc = o.pcg(f, 1e-12, 50000)
        assert rc == 0
        outs.append(o.eval(u, ez, (se, sf)))
    return o, outs


@pytest.mark.parametrize("which,condense", [("2d", True), ("2d", False), ("3d", True)])
def test_assembly_and_solve_match_oracle(which, condense, mesh2d, mesh3d, gpu_ctx):
    from remo3d_amd import solver
    mesh = mesh2d if which == "2d" else mesh3d
    o, ref = _oracle_solve(mesh, SIGMA3, condense)
    b = gpu_ctx.batch(mesh, SIGMA3, SRC, EVAL)
    rc = b.run(solver.make_opts(preconditioner="local", condense=condense, rtol=1e-12, maxsteps=20000))
    assert rc == 0, gpu_ctx.last_error()
    st = b.stats
    assert st["n_free"] == o.nfree and st["nnz"] == o.nnz
    rowptr, col, val, dinv, freeid = b.system()
    rp, oc, ov = o.csr()
    assert np.array_equal(rowptr, rp) and np.array_equal(col, oc)
    assert np.array_equal(freeid, o.freeid())
    assert np.max(np.abs(val - ov)) <= 1e-12 * np.max(np.abs(ov))
    diag = ov[[np.searchsorted(oc[rp[i]:rp[i + 1]], i) + rp[i] for i in range(0, o.nfree, 97)]]
    assert np.allclose(dinv[::97] * diag, 1.0, rtol=1e-12)
    got = b.fetch()
    for g, r in zip(got, ref):
        assert np.all(np.isfinite(g))
        assert np.max(np.abs(g - r)) <= 1e-8 * np.max(np.abs(r)), (g, r)
    # iteration counts of the two Jacobi-PCGs agree closely (same algorithm, different summation order)
    assert max(st["iterations"][:3]) > 10
    b.close()


@pytest.mark.parametrize("which", ["2d", "3d"])
@pytest.mark.parametrize("k", [1, 3, 5, 8])
def test_spmv_matches_oracle(which, k, mesh2d, mesh3d, gpu_ctx):
    from remo3d_amd import solver
    mesh = mesh2d if which == "2d" else mesh3d
    from oracle.fem_oracle import Oracle
    o = Oracle(mesh, SIGMA3, condense=True)
    b = gpu_ctx.batch(mesh, SIGMA3, SRC[:1], EVAL[:1])
    b.run(solver.make_opts(preconditioner="local", rtol=1e-2))
    rng = np.random.default_rng(k)
    x = rng.standard_normal((o.nfree, k))
    y, ms = b.spmv(x if k > 1 else x[:, 0], reps=3)
    y = y.reshape(o.nfree, k)
    for c in range(k):
        yr = o.spmv(x[:, c])
        assert np.max(np.abs(y[:, c] - yr)) <= 1e-13 * np.max(np.abs(yr)) * 50
    b.close()


@pytest.mark.parametrize("k", [1, 5, 8])
def test_spmv_row_schedules_agree(k, mesh3d, gpu_ctx):
    """The row schedules of the pair SpMM (remo_debug_tune key 3: 0 plain, 1 XCD windows, 16 * nc XCD regions of nc chunks; the
    library picks regions only above 20 M stored entries, which no test mesh reaches) walk the rows in different orders but
    sum every row the same way: results are bit-identical, also when chunks are nearly or completely empty (nc = 512), and a
    solve under the region schedule gives the potentials of the default one."""
    from remo3d_amd import _lib, solver
    from oracle.fem_oracle import Oracle
    L = _lib.load()
    o = Oracle(mesh3d, SIGMA3, condense=True)
    b = gpu_ctx.batch(mesh3d, SIGMA3, SRC[:2], EVAL[:2])
    try:
        b.run(solver.make_opts(rtol=1e-9, op="csr"))
        u_default = b.fetch()[0].copy()
        x = np.random.default_rng(k).standard_normal((o.nfree, k))
        xx = x if k > 1 else x[:, 0]
        ref = None
        for mapping in (1, 0, 16, 64, 16 * 37, 16 * 512):
            L.remo_debug_tune(3, mapping)
            y, _ = b.spmv(xx, reps=2)
            if ref is None:
                ref = y
                yr = np.stack([o.spmv(x[:, c]) for c in range(k)], 1).reshape(ref.shape)
                assert np.max(np.abs(ref - yr)) <= 5e-12 * np.max(np.abs(yr))
            assert np.array_equal(y, ref), mapping
        L.remo_debug_tune(3, 64)
        b.run(solver.make_opts(rtol=1e-9, op="csr"))
        assert np.max(np.abs(b.fetch()[0] - u_default)) <= 1e-9 * np.max(np.abs(u_default))
    finally:
        L.remo_debug_tune(3, -1)
        b.close()


@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_more_rhs_than_one_chunk(precision, mesh2d, gpu_ctx):
    """11 right-hand sides -> chunks of 8 + 3; every column must equal its single-RHS solve (in the mixed
    mode too, where every chunk runs its own refinement cycles)."""
    from remo3d_amd import solver
    zs = np.linspace(-0.5, 0.5, 11)
    src = [([z], [1.0]) for z in zs]
    ev = [[z + 0.4, z + 6.4] for z in zs]
    opts = solver.make_opts(preconditioner="local", rtol=1e-12, maxsteps=20000, precision=precision)
    outs, st, rc = gpu_ctx.solve_batch(mesh2d, SIGMA3, src, ev, opts)
    assert rc == 0
    for i in (0, 7, 8, 10):
        single, _, rc1 = gpu_ctx.solve_batch(mesh2d, SIGMA3, [src[i]], [ev[i]], opts)
        assert rc1 == 0
        assert np.allclose(outs[i], single[0], rtol=1e-9, atol=0)


@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_full_chunk_of_eight_with_the_two_level_preconditioner_in_3d(precision, mesh3d, gpu_ctx):
    """9 right-hand sides on a 3D mesh -> a chunk of 8 (the widest instantiation of every PCG kernel, including the update
    launch that carries the first Chebyshev step) and a chunk of 1: every column equals its single-RHS solve."""
    from remo3d_amd import solver
    zs = np.linspace(-0.4, 0.4, 9)
    src = [([z], [1.0]) for z in zs]
    ev = [[z + 0.4, z + 6.4] for z in zs]
    opts = solver.make_opts(preconditioner="multigrid", rtol=1e-11, maxsteps=5000, precision=precision)
    outs, st, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, src, ev, opts)
    assert rc == 0
    for i in (0, 4, 7, 8):
        single, _, rc1 = gpu_ctx.solve_batch(mesh3d, SIGMA3, [src[i]], [ev[i]], opts)
        assert rc1 == 0
        assert np.allclose(outs[i], single[0], rtol=1e-8, atol=0), i


@pytest.mark.parametrize("dim", [2, 3])
def test_homogeneous_medium_gives_true_resistivity(dim, gpu_ctx):
    """Physics identity of remo3d.py:285-306: in a homogeneous medium Ra == R for every tool."""
    from remo3d_amd import solver
    from remo3d_amd.meshgen import make_mesh
    R, sigma = 50.0, 0.1
    mesh = make_mesh(dim, R, [0.0], scale=1.0 if dim == 2 else 3.0, snap_z=[0.4, 6.4, 2.0, 2.5])
    outs, st, rc = gpu_ctx.solve_batch(mesh, [sigma], [([0.0], [1.0])], [[0.4, 6.4, 2.0, 2.5]],
                                       solver.make_opts(preconditioner="local", rtol=1e-10, maxsteps=20000))
    assert rc == 0
    u = outs[0] / (1.0 if dim == 2 else 2.0)   # half-space model carries twice the potential (worker.py:129)
    ra_normal = abs(4 * np.pi * 0.4 * 6.4 / 6.0 * (u[1] - u[0]))
    ra_lateral = abs(4 * np.pi * 2.0 * 2.5 / 0.5 * (u[3] - u[2]))
    assert abs(ra_normal - 10.0) < 5e-3 and abs(ra_lateral - 10.0) < 5e-3, (ra_normal, ra_lateral)


def test_error_paths_fill_nan(mesh2d, gpu_ctx):
    from remo3d_amd import solver
    # evaluation point outside the domain -> REMO_ERR_POINT, NaN outputs (worker.py:135-138 convention)
    outs, st, rc = gpu_ctx.solve_batch(mesh2d, SIGMA3, [([0.0], [1.0])], [[0.4, 80.0]], solver.make_opts(), raise_on_error=False)
    assert rc == -4 and np.all(np.isnan(outs[0]))
    # too few steps -> warning code, finite outputs (the reference is silent, ngsolve_functions.py:50)
    outs, st, rc = gpu_ctx.solve_batch(mesh2d, SIGMA3, [([0.0], [1.0])], [[0.4]],
                                       solver.make_opts(preconditioner="local", rtol=1e-12, maxsteps=5), raise_on_error=False)
    assert rc == 1 and np.all(np.isfinite(outs[0]))
    # zero-strength sources are skipped (ngsolve_functions.py:43): all-zero RHS gives u == 0
    outs, st, rc = gpu_ctx.solve_batch(mesh2d, SIGMA3, [([0.0], [0.0])], [[0.4]], solver.make_opts(), raise_on_error=False)
    assert rc == 0 and outs[0][0] == 0.0


def test_solvebvp_plugin_interface_matches_batch(mesh3d, gpu_ctx):
    """The per-RHS module interface of the reference (worker.py:100-131) gives the same potentials
    as the batch call."""
    from remo3d_amd import ngsolve_functions_hip as ngsf
    from remo3d_amd import solver
    mesh = ngsf.Mesh(mesh3d)
    sigma = ngsf.CoefficientFunction(SIGMA3)
    fes, gfu = ngsf.SolveBVP(mesh, sigma, np.array([-0.3, 0.0, 2.0]), np.array([0.0, 1.0, 0.0]), "dirichlet_boundary", "local", True)
    assert fes.ndof > fes.nfree > 0
    got = np.array([gfu(mesh(0.0, 0.0, z)) for z in (0.4, 6.4, -2.0)])
    ref, st, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, [([0.0], [1.0])], [[0.4, 6.4, -2.0]], solver.make_opts(preconditioner="local"))
    assert rc == 0 and np.allclose(got, ref[0], rtol=1e-12)
    with pytest.raises(solver.RemoError):
        gfu(mesh(0.0, 0.0, 75.0))          # outside the domain


def test_model_end_to_end_example_01(examples_dir, gpu_ctx):
    """Model.compute_synthetic_logs on the reference's Example_01 inputs against its committed
    output log: the end-to-end pin of the whole path (tool tables, batching, windowing, conforming
    mesh, assembly, two-level PCG, evaluation, Ra) on the reference's OWN results.  The meshes differ
    (in-repo interface-conforming Delaunay mesh vs the reference's Netgen mesh), so the tolerance is a
    mesh tolerance, not the 1e-6 solver tolerance: the reference's own two runs (Example_01 vs 02)
    differ by up to 3.1e-4, median 2e-5."""
    import os
    from remo3d_amd.model import Model
    tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
    gold = np.loadtxt(os.path.join(examples_dir, "Example_01/Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)
    depths = np.arange(0, 25.1, 2.5)
    m = Model.compute_synthetic_logs(tools, depths, os.path.join(examples_dir, "Example_01/Input/Formation.txt"),
                                     os.path.join(examples_dir, "Example_01/Input/Borehole.txt"), gpu_workers=1, verbose=False)
    rows = np.rint(depths / 0.1).astype(int)
    rel = []
    for i, t in enumerate(tools):
        assert np.all(np.isfinite(m.logs[t][:, 1]))
        rel.append(np.abs(m.logs[t][:, 1] - gold[rows, 1 + i]) / gold[rows, 1 + i])
    rel = np.array(rel)
    print("Example_01 vs reference log: median rel diff %.2e, max %.2e" % (np.median(rel), rel.max()))
    # measured (full log, 1506 points, tools/run_example01.py): see profiles/r01_b_example01_parity.json
    assert np.median(rel) < 1e-3 and rel.max() < 2e-2


@pytest.mark.parametrize("which", ["2d", "3d"])
def test_two_level_preconditioner_same_solution_fewer_steps(which, mesh2d, mesh3d, gpu_ctx):
    """preconditioner="multigrid" (vertex-block Chebyshev + Jacobi) changes the iteration, not the
    solution: potentials agree with the Jacobi-PCG ("local") and with the oracle to 1e-8."""
    from remo3d_amd import solver
    mesh = mesh2d if which == "2d" else mesh3d
    o, ref = _oracle_solve(mesh, SIGMA3, True)
    res = {}
    for pre in ("local", "multigrid"):
        outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(preconditioner=pre, rtol=1e-12, maxsteps=20000))
        assert rc == 0
        res[pre] = (outs, max(st["iterations"][:3]))
        for g, r in zip(outs, ref):
            assert np.max(np.abs(g - r)) <= 1e-8 * np.max(np.abs(r))
    print(which, "PCG steps: local", res["local"][1], "multigrid", res["multigrid"][1])
    assert res["multigrid"][1] < res["local"][1]
    # other degrees / intervals are valid preconditioners too
    outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-12, maxsteps=20000, coarse_degree=3, coarse_ratio=5))
    assert rc == 0
    for g, r in zip(outs, ref):
        assert np.max(np.abs(g - r)) <= 1e-8 * np.max(np.abs(r))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["2d", "3d"])
@pytest.mark.parametrize("pre", ["local", "multigrid"])
def test_mixed_precision_refines_to_the_fp64_answer(which, pre, mesh2d, mesh3d, gpu_ctx):
    """BASELINE config 5: PCG in fp32 storage inside an fp64 residual-refinement loop.  The stopping test
    is on the true fp64 residual, so the potentials agree with the oracle as the fp64 path does
    (1e-8 at rtol 1e-12; fp32 alone would stall near 1e-6), with several residual replacements."""
    from remo3d_amd import solver
    mesh = mesh2d if which == "2d" else mesh3d
    o, ref = _oracle_solve(mesh, SIGMA3, True)
    outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL,
                                       solver.make_opts(preconditioner=pre, rtol=1e-12, maxsteps=20000, precision="mixed"))
    assert rc == 0, st
    assert st["refinement_cycles"] >= 2
    assert max(st["relres"][:3]) <= 1e-12
    for g, r in zip(outs, ref):
        assert np.max(np.abs(g - r)) <= 1e-8 * np.max(np.abs(r))
    # the reference's default tolerance, other digits per cycle
    outs64, _, rc64 = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(preconditioner=pre, rtol=1e-8))
    for digits in (2, 5):
        outs32, st32, rc32 = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL,
                                                 solver.make_opts(preconditioner=pre, rtol=1e-8, precision="mixed", inner_digits=digits))
        assert rc64 == 0 and rc32 == 0
        for g, r in zip(outs32, outs64):
            assert np.max(np.abs(g - r)) <= 1e-6 * np.max(np.abs(r))


@pytest.mark.gpu
def test_mixed_precision_reports_non_convergence(mesh3d, gpu_ctx):
    """maxsteps bounds the SUM of the inner solves; running out gives rc 1 and finite potentials."""
    from remo3d_amd import solver
    outs, st, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-12, maxsteps=7, precision="mixed"))
    assert rc == 1
    assert all(np.all(np.isfinite(g)) for g in outs)


@pytest.mark.gpu
def test_3d_path_reproduces_the_axisymmetric_solution(gpu_ctx):
    """The same layered model (no dip) solved twice: axisymmetric 2D mesh / 2D kernels, and the conforming
    3D half-ball mesh / 3D kernels.  The half-space carries the full current, so u_3d = 2 u_2d
    (worker.py:129-131 divides the 3D reading by 2).  Different meshes, element types and quadrature-free
    tensors: agreement at the discretisation level (measured 5e-3 next to the source, 1.5e-3 at 6 m, 8 sectors;
    tools/run_3d_vs_2d.py shows it falling with the sector count) pins the 3D path to the 2D path, which is
    pinned to the reference's own logs."""
    from remo3d_amd import meshgen, solver
    R = 50.0
    fg = np.array([[-80.0, -1.0, np.nan], [-1.0, 1.5, 0.5], [1.5, 80.0, np.nan]])
    bh = np.array([[-80.0, 0.1], [80.0, 0.1]])
    sigma = [1.0 / 0.5, 1.0 / 20.0, 1.0 / 5.0, 1.0 / 50.0, 1.0 / 10.0]
    src = [([0.0], [1.0]), ([2.0], [1.0])]
    ev = [[0.4, 6.4], [0.5, -1.5]]
    polys = meshgen.layer_interfaces_2d(fg, bh, R)
    m2 = meshgen.make_mesh(2, R, sources_z=[0.0, 2.0], snap_z=[0.4, 6.4, 0.5, -1.5], scale=1.0, interfaces=polys,
                           material_fn=meshgen.layered_material_fn(2, fg, bh))
    m3 = meshgen.make_mesh_3d_conforming(R, fg, bh, 0.0, sources_z=[0.0, 2.0], snap_z=[0.4, 6.4, 0.5, -1.5], scale=1.0, sectors=8)
    u2, st2, rc2 = gpu_ctx.solve_batch(m2, sigma, src, ev, solver.make_opts(rtol=1e-10))
    u3, st3, rc3 = gpu_ctx.solve_batch(m3, sigma, src, ev, solver.make_opts(rtol=1e-10))
    assert rc2 == 0 and rc3 == 0
    for a, b in zip(u2, u3):
        assert np.max(np.abs(0.5 * b - a) / np.abs(a)) < 1e-2, (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["2d", "3d"])
def test_paired_chebyshev_steps_are_the_same_preconditioner(which, mesh2d, mesh3d, gpu_ctx):
    """Two Richardson factors of the Chebyshev polynomial per launch on B = A_vv D^-1 A_vv (remo_debug_tune key 6) is
    the same polynomial as one three-term-recurrence step per launch: same PCG step counts (+-1), same potentials."""
    from remo3d_amd import _lib, solver
    mesh = mesh2d if which == "2d" else mesh3d
    L = _lib.load()
    got = {}
    try:
        for mode in (0, 2):
            L.remo_debug_tune(6, mode)
            outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-10, coarse="chebyshev"))
            assert rc == 0 and st["coarse_used"] == 1
            got[mode] = (np.concatenate(outs), st["iterations"][:3])
    finally:
        L.remo_debug_tune(6, 1)
    assert np.max(np.abs(got[0][0] - got[2][0]) / np.abs(got[0][0])) < 1e-7
    assert max(abs(a - b) for a, b in zip(got[0][1], got[2][1])) <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_two_half_spaces_image_solution_on_the_gpu(precision, gpu_ctx):
    """Heterogeneous analytic pin: point source next to a plane interface between two conductivities (image
    solution), differences of axis potentials to 5e-3 (the grounded sphere at R = 50 m shifts them all alike)."""
    from remo3d_amd import solver
    from test_oracle import _two_half_spaces
    mesh, sigma, zs, exact = _two_half_spaces(1.0)
    outs, st, rc = gpu_ctx.solve_batch(mesh, sigma, [([0.0], [1.0])], [list(zs)], solver.make_opts(rtol=1e-10, precision=precision))
    assert rc == 0
    got = outs[0]
    d_got, d_ex = got[:-1] - got[1:], exact[:-1] - exact[1:]
    assert np.max(np.abs(d_got - d_ex) / np.abs(d_ex)) < 5e-3, (d_got, d_ex)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_ragged_and_degenerate_right_hand_sides(precision, mesh2d, gpu_ctx):
    """Edge cases of the batch entry (worker.py:104-131 never produces them, the ABI must still behave):
    a zero-strength source is skipped (ngsolve_functions.py:43), a right-hand side without evaluation points,
    one whose sources are all zero (u = 0, converged at once), an evaluation point on a source."""
    from remo3d_amd import solver
    opts = solver.make_opts(rtol=1e-10, precision=precision)
    src = [([0.0, 0.1], [1.0, 0.0]),      # second electrode carries no current
           ([0.1], [1.0]),                # no evaluation points at all
           ([0.0], [0.0]),                # nothing injected
           ([0.0], [1.0])]
    ev = [[0.4, 6.4], [], [0.4, 6.4], [0.0, 0.4]]
    outs, st, rc = gpu_ctx.solve_batch(mesh2d, SIGMA3, src, ev, opts)
    assert rc == 0, st
    ref, _, rc1 = gpu_ctx.solve_batch(mesh2d, SIGMA3, [([0.0], [1.0])], [[0.4, 6.4, 0.0]], opts)
    assert rc1 == 0
    assert np.allclose(outs[0], ref[0][:2], rtol=1e-8, atol=0)
    assert outs[1].size == 0
    assert np.all(outs[2] == 0.0)
    assert np.allclose(outs[3], [ref[0][2], ref[0][0]], rtol=1e-8, atol=0) and np.isfinite(outs[3][0]) and outs[3][0] > outs[3][1]


def _batch_properties(gpu_ctx, mesh, sigma, precision, tol=2e-8):
    """Properties of a batch that need no second solver: reciprocity u_a(z_b) = u_b(z_a) of the symmetric operator, linearity in the
    source strengths (dipole = difference of its poles), scaling of all conductivities by c (u -> u / c), and the two-level and the
    Jacobi preconditioner agreeing on the solution.  Returns (potentials of the first right-hand side, stats)."""
    from remo3d_amd import solver
    za, zb, zc = 0.0, 0.4, 6.4
    src = [([za], [1.0]), ([zb], [1.0]), ([za, zb], [1.0, -1.0]), ([za], [2.5])]
    ev = [[zb, zc], [za, zc], [zc, 2.0], [zb, zc]]
    opts = solver.make_opts(rtol=1e-11, precision=precision, maxsteps=3000)
    out, st, rc = gpu_ctx.solve_batch(mesh, sigma, src, ev, opts)
    assert rc == 0, st
    assert abs(out[0][0] - out[1][0]) <= tol * abs(out[0][0])                               # reciprocity
    assert abs(out[2][0] - (out[0][1] - out[1][1])) <= tol * abs(out[0][1])                 # linearity: dipole at z_c
    assert np.allclose(out[3], 2.5 * out[0], rtol=tol, atol=0)                              # strength scaling
    out_c, _, rc = gpu_ctx.solve_batch(mesh, 4.0 * sigma, src[:1], ev[:1], opts)
    assert rc == 0 and np.allclose(out_c[0], out[0] / 4.0, rtol=tol, atol=0)                # conductivity scaling
    out_j, st_j, rc = gpu_ctx.solve_batch(mesh, sigma, src[:1], ev[:1], solver.make_opts(preconditioner="local", rtol=1e-11, precision=precision, maxsteps=5000))
    assert rc == 0 and np.allclose(out_j[0], out[0], rtol=tol, atol=0)
    return out[0], st


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_full_size_batch_properties(precision, gpu_ctx):
    """A batch of the round-1/2 headline size (bench.py size S: BM3, dip 30 degrees, ~70 k tetrahedra, ~320 k unknowns, 15 M stored
    entries - below 200 k tetrahedra the whole matrix is assembled beside the patch operator) through its properties, and one
    right-hand side of the same batch against the oracle (a few seconds on one core)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle.fem_oracle import solve_batch as oracle_batch
    w = bench.build_workload(0, 1, 5, bench.SIZES["S"])["work"][0]
    mesh, sigma = w["mesh"], np.asarray(w["sigma"], dtype=np.float64)
    u0, st = _batch_properties(gpu_ctx, mesh, sigma, precision)
    assert st["n_free"] > 250000 and st["nnz"] > 12000000 and st["op_used"] == 3, st
    ref, rc_o, st_o = oracle_batch(mesh, sigma, [0, 1], [0.0], [1.0], [0, 2], [0.4, 6.4], condense=True, rtol=1e-11, maxit=5000)
    assert rc_o == 0 and st_o["n"] == st["n_free"] and st_o["nnz"] == st["nnz"]
    assert np.allclose(u0, ref, rtol=2e-8, atol=0), (u0, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_headline_size_batch_on_the_default_path(precision, gpu_ctx, size_L_case):
    """The HEADLINE workload at its own size (bench.py default, size L: batch 0 of the 100-depth sweep, ~370 k tetrahedra, ~1.7 M
    unknowns) on the path the library picks there by itself: patch operator (op_used 3), only the Jacobi diagonal and the P1
    block assembled (nnz 0: `assemble` by size), Chebyshev polynomial on the vertex block - the same properties as at size S, and
    one right-hand side against the oracle (started in a thread by the fixture: ~1 minute of one core)."""
    w = size_L_case["work"][0]
    mesh, sigma = w["mesh"], np.asarray(w["sigma"], dtype=np.float64)
    u0, st = _batch_properties(gpu_ctx, mesh, sigma, precision)
    assert mesh.n_elems > 300000 and st["n_free"] > 1400000, (mesh.n_elems, st["n_free"])
    assert st["op_used"] == 3 and st["nnz"] == 0 and st["coarse_used"] == 1, st
    ref, rc_o, st_o = size_L_case["futs"]["properties"].result(timeout=900)
    assert rc_o == 0 and st_o["n"] == st["n_free"]
    assert np.allclose(u0, ref, rtol=2e-8, atol=0), (u0, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_chebyshev_launch_folds_keep_the_preconditioner(precision, mesh3d, gpu_ctx):
    """The first Chebyshev step rides on the update launch for degree >= 3 (remo_debug_tune key 9, default on) and the last
    term on the launch before it for degree >= 2: the same polynomial, so the same potentials and (up to rounding in the
    first operand, formed per gathered entry instead of from the stored residual) the same step counts as with one launch per
    step; degrees 1 and 2 take the unfolded route."""
    from remo3d_amd import _lib, solver
    L = _lib.load()
    b = gpu_ctx.batch(mesh3d, SIGMA3, SRC, EVAL)
    try:
        ref = None
        for degree in (1, 2, 3, 4, 7):
            res = {}
            for fold in (0, 1):
                L.remo_debug_tune(9, fold)
                rc = b.run(solver.make_opts(rtol=1e-10, coarse_degree=degree, coarse_ratio=30, precision=precision))
                assert rc == 0
                res[fold] = (np.concatenate(b.fetch()), b.stats["pcg_steps"])
            # (the patch operator sums in an order that varies from run to run - LDS and <p, A p> atomics: two fp64 solves agree
            # to 1e-11, two solves of the mixed mode, whose refinement cycles end on a threshold, to a few 1e-8)
            tol = 1e-8 if precision == "fp64" else 3e-7
            assert np.allclose(res[0][0], res[1][0], rtol=tol, atol=0), degree
            assert abs(res[0][1] - res[1][1]) <= max(2, res[0][1] // (50 if precision == "fp64" else 20)), (degree, res[0][1], res[1][1])
            if ref is None:
                ref = res[1][0]
            assert np.allclose(res[1][0], ref, rtol=max(tol, 1e-7), atol=0), degree
    finally:
        L.remo_debug_tune(9, 1)
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_compact_vertex_block_is_the_same_operator(precision, mesh3d, gpu_ctx):
    """Above 16 k vertices the Chebyshev launches read a compact copy of the vertex block instead of the leading entries of A's
    rows (remo_debug_tune key 13; 2 forces it on the small test mesh): same entries in the same order, so the same steps and
    potentials, with the first step inside the update launch (key 9) or on its own."""
    from remo3d_amd import _lib, solver
    L = _lib.load()
    b = gpu_ctx.batch(mesh3d, SIGMA3, SRC, EVAL)
    try:
        res = {}
        for fold in (1, 0):
            for compact in (0, 2):
                L.remo_debug_tune(9, fold); L.remo_debug_tune(13, compact)
                # (bit for bit: on the CSR product - the patch operator's LDS sums are unordered)
                assert b.run(solver.make_opts(rtol=1e-10, precision=precision, op="csr")) == 0
                res[(fold, compact)] = (np.concatenate(b.fetch()), b.stats["pcg_steps"])
            assert res[(fold, 0)][1] == res[(fold, 2)][1], fold
            assert np.array_equal(res[(fold, 0)][0], res[(fold, 2)][0]), fold
    finally:
        L.remo_debug_tune(9, 1); L.remo_debug_tune(13, 1)
        b.close()


def _hub_mesh():
    """A ball of tetrahedra around a HUB vertex with ~45 neighbours (a shell of points with nothing else inside it): vertex rows of
    the P1 block far longer than the 24 entries of the fixed-width image, next to ordinary ones."""
    from scipy.spatial import Delaunay
    from remo3d_amd.meshgen import Mesh, _boundary_facets
    rng = np.random.default_rng(11)

    def sphere(m, radius):
        v = rng.standard_normal((m, 3))
        return radius * v / np.linalg.norm(v, axis=1)[:, None]
    hub = np.array([[0.0, 0.0, 0.05]])
    pts = np.concatenate([hub, hub + sphere(44, 1.0), sphere(160, 2.2) * rng.uniform(0.85, 1.0, (160, 1)),
                          sphere(300, 4.0) * rng.uniform(0.7, 1.0, (300, 1)), sphere(260, 6.0)])
    conn = Delaunay(pts).simplices.astype(np.int32)
    p = pts[conn]
    vol = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])
    conn = conn[np.abs(vol) > 1e-9]
    flip = vol[np.abs(vol) > 1e-9] < 0
    conn[flip] = conn[flip][:, [1, 0, 2, 3]]
    mat = (np.linalg.norm(pts[conn].mean(axis=1), axis=1) > 2.0).astype(np.int32)
    bconn = _boundary_facets(conn).astype(np.int32)
    return Mesh(3, np.ascontiguousarray(pts), np.ascontiguousarray(conn), mat, bconn, np.ones(len(bconn), np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_fixed_width_vertex_block_with_long_rows(precision, gpu_ctx):
    """The Chebyshev launches read the first 24 entries of a vertex row from a fixed-width image and the rest from the CSR form
    (remo_debug_tune key 24; kernels.hip k_vblock_ell): on a mesh whose hub vertex has ~45 neighbours the image, the CSR form and
    the oracle give the same potentials, with the block read in place or from its compact copy (key 13) and the first step
    inside the update launch or on its own (key 9)."""
    from remo3d_amd import _lib, solver
    from oracle.fem_oracle import Oracle
    L = _lib.load()
    mesh = _hub_mesh()
    import itertools
    pairs = np.unique(np.sort(np.concatenate([mesh.conn[:, [i, j]] for i, j in itertools.combinations(range(4), 2)]), axis=1), axis=0)
    deg = np.bincount(pairs.ravel(), minlength=mesh.n_nodes) + 1
    assert deg[0] > 40 and (deg > 24).sum() >= 3 and np.median(deg) < 24, (deg[0], (deg > 24).sum())
    sigma = [0.5, 0.05]
    src, ev = [([0.4], [1.0]), ([-0.6, 0.9], [1.0, -1.0])], [[1.2, -1.5, 0.05], [0.0, 2.5]]
    o = Oracle(mesh, sigma, condense=True)
    ref = []
    for (z, I), ez in zip(src, ev):
        f, se, sf = o.rhs(z, I)
        u, it, rr, rc = o.pcg(f, 1e-12, 50000)
        assert rc == 0
        ref.append(o.eval(u, ez, (se, sf)))
    b = gpu_ctx.batch(mesh, sigma, src, ev)
    try:
        got = {}
        for ell in (0, 1):
            for compact in (0, 2):
                for fold in (1, 0):
                    L.remo_debug_tune(24, ell); L.remo_debug_tune(13, compact); L.remo_debug_tune(9, fold)
                    assert b.run(solver.make_opts(rtol=1e-11, precision=precision, coarse_degree=6, coarse_ratio=30, op="csr")) == 0
                    assert b.stats["coarse_used"] == 1
                    got[(ell, compact, fold)] = (np.concatenate(b.fetch()), b.stats["pcg_steps"])
        base = got[(0, 0, 1)]
        for key, (u, steps) in got.items():
            assert np.allclose(u, base[0], rtol=1e-8, atol=0), key
            # (fp32 inner solves: the image sums a row in another order than the CSR walk, and the refinement cycles amplify that)
            assert abs(steps - base[1]) <= max(2, base[1] // (50 if precision == "fp64" else 15)), (key, steps, base[1])
        assert np.allclose(base[0], np.concatenate(ref), rtol=(1e-7 if precision == "fp64" else 1e-6), atol=0)
    finally:
        L.remo_debug_tune(24, 1); L.remo_debug_tune(13, 1); L.remo_debug_tune(9, 1)
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nrhs", [1, 2, 8])
def test_patch_operator_on_a_hub_mesh(nrhs, gpu_ctx):
    """A vertex shared by ~80 tetrahedra: its row is spread over several patches (boundary slab slots summed by the update
    launch), with 256 / nrhs elements per patch; same potentials as the CSR product, with and without the assembled matrix."""
    from remo3d_amd import solver
    mesh = _hub_mesh()
    sigma = [0.5, 0.05]
    zs = np.linspace(-1.4, 1.6, nrhs)
    src = [([float(z)], [1.0]) for z in zs]
    ev = [[float(z) + 0.7, 0.05] for z in zs]
    b = gpu_ctx.batch(mesh, sigma, src, ev)
    try:
        assert b.run(solver.make_opts(rtol=1e-11, op="csr")) == 0 and b.stats["op_used"] == 0
        ref = np.concatenate(b.fetch())
        for assemble in ("full", "vertex_block"):
            assert b.run(solver.make_opts(rtol=1e-11, op="patch", assemble=assemble)) == 0 and b.stats["op_used"] == 3
            assert np.allclose(np.concatenate(b.fetch()), ref, rtol=1e-8, atol=0), assemble
    finally:
        b.close()


@pytest.mark.gpu
def test_operator_option_in_2d_and_removed_values(mesh2d, gpu_ctx):
    """A 2D batch runs on the CSR product whatever `op` says (the patch operator is 3D only: op = "patch" used to turn every 2D
    batch of a sweep into NaN); op = 1, the round-2 element-wise operator, left the library with ABI 7 and is an argument error."""
    from remo3d_amd import solver
    a, sa, rca = gpu_ctx.solve_batch(mesh2d, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-10, op="patch"))
    c, sc, rcc = gpu_ctx.solve_batch(mesh2d, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-10, op="csr"))
    assert rca == 0 and rcc == 0 and sa["op_used"] == 0 and sc["op_used"] == 0 and sa["assembled"] == 1
    for u, v in zip(a, c):
        assert np.array_equal(u, v)
    o = solver.make_opts(rtol=1e-10)
    o.op = 1
    out, st, rc = gpu_ctx.solve_batch(mesh2d, SIGMA3, SRC, EVAL, o, raise_on_error=False)
    assert rc == solver.REMO_ERR_ARG
    assert all(np.all(np.isnan(u)) for u in out)


def _rhs_block(k):
    zs = np.linspace(-0.1, 0.1, k)
    return [([float(z)], [1.0]) for z in zs], [[float(z) + 0.4, float(z) + 6.4] for z in zs]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3, 5, 8])
def test_patch_operator_is_the_assembled_matrix(k, mesh3d, gpu_ctx):
    """remo_opts_t.op = 3 (3D): y = A x patch by patch (x rows staged in LDS, factorised reference tensors, LDS accumulation,
    boundary slab) equals the CSR product and the oracle's; also with fewer columns than the batch was laid out for.  The sums
    inside a patch are unordered: agreement to rounding, not bit for bit."""
    from remo3d_amd import solver
    from oracle.fem_oracle import Oracle
    o = Oracle(mesh3d, SIGMA3, condense=True)
    src, ev = _rhs_block(k)
    b = gpu_ctx.batch(mesh3d, SIGMA3, src, ev)
    try:
        ys = {}
        for kk in sorted({k, max(1, k - 1), 1}):
            x = np.random.default_rng(10 * k + kk).standard_normal((o.nfree, kk))
            xx = x if kk > 1 else x[:, 0]
            for op in ("csr", "patch"):
                b.run(solver.make_opts(preconditioner="local", rtol=1e-2, op=op))
                assert b.stats["op_used"] == (3 if op == "patch" else 0)
                ys[op], _ = b.spmv(xx, reps=2)
            yr = np.stack([o.spmv(x[:, c]) for c in range(kk)], 1).reshape(ys["csr"].shape)
            scale = np.max(np.abs(yr))
            assert np.max(np.abs(ys["patch"] - yr)) <= 5e-12 * scale, (k, kk)
            assert np.max(np.abs(ys["patch"] - ys["csr"])) <= 5e-12 * scale, (k, kk)
    finally:
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3, 5, 8])
def test_persistent_patch_kernel_is_the_same_operator(k, mesh3d, gpu_ctx):
    """The patch operator's two launch forms - one workgroup per patch (the default) and persistent workgroups that walk a run of
    patches while the next patch's rows, tables and element data arrive - by LDS-DMA (remo_debug_tune key 34 = 1) or through registers
    (key 34 = 2) - are the same operator: products agree with each other, with the CSR product and with the oracle's to rounding, also when a workgroup walks
    MANY patches (key 35: few workgroups per XCD; the small test mesh would otherwise give every workgroup one patch and never
    reach the prefetch) or exactly one or two, in fp64 and through a mixed-precision solve."""
    from remo3d_amd import _lib, solver
    from oracle.fem_oracle import Oracle
    L = _lib.load()
    if L.remo_debug_tune(34, 0) != 0:
        pytest.skip("the persistent forms are rejected experiments: they live in the -DREMO_PROBES build only (make -C remo3d_amd/csrc probes; "
                    "REMO_LIB=remo3d_amd/libremo3d_hip_probes.so); this run loaded the product library - record of them passing: profiles/r04_m_*")
    o = Oracle(mesh3d, SIGMA3, condense=True)
    src, ev = _rhs_block(k)
    b = gpu_ctx.batch(mesh3d, SIGMA3, src, ev)
    x = np.random.default_rng(100 + k).standard_normal((o.nfree, k))
    xx = x if k > 1 else x[:, 0]
    yr = np.stack([o.spmv(x[:, c]) for c in range(k)], 1).reshape(xx.shape)
    scale = np.max(np.abs(yr))
    try:
        assert b.run(solver.make_opts(preconditioner="local", rtol=1e-2, op="patch")) == 0 and b.stats["op_used"] == 3
        ys = {}
        for persist, wgs in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 7), (2, 0), (2, 1), (2, 2), (2, 3), (2, 7)):
            L.remo_debug_tune(34, persist); L.remo_debug_tune(35, wgs)
            ys[(persist, wgs)], _ = b.spmv(xx, reps=3)
            assert np.max(np.abs(ys[(persist, wgs)] - yr)) <= 5e-12 * scale, (persist, wgs)
        for form, wgs in ((1, 1), (1, 3), (2, 1), (2, 3), (2, 0)):   # whole solves, the rows shared by patches summed by the update launch, <p, A p> added by the workgroups
            for precision in ("fp64", "mixed"):
                L.remo_debug_tune(34, form); L.remo_debug_tune(35, wgs)
                assert b.run(solver.make_opts(rtol=1e-11, op="patch", precision=precision, maxsteps=5000)) == 0
                u1 = np.concatenate(b.fetch())
                assert np.max(b.true_relres()) < 5e-11
                L.remo_debug_tune(34, 0)
                assert b.run(solver.make_opts(rtol=1e-11, op="patch", precision=precision, maxsteps=5000)) == 0
                u0 = np.concatenate(b.fetch())
                assert np.allclose(u1, u0, rtol=1e-8 if precision == "fp64" else 3e-7, atol=0), (form, wgs, precision)
    finally:
        L.remo_debug_tune(34, 0); L.remo_debug_tune(35, 0)
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
@pytest.mark.parametrize("pre", ["local", "multigrid"])
def test_patch_operator_solves_like_the_csr_path(precision, pre, mesh3d, gpu_ctx):
    """The same PCG on the patch operator: potentials of the CSR path and of the oracle, similar step counts, the TRUE residual
    of the returned solution; 9 right-hand sides (chunks of 8 + 1: the single column runs through tables laid out for 8)."""
    from remo3d_amd import solver
    o, ref = _oracle_solve(mesh3d, SIGMA3, True)
    res = {}
    for op in ("csr", "patch"):
        outs, st, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC, EVAL, solver.make_opts(preconditioner=pre, rtol=1e-12, maxsteps=20000, precision=precision, op=op))
        assert rc == 0 and st["op_used"] == (3 if op == "patch" else 0)
        res[op] = (outs, max(st["iterations"][:3]))
        for g, r in zip(outs, ref):
            assert np.max(np.abs(g - r)) <= 1e-8 * np.max(np.abs(r))
    assert abs(res["csr"][1] - res["patch"][1]) <= max(3, res["csr"][1] // 20), (res["csr"][1], res["patch"][1])
    zs = np.linspace(-0.4, 0.4, 9)
    src = [([z], [1.0]) for z in zs]
    ev = [[z + 0.4, z + 6.4] for z in zs]
    a, _, rca = gpu_ctx.solve_batch(mesh3d, SIGMA3, src, ev, solver.make_opts(preconditioner=pre, rtol=1e-11, precision=precision, op="patch", maxsteps=5000))
    c, _, rcc = gpu_ctx.solve_batch(mesh3d, SIGMA3, src, ev, solver.make_opts(preconditioner=pre, rtol=1e-11, precision=precision, op="csr", maxsteps=5000))
    assert rca == 0 and rcc == 0
    for u, v in zip(a, c):
        assert np.allclose(u, v, rtol=1e-8, atol=0)
    b = gpu_ctx.batch(mesh3d, SIGMA3, SRC, EVAL)
    try:
        assert b.run(solver.make_opts(preconditioner=pre, rtol=1e-11, maxsteps=5000, precision=precision, op="patch")) == 0
        assert np.max(b.true_relres()) < 5e-11
    finally:
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_patch_operator_with_the_shared_rows_summed_by_the_update_launch(precision, mesh3d, gpu_ctx):
    """Inside the PCG the patch operator leaves the rows shared by several patches in its boundary slab and the update launch
    sums them while it reads q (no k_patch_reduce launch; remo_debug_tune 22 = 0 restores it).  Large meshes run that way by
    default; the small test mesh does once the first Chebyshev step is kept out of the update launch (key 9 = 0), which would
    gather q.  Same potentials as the CSR path, same step counts, true residual of the returned solution."""
    from remo3d_amd import _lib, solver
    L = _lib.load()
    ref, st0, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-11, maxsteps=5000, precision=precision, op="csr"))
    assert rc == 0
    L.remo_debug_tune(9, 0)
    b = gpu_ctx.batch(mesh3d, SIGMA3, SRC, EVAL)
    try:
        steps = {}
        for defer in (1, 0):
            L.remo_debug_tune(22, defer)
            assert b.run(solver.make_opts(rtol=1e-11, maxsteps=5000, precision=precision, op="patch")) == 0
            assert b.stats["op_used"] == 3
            steps[defer] = b.stats["pcg_steps"]
            for u, v in zip(b.fetch(), ref):
                assert np.allclose(u, v, rtol=1e-8, atol=0)
            assert np.max(b.true_relres()) < 5e-11
        assert abs(steps[0] - steps[1]) <= max(3, steps[0] // 20), steps
    finally:
        L.remo_debug_tune(9, 1); L.remo_debug_tune(22, 1)
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
@pytest.mark.parametrize("pre", ["local", "multigrid"])
def test_patch_operator_without_the_assembled_matrix(precision, pre, mesh3d, gpu_ctx):
    """remo_opts_t.assemble = 2 (what batches above 200 k tetrahedra do by default): only the Jacobi diagonal and the P1 block
    the preconditioner solves are assembled, the CG never reads stored entries.  Same potentials and step counts as with the
    whole matrix; the diagonal is the assembled one; the matrix inspection hook says why it has nothing to show; also with the
    multigrid cycle on the P1 block."""
    from oracle.fem_oracle import Oracle
    from remo3d_amd import solver
    ref, st0, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC, EVAL, solver.make_opts(preconditioner=pre, rtol=1e-11, maxsteps=5000, precision=precision, op="patch", assemble="full"))
    assert rc == 0 and st0["nnz"] > 0
    b = gpu_ctx.batch(mesh3d, SIGMA3, SRC, EVAL)
    try:
        for coarse in (("auto", "amg") if pre == "multigrid" else ("auto",)):
            assert b.run(solver.make_opts(preconditioner=pre, rtol=1e-11, maxsteps=5000, precision=precision, op="patch", assemble="vertex_block", coarse=coarse)) == 0
            assert b.stats["op_used"] == 3 and b.stats["nnz"] == 0
            for u, v in zip(b.fetch(), ref):
                assert np.allclose(u, v, rtol=1e-8, atol=0)
            if coarse == "auto":
                assert abs(b.stats["pcg_steps"] - st0["pcg_steps"]) <= max(3, st0["pcg_steps"] // 20)
        o = Oracle(mesh3d, SIGMA3, condense=True)
        rp, col, val = o.csr()
        diag = np.array([val[rp[i]:rp[i + 1]][col[rp[i]:rp[i + 1]] == i][0] for i in range(0, o.nfree, 97)])
        assert np.allclose(1.0 / b.jacobi()[::97], diag, rtol=1e-12)
        x = np.random.default_rng(0).standard_normal((o.nfree, 3))
        y, _ = b.spmv(x)
        yr = np.stack([o.spmv(x[:, c]) for c in range(3)], 1)
        assert np.max(np.abs(y - yr)) <= 5e-12 * np.max(np.abs(yr))
        with pytest.raises(solver.RemoError) as e:
            b.system()
        assert "assemble" in str(e.value)
        with pytest.raises(solver.RemoError):
            b.spmv(np.zeros((o.nfree, 8)))          # more columns than the batch's tables hold, and no matrix to fall back on
    finally:
        b.close()


@pytest.mark.gpu
def test_patch_operator_without_element_locality(mesh3d, gpu_ctx):
    """The library orders the elements itself (by their smallest vertices): a mesh handed over with shuffled elements gives
    the same potentials through the patch operator, and so does the shuffled list taken as it comes (remo_debug_tune 18 = 0:
    patches of unrelated elements hold up to 20 distinct rows per element - slower, not wrong)."""
    import copy
    from remo3d_amd import _lib, solver
    L = _lib.load()
    ref, _, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-11, op="csr"))
    assert rc == 0
    perm = np.random.default_rng(3).permutation(mesh3d.n_elems)
    sh = copy.copy(mesh3d)
    sh.conn = np.ascontiguousarray(mesh3d.conn[perm]); sh.mat = np.ascontiguousarray(mesh3d.mat[perm])
    for order in (1, 0):
        L.remo_debug_tune(18, order)
        try:
            outs, st, rc = gpu_ctx.solve_batch(sh, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-11, op="patch"))
        finally:
            L.remo_debug_tune(18, 1)
        assert rc == 0 and st["op_used"] == 3
        for u, v in zip(outs, ref):
            assert np.allclose(u, v, rtol=1e-8, atol=0)


@pytest.mark.gpu
def test_operator_and_assembly_choice(mesh3d, gpu_ctx):
    """op = "auto" (the default) in 3D: the patch operator wherever its tables fit; what is assembled beside it follows the size
    (the whole matrix up to 200 k tetrahedra: inspection hooks at a small cost) or `assemble`; the statistics say which."""
    from remo3d_amd import solver
    _, st, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC[:1], EVAL[:1], solver.make_opts(rtol=1e-6))
    assert rc == 0 and st["op_used"] == 3 and st["assembled"] == 1 and st["nnz"] > 0
    _, st, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC[:1], EVAL[:1], solver.make_opts(rtol=1e-6, assemble="vertex_block"))
    assert rc == 0 and st["op_used"] == 3 and st["assembled"] == 2 and st["nnz"] == 0
    _, st, rc = gpu_ctx.solve_batch(mesh3d, SIGMA3, SRC[:1], EVAL[:1], solver.make_opts(rtol=1e-6, op="csr"))
    assert rc == 0 and st["op_used"] == 0 and st["assembled"] == 1


@pytest.mark.gpu
def test_fp32_chebyshev_chain_inside_the_fp64_solve(mesh3d, gpu_ctx):
    """Above 32 k vertex rows the Chebyshev launches of an fp64 solve run in fp32 storage (remo_debug_tune key 15; forced here on the
    small mesh together with the compact vertex block and without the folded first step): an inexactly applied preconditioner
    changes the iteration slightly, not the solution."""
    from remo3d_amd import _lib, solver
    L = _lib.load()
    b = gpu_ctx.batch(mesh3d, SIGMA3, SRC, EVAL)
    try:
        res = {}
        L.remo_debug_tune(9, 0); L.remo_debug_tune(13, 2)
        for mode in (0, 2):
            L.remo_debug_tune(15, mode)
            assert b.run(solver.make_opts(rtol=1e-11, maxsteps=5000)) == 0
            res[mode] = (np.concatenate(b.fetch()), b.stats["pcg_steps"], b.true_relres())
        assert np.allclose(res[0][0], res[2][0], rtol=1e-8, atol=0)
        assert abs(res[0][1] - res[2][1]) <= max(3, res[0][1] // 20), (res[0][1], res[2][1])
        assert np.max(res[2][2]) < 5e-11
    finally:
        L.remo_debug_tune(9, 1); L.remo_debug_tune(13, 1); L.remo_debug_tune(15, 1)
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["2d", "3d"])
@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_multigrid_cycle_on_the_vertex_block(which, precision, mesh2d, mesh3d, gpu_ctx):
    """remo_opts_t.coarse: the smoothed-aggregation V(1,1) cycle (amg.hip) and the Chebyshev polynomial are two solvers of the
    same P1 block inside the same two-level preconditioner: potentials agree with the oracle to 1e-8 either way, the run reports
    which one it used, the default is the cycle in 2D and the polynomial in 3D, an explicit degree selects the polynomial under
    "auto", and "amg_or_chebyshev" takes the cycle wherever its hierarchy can be built."""
    from remo3d_amd import solver
    mesh = mesh2d if which == "2d" else mesh3d
    o, ref = _oracle_solve(mesh, SIGMA3, True)
    steps = {}
    for coarse, used in (("chebyshev", 1), ("amg", 2), ("auto", 2 if which == "2d" else 1), ("amg_or_chebyshev", 2)):
        outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-12, maxsteps=20000, precision=precision, coarse=coarse))
        assert rc == 0 and st["coarse_used"] == used
        steps[coarse] = max(st["iterations"][:3])
        for g, r in zip(outs, ref):
            assert np.max(np.abs(g - r)) <= 1e-8 * np.max(np.abs(r))
    print(which, precision, "PCG steps:", steps)
    outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-10, coarse_degree=8, coarse_ratio=100))
    assert rc == 0 and st["coarse_used"] == 1
    outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(preconditioner="local", rtol=1e-10, maxsteps=20000, coarse="amg"))
    assert rc == 0 and st["coarse_used"] == 0
    # "amg_or_chebyshev" (what Model asks for on its conforming 3D meshes) with an explicit degree: still the cycle where it can be built
    outs, st, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, solver.make_opts(rtol=1e-10, coarse="amg_or_chebyshev", coarse_degree=8, coarse_ratio=100))
    assert rc == 0 and st["coarse_used"] == 2
    o_bad = solver.make_opts(rtol=1e-10)
    o_bad.coarse = 7
    _, _, rc = gpu_ctx.solve_batch(mesh, SIGMA3, SRC, EVAL, o_bad, raise_on_error=False)
    assert rc == solver.REMO_ERR_ARG


@pytest.mark.gpu
def test_multigrid_cycle_with_chunks_of_eight_and_one(mesh2d, gpu_ctx):
    """Nine right-hand sides -> the cycle runs with 8 columns and with 1; every column equals its single-RHS solve."""
    from remo3d_amd import solver
    zs = np.linspace(-0.4, 0.4, 9)
    src = [([z], [1.0]) for z in zs]
    ev = [[z + 0.4, z + 6.4] for z in zs]
    opts = solver.make_opts(rtol=1e-12, maxsteps=20000, coarse="amg")
    outs, st, rc = gpu_ctx.solve_batch(mesh2d, SIGMA3, src, ev, opts)
    assert rc == 0 and st["coarse_used"] == 2
    for i in (0, 4, 7, 8):
        single, _, rc1 = gpu_ctx.solve_batch(mesh2d, SIGMA3, [src[i]], [ev[i]], opts)
        assert rc1 == 0
        assert np.allclose(outs[i], single[0], rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_multigrid_cycle_is_reproducible_and_cuts_the_steps_on_a_config2_batch(gpu_ctx, examples_dir):
    """One batch of BASELINE configs[1] (Benchmark model 1, 80 k vertices): the hierarchy is built without floating-point
    atomics, so two runs give bit-identical potentials; the cycle needs at most 0.75 x the PCG steps of the polynomial."""
    import os
    from remo3d_amd import geometry, solver, tasks
    from remo3d_amd.model import Model, default_mesh_provider
    ex = os.path.join(examples_dir, "Benchmark models", "Benchmark model 1")
    m = Model(["A0.4M6.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
    sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(5, 55, 100), 5)
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[5], sim[5], 50.0)
    mesh = default_mesh_provider()(2, 50.0, batches[5], fg, bh, 0.0)
    sources, evals, _ = tasks.batch_rhs(batches[5], m.tools)
    got = {}
    for coarse in ("chebyshev", "amg", "amg"):
        outs, st, rc = gpu_ctx.solve_batch(mesh, sigma, sources, evals, solver.make_opts(rtol=1e-10, coarse=coarse))
        assert rc == 0
        got.setdefault(coarse, []).append((np.concatenate(outs), st["pcg_steps"]))
    (uc, sc), = got["chebyshev"]
    (u1, s1), (u2, s2) = got["amg"]
    print("PCG steps: chebyshev", sc, "cycle", s1)
    assert np.array_equal(u1, u2) and s1 == s2
    assert np.max(np.abs(u1 - uc) / np.abs(uc)) < 1e-7
    assert s1 <= 0.75 * sc


@pytest.mark.gpu
def test_multigrid_cycle_is_a_symmetric_positive_convergent_operator(mesh2d, gpu_ctx):
    """One cycle C of the hierarchy, applied through the inspection hook: symmetric (r1' C r2 = r2' C r1 to rounding: the PCG needs
    a fixed symmetric preconditioner), positive, and a convergent iteration for the P1 block by itself - the energy norm of the
    error shrinks by at least a third per cycle on random vectors (the block comes from remo_batch_get_system); the fp32 image
    is the same operator to fp32 rounding."""
    import scipy.sparse as sp
    from remo3d_amd import solver
    b = gpu_ctx.batch(mesh2d, SIGMA3, SRC, EVAL)
    try:
        rc = b.run(solver.make_opts(rtol=1e-10, coarse="amg"))
        assert rc == 0 and b.stats["coarse_used"] == 2
        rowptr, col, val, dinv, freeid = b.system()
        n = len(rowptr) - 1
        A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
        rng = np.random.default_rng(5)
        # the block size: ask with a first call
        import ctypes as C
        nvc = C.c_int64(0)
        assert b._L.remo_batch_apply_coarse(b.ctx._h, b._h, 1, None, None, 0, C.byref(nvc)) == 0
        nv = nvc.value
        Avv = A[:nv, :nv].tocsr()
        R = rng.standard_normal((nv, 3))
        Z = b.apply_vertex_solver(R)
        G = R.T @ Z
        assert np.max(np.abs(G - G.T)) <= 1e-11 * np.max(np.abs(G))
        assert np.all(np.diag(G) > 0)
        E = rng.standard_normal((nv, 3))
        E1 = E - b.apply_vertex_solver(Avv @ E)
        before = np.sqrt(np.einsum("ik,ik->k", E, Avv @ E))
        after = np.sqrt(np.einsum("ik,ik->k", E1, Avv @ E1))
        print("energy-norm contraction of one cycle:", after / before)
        assert np.all(after < 0.67 * before)
        Z32 = b.apply_vertex_solver(R, fp32=True)
        assert np.max(np.abs(Z32 - Z)) <= 1e-4 * np.max(np.abs(Z))
        # three columns and one: the same operator column by column; more columns than the batch has right-hand sides are refused
        Z1 = b.apply_vertex_solver(R[:, 2:3])
        assert np.max(np.abs(Z[:, 2:3] - Z1)) <= 1e-13 * np.max(np.abs(Z1))
        with pytest.raises(solver.RemoError):
            b.apply_vertex_solver(rng.standard_normal((nv, 8)))
        # a run with the polynomial leaves no hierarchy behind
        rc = b.run(solver.make_opts(rtol=1e-10, coarse="chebyshev"))
        assert rc == 0
        with pytest.raises(solver.RemoError):
            b.apply_vertex_solver(R)
    finally:
        b.close()
