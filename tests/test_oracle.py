"""Pins of the CPU oracle itself (it has no NGSolve golden vectors to lean on: "parity unpinned"):
closed-form physics and internal consistency."""
import numpy as np
import pytest

from conftest import SIGMA3


@pytest.mark.parametrize("dim,scale,tol", [(2, 1.0, 2e-4), (3, 6.0, 5e-3)])
def test_homogeneous_grounded_sphere(dim, scale, tol):
    """u = I/(4 pi sigma) (1/r - 1/R) for a source at the centre of a grounded sphere; and the
    identity Ra == R of the geometric factors (remo3d.py:285-306)."""
    from oracle.fem_oracle import Oracle
    from remo3d_amd.meshgen import make_mesh
    R, sigma = 50.0, 0.1
    m = make_mesh(dim, R, [0.0], scale=scale, snap_z=[0.4, 6.4, 2.0, 2.5])
    o = Oracle(m, [sigma])
    f, se, sf = o.rhs([0.0], [1.0])
    u, it, rr, rc = o.pcg(f, 1e-12, 50000)
    assert rc == 0
    z = np.array([0.4, 6.4, 2.0, 2.5])
    uh = o.eval(u, z, (se, sf)) / (1.0 if dim == 2 else 2.0)      # half-space model, worker.py:129
    exact = 1.0 / (4 * np.pi * sigma) * (1 / z - 1 / R)
    assert np.max(np.abs(uh - exact) / exact) < 5e-3                # polyhedral outer boundary: constant offset
    ra_n = abs(4 * np.pi * 0.4 * 6.4 / 6.0 * (uh[1] - uh[0]))
    ra_l = abs(4 * np.pi * 2.0 * 2.5 / 0.5 * (uh[3] - uh[2]))
    assert abs(ra_n - 10) < 10 * tol and abs(ra_l - 10) < 10 * tol


def test_condensed_and_full_2d_systems_agree(mesh2d):
    """condense=True eliminates the cell bubbles exactly (ngsolve_functions.py:31, 53-56)."""
    from oracle.fem_oracle import Oracle
    outs = []
    for cond in (True, False):
        o = Oracle(mesh2d, SIGMA3, condense=cond)
        f, se, sf = o.rhs([0.0, 0.3], [1.0, -1.0])
        u, it, rr, rc = o.pcg(f, 1e-13, 50000)
        outs.append(o.eval(u, [0.4, 1.0, 6.4, -3.0], (se, sf)))
    assert np.max(np.abs(outs[0] - outs[1])) < 1e-9 * np.max(np.abs(outs[0]))


def test_matrix_is_symmetric_positive_with_constant_nullspace_removed(mesh3d):
    import scipy.sparse as sp
    from oracle.fem_oracle import Oracle
    o = Oracle(mesh3d, SIGMA3)
    rp, col, val = o.csr()
    A = sp.csr_matrix((val, col, rp), shape=(o.nfree, o.nfree))
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()
    assert A.diagonal().min() > 0
    x = np.random.default_rng(1).standard_normal(o.nfree)
    assert x @ (A @ x) > 0
    assert np.allclose(o.spmv(x), A @ x, rtol=1e-13, atol=1e-13 * np.abs(A @ x).max())


def test_linearity_and_zero_strength_sources(mesh2d):
    from oracle.fem_oracle import Oracle
    o = Oracle(mesh2d, SIGMA3)
    fa, *_ = o.rhs([0.0], [1.0]); fb, *_ = o.rhs([0.1], [1.0]); fc, *_ = o.rhs([0.0, 0.1], [2.0, -3.0])
    assert np.allclose(fc, 2 * fa - 3 * fb, rtol=0, atol=1e-15)
    fz, *_ = o.rhs([0.0, 0.1], [0.0, 0.0])          # zero strengths are skipped (ngsolve_functions.py:43)
    assert not fz.any()
    with pytest.raises(RuntimeError):
        o.rhs([80.0], [1.0])                        # outside the domain


def test_solve_batch_wrapper_matches_object_api(mesh2d):
    from oracle.fem_oracle import Oracle, solve_batch
    out, rc, st = solve_batch(mesh2d, SIGMA3, [0, 1, 3], [0.0, -0.1, 0.1], [1.0, 1.0, -1.0], [0, 2, 3], [0.4, 6.4, 2.0], rtol=1e-12)
    assert rc == 0 and st["n"] > 0
    o = Oracle(mesh2d, SIGMA3)
    f, se, sf = o.rhs([0.0], [1.0]); u, *_ = o.pcg(f, 1e-12)
    assert np.allclose(out[:2], o.eval(u, [0.4, 6.4], (se, sf)), rtol=1e-9)


def _two_half_spaces(scale):
    """Point source at the origin of medium 1 (z < h), plane interface z = h to medium 2: image solution
    u1 = I/(4 pi s1) (1/|z| + k/|2h - z|), u2 = I/(2 pi (s1 + s2) |z|), k = (s1 - s2)/(s1 + s2)."""
    from remo3d_amd.meshgen import make_mesh
    R, h, s1, s2 = 50.0, 1.5, 0.2, 0.02
    iface = [np.array([[0.0, h], [np.sqrt(R * R - h * h), h]])]
    zs = np.array([0.4, 1.0, -0.7, -3.0, 2.0, 4.0, 6.4])
    mesh = make_mesh(2, R, [0.0], scale=scale, snap_z=list(zs), interfaces=iface,
                     material_fn=lambda c: (c[:, 1] > h).astype(np.int32))
    k = (s1 - s2) / (s1 + s2)
    exact = np.where(zs < h, (1.0 / np.abs(zs) + k / np.abs(2 * h - zs)) / (4 * np.pi * s1), 1.0 / (2 * np.pi * (s1 + s2) * np.abs(zs)))
    return mesh, [s1, s2], zs, exact


def test_two_half_spaces_image_solution():
    """Heterogeneous analytic pin: the oracle reproduces the image solution of a point source next
    to a plane interface.  Differences of potentials are compared: the grounded sphere at R = 50 m shifts all of
    them by almost the same constant."""
    from oracle.fem_oracle import Oracle
    mesh, sigma, zs, exact = _two_half_spaces(2.0)
    o = Oracle(mesh, sigma)
    f, se, sf = o.rhs([0.0], [1.0])
    u, it, rr, rc = o.pcg(f, 1e-11, 50000)
    assert rc == 0
    got = o.eval(u, zs, (se, sf))
    d_got, d_ex = got[:-1] - got[1:], exact[:-1] - exact[1:]
    assert np.max(np.abs(d_got - d_ex) / np.abs(d_ex)) < 5e-3, (d_got, d_ex)
