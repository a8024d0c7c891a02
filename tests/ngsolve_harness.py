"""The build's OWN NGSolve harness (SURVEY.md 8d "Baseline A"): the calls of remo3d/ngsolve_functions.py:27-56 and
remo3d/workers/worker.py:100-131 issued on mesh ARRAYS fed through netgen.meshing, so that the HIP path and NGSolve
solve the same mesh.  Test infrastructure: imported only by tests/test_gpu_configs.py, and only when `import ngsolve`
works on the box (it does not in the build container; the reference's own files never travel)."""
import numpy as np


def available():
    try:
        import ngsolve  # noqa: F401
        import netgen.meshing  # noqa: F401
        return True, getattr(ngsolve, "__version__", "?")
    except Exception as ex:                      # ModuleNotFoundError in this image
        return False, "%s: %s" % (type(ex).__name__, ex)


def to_ngsolve_mesh(mesh):
    """meshgen.Mesh arrays -> ngsolve.Mesh: material m of the arrays becomes domain m + 1 (sigma list order),
    Dirichlet facets get the boundary name the reference uses."""
    import netgen.meshing as msh
    import ngsolve as ngs
    dim = mesh.dim
    m = msh.Mesh(dim=dim)
    pts = [m.Add(msh.MeshPoint(msh.Pnt(float(p[0]), float(p[1]), float(p[2]) if dim == 3 else 0.0))) for p in mesh.coords]
    nmat = int(mesh.mat.max()) + 1
    for k in range(nmat):
        m.SetMaterial(k + 1, "mat%d" % (k + 1))
    names = {1: "dirichlet_boundary", 2: "neumann_boundary"}
    for bc, name in names.items():
        fd = msh.FaceDescriptor(bc=bc)
        fd.bcname = name
        m.SetBCName(bc - 1, name)
        m.Add(fd)
    X = mesh.coords
    for t, c in enumerate(mesh.conn):
        c = [int(v) for v in c]
        if dim == 3:
            e = X[c[1:]] - X[c[0]]
            if np.linalg.det(e) < 0:             # Gmsh orientation, what ReadGmsh hands to Netgen unchanged
                c[2], c[3] = c[3], c[2]
            m.Add(msh.Element3D(int(mesh.mat[t]) + 1, [pts[v] for v in c]))
        else:
            e = X[c[1:]] - X[c[0]]
            if e[0, 0] * e[1, 1] - e[0, 1] * e[1, 0] < 0:
                c[1], c[2] = c[2], c[1]
            m.Add(msh.Element2D(int(mesh.mat[t]) + 1, [pts[v] for v in c]))
    for f, d in zip(mesh.bconn, mesh.bdirichlet):
        idx = 1 if d else 2
        if dim == 3:
            m.Add(msh.Element2D(idx, [pts[int(v)] for v in f]))
        else:
            m.Add(msh.Element1D([pts[int(v)] for v in f], index=idx))
    return ngs.Mesh(m)


def solve(mesh, sigma, source_z, source_I, eval_z, preconditioner="local", condense=True, rtol=1e-12):
    """One right-hand side: u_h at eval_z on the axis.  Same calls as the reference's SolveBVP."""
    import ngsolve as ngs
    dim = mesh.dim
    nm = to_ngsolve_mesh(mesh)
    sig = ngs.CoefficientFunction([float(s) for s in sigma])
    fes = ngs.H1(nm, order=3, dirichlet="dirichlet_boundary")
    u, v = fes.TnT()
    a = ngs.BilinearForm(fes, symmetric=False, condense=condense)
    a += (2 * np.pi * ngs.grad(u) * ngs.grad(v) * ngs.x * sig * ngs.dx) if dim == 2 else (ngs.grad(u) * ngs.grad(v) * sig * ngs.dx)
    f = ngs.LinearForm(fes)
    f.Assemble()
    for z, I in zip(source_z, source_I):
        if I == 0:
            continue
        mp = nm(0.0, float(z)) if dim == 2 else nm(0.0, 0.0, float(z))
        ei = ngs.ElementId(ngs.VOL, mp.nr)
        shape = fes.GetFE(ei).CalcShape(mp.pnt[0], mp.pnt[1], mp.pnt[2])
        for d, s in zip(fes.GetDofNrs(ei), shape):
            if d >= 0:
                f.vec[d] += I * s
    c = ngs.Preconditioner(a, preconditioner)
    a.Assemble()
    gfu = ngs.GridFunction(fes)
    inv = ngs.CGSolver(a.mat, c.mat, maxsteps=20000, precision=rtol)
    gfu.vec.data = inv * f.vec
    if condense:
        f.vec.data += a.harmonic_extension_trans * f.vec
        gfu.vec.data += a.harmonic_extension * gfu.vec
        gfu.vec.data += a.inner_solve * f.vec
    return np.array([gfu(nm(0.0, float(z))) if dim == 2 else gfu(nm(0.0, 0.0, float(z))) for z in eval_z]), fes.ndof
