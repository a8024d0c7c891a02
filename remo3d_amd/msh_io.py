"""MSH 2.2 ASCII mesh ingress / egress: the file format through which the reference's Gmsh path hands
its mesh to the FEM solve (gmsh_functions.py:531-536 writes ./tmp/fm_<rank>.msh, ReadGmsh at
gmsh_functions.py:177-382 parses it into a Netgen mesh).  This module turns such a file into the
arrays of `remo_mesh_t` with the reference's numbering conventions:

  * material index of a volume element = order of FIRST APPEARANCE of its elementary tag in the
    $Elements section, starting at 0 here (ReadGmsh: `materialmap`, 1-based, lines 332-361);
    the sigma list of the reference is ordered the same way (gmsh_functions.py:172);
  * a boundary facet is Dirichlet iff the physical name of its group is "dirichlet_boundary"
    (gmsh_functions.py:518-519, 666-667; worker.py:90);
  * 2D models live in the (x, y) plane of the file with x = r and y = z (gmsh_functions.py:392-409).

Only first-order lines / triangles / tetrahedra (types 1, 2, 4) and points (15, ignored) occur in
the reference's meshes; anything else raises.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from .meshgen import Mesh

_NODES = {1: 2, 2: 3, 4: 4, 15: 1}
_DIM = {1: 1, 2: 2, 4: 3, 15: 0}


def morton_renumber(mesh: Mesh) -> Mesh:
    """The same mesh with its nodes renumbered along a Morton curve (21 bits per coordinate).  The device path numbers edge / face
    dofs by their vertices and cuts the (vertex-sorted) element list into patches, so its vectors and patches are compact when
    neighbouring vertices have neighbouring numbers; the in-repo meshers deliver that, a Gmsh file need not.  Potentials do not
    depend on the numbering."""
    pts = mesh.coords
    q = np.floor((pts - pts.min(0)) / (pts.max(0) - pts.min(0) + 1e-300) * ((1 << 21) - 1)).astype(np.uint64)
    key = np.zeros(len(pts), dtype=np.uint64)
    for bit in range(21):
        for k in range(mesh.dim):
            key |= ((q[:, k] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(bit * mesh.dim + k)
    order = np.argsort(key, kind="stable")
    new_id = np.empty(len(pts), dtype=np.int64)
    new_id[order] = np.arange(len(pts))
    meta = dict(mesh.meta)
    for name in ("node_h", "node_iface"):       # per-node arrays of the in-repo meshers follow the nodes
        if name in meta and hasattr(meta[name], "__len__") and len(meta[name]) == len(pts):
            meta[name] = np.asarray(meta[name])[order]
    return Mesh(mesh.dim, np.ascontiguousarray(pts[order]), np.ascontiguousarray(new_id[mesh.conn].astype(np.int32)), mesh.mat,
                np.ascontiguousarray(new_id[mesh.bconn].astype(np.int32)) if len(mesh.bconn) else mesh.bconn, mesh.bdirichlet, meta)


def read_msh(path: str, dim: int, dirichlet_name: str = "dirichlet_boundary", renumber: bool = False) -> Mesh:
    """renumber = True: nodes in Morton order (morton_renumber) instead of file order - what a mesh on its way to the GPU wants."""
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    names: Dict[int, str] = {}
    node_id, coords = [], []
    vol_conn, vol_tag, bnd_conn, bnd_phys = [], [], [], []
    while i < len(lines):
        head = lines[i].strip()
        if head == "$MeshFormat":
            version = lines[i + 1].split()
            if not version or not version[0].startswith("2."):
                raise ValueError("only MSH 2.x ASCII files are supported")
            if len(version) > 1 and version[1] != "0":
                raise ValueError("binary MSH files are not supported")
            i += 2
        elif head == "$PhysicalNames":
            n = int(lines[i + 1])
            for k in range(n):
                parts = lines[i + 2 + k].split(None, 2)
                names[int(parts[1])] = parts[2].strip().strip('"')
            i += 2 + n
        elif head == "$Nodes":
            n = int(lines[i + 1].split()[0])
            for k in range(n):
                p = lines[i + 2 + k].split()
                node_id.append(int(p[0])); coords.append((float(p[1]), float(p[2]), float(p[3])))
            i += 2 + n
        elif head == "$Elements":
            n = int(lines[i + 1].split()[0])
            for k in range(n):
                p = lines[i + 2 + k].split()
                etype, ntags = int(p[1]), int(p[2])
                if etype not in _NODES:
                    raise ValueError(f"element type {etype} not supported (first-order simplices only)")
                tags = [int(t) for t in p[3:3 + ntags]]
                nodes = [int(t) for t in p[3 + ntags:3 + ntags + _NODES[etype]]]
                d = _DIM[etype]
                if d == dim:
                    vol_conn.append(nodes); vol_tag.append(tags[1] if ntags > 1 else 0)
                elif d == dim - 1:
                    bnd_conn.append(nodes); bnd_phys.append(tags[0] if ntags > 0 else 0)
            i += 2 + n
        else:
            i += 1
    if not coords or not vol_conn:
        raise ValueError("no nodes or no volume elements of dimension %d in %s" % (dim, path))
    node_id = np.asarray(node_id)
    lookup = {int(g): k for k, g in enumerate(node_id)}
    remap = np.vectorize(lookup.__getitem__)
    conn = remap(np.asarray(vol_conn)).astype(np.int32)
    first: Dict[int, int] = {}
    mat = np.empty(len(vol_tag), dtype=np.int32)
    for k, t in enumerate(vol_tag):
        mat[k] = first.setdefault(t, len(first))
    if bnd_conn:
        bconn = remap(np.asarray(bnd_conn)).astype(np.int32)
        bdir = np.array([names.get(p) == dirichlet_name for p in bnd_phys], dtype=np.uint8)
    else:
        bconn = np.zeros((0, dim), dtype=np.int32); bdir = np.zeros(0, dtype=np.uint8)
    xyz = np.asarray(coords, dtype=np.float64)[:, :dim]
    mesh = Mesh(dim, np.ascontiguousarray(xyz), np.ascontiguousarray(conn), mat, np.ascontiguousarray(bconn), bdir,
                dict(physical_names=names, elementary_tags=list(first.keys())))
    return morton_renumber(mesh) if renumber else mesh


def write_msh(path: str, mesh: Mesh, elementary_tags: Optional[Sequence[int]] = None) -> None:
    """Write `mesh` the way the reference's Gmsh models do: physical groups "dirichlet_boundary" (1)
    and "neumann_boundary" (2) on the boundary, one physical volume/surface per material
    (gmsh_functions.py:518-528, 660-670).  elementary_tags[m] = entity tag of material m (default the
    reference's numbering, 4.. in 2D and 3.. in 3D)."""
    dim = mesh.dim
    nmat = int(mesh.mat.max()) + 1
    base = 4 if dim == 2 else 3
    tags = list(elementary_tags) if elementary_tags is not None else [base + m for m in range(nmat)]
    vname = "surf_" if dim == 2 else "vol_"
    btype, vtype = (1, 2) if dim == 2 else (2, 4)
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$PhysicalNames\n%d\n" % (2 + nmat))
        f.write('%d 1 "dirichlet_boundary"\n%d 2 "neumann_boundary"\n' % (dim - 1, dim - 1))
        for m in range(nmat):
            f.write('%d %d "%s%d"\n' % (dim, tags[m], vname, m + 1))
        f.write("$EndPhysicalNames\n$Nodes\n%d\n" % mesh.n_nodes)
        for k, p in enumerate(mesh.coords):
            x, y = p[0], p[1]
            z = p[2] if dim == 3 else 0.0
            f.write("%d %.17g %.17g %.17g\n" % (k + 1, x, y, z))
        f.write("$EndNodes\n$Elements\n%d\n" % (len(mesh.bconn) + mesh.n_elems))
        eid = 1
        for fac, dflag in zip(mesh.bconn, mesh.bdirichlet):
            phys = 1 if dflag else 2
            f.write("%d %d 2 %d %d %s\n" % (eid, btype, phys, 100 + phys, " ".join(str(int(v) + 1) for v in fac)))
            eid += 1
        order = np.argsort(np.asarray(tags)[mesh.mat], kind="stable")     # Gmsh writes entity by entity
        for t in order:
            m = int(mesh.mat[t])
            f.write("%d %d 2 %d %d %s\n" % (eid, vtype, tags[m], tags[m], " ".join(str(int(v) + 1) for v in mesh.conn[t])))
            eid += 1
        f.write("$EndElements\n")
