"""Seeded synthetic mesh makers for the ReMo3D hot path (no Gmsh / Netgen in the image).

The reference obtains its meshes from Gmsh (gmsh_functions.py:384-684) or Netgen
(netgen_functions.py:120-335); both are third-party and absent here, and meshing is outside the
hot path ("mesh generation excluded").  These makers produce the *input* of
the hot path - straight-sided triangle / tetrahedron meshes of the reference's domain:

* 2D: half disc  {(r, z): r >= 0, r^2 + z^2 <= R^2}           (gmsh_functions.py:392-449)
* 3D: half ball  {(x, y, z): y >= 0, |x| <= R}                 (gmsh_functions.py:581, angle3 = pi)

graded with the reference's background size field (gmsh_functions.py:487-500, 630-643)

    h(x) = min(rho + 0.1,  min_s (d_s^2 / 2 + 0.01)),

rho = distance from the borehole axis, d_s = distance from current electrode s.  (In 3D the
reference's field string uses ``y`` where ``z`` is meant, gmsh_functions.py:637 / SURVEY.md
section 7.3-5; the intended, 2D-consistent form is used here and stated in DESIGN.md.)

Method: a graded quadtree/octree is refined until every leaf is smaller than h on it; leaf
corners and leaf centres (a body-centred lattice, whose Delaunay triangulation is well shaped)
are triangulated with scipy.spatial.Delaunay (the domains are convex).  Points near the outer
sphere are projected onto it; electrode positions are snapped onto axis vertices.  Material
numbers are assigned by element centroid through a caller-supplied function, so interfaces are
resolved to within one (locally refined) element - the meshes are benchmark inputs, not a
replacement for a CAD mesher.

Outputs are plain arrays in exactly the layout of ``remo_mesh_t`` (include/remo3d_hip.h).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np


@dataclass
class Mesh:
    dim: int
    coords: np.ndarray      # [n_nodes, dim] float64
    conn: np.ndarray        # [n_elems, dim+1] int32
    mat: np.ndarray         # [n_elems] int32 (0-based index into sigma)
    bconn: np.ndarray       # [n_bfacets, dim] int32
    bdirichlet: np.ndarray  # [n_bfacets] uint8
    meta: dict = field(default_factory=dict)

    @property
    def n_nodes(self) -> int:
        return int(self.coords.shape[0])

    @property
    def n_elems(self) -> int:
        return int(self.conn.shape[0])


def size_field(pts: np.ndarray, dim: int, sources_z: Sequence[float], scale: float = 1.0,
               h_axis: float = 0.1, h_src: float = 0.01) -> np.ndarray:
    """Reference background mesh size at points (gmsh_functions.py:487-500 / 630-643)."""
    if dim == 2:
        rho = np.abs(pts[:, 0]); z = pts[:, 1]
    else:
        rho = np.hypot(pts[:, 0], pts[:, 1]); z = pts[:, 2]
    h = rho + h_axis
    for zs in sources_z:
        h = np.minimum(h, 0.5 * (rho * rho + (z - zs) ** 2) + h_src)
    return scale * h


class LayerCap:
    """Upper bound on the mesh size from the thickness of horizontal layers: h <= factor * thickness
    of every layer a cell / point touches (z intervals `edges[k] .. edges[k+1]`).  Without it the
    size field ignores thin beds, and interface-conforming sampling (make_mesh(interfaces=...))
    cannot keep boundaries that are closer to each other than the local element size."""

    def __init__(self, edges, factor: float = 1.5):
        self.edges = np.asarray(edges, dtype=np.float64)
        cap = factor * np.diff(self.edges)
        self.cap = np.concatenate([[np.inf], cap, [np.inf]])       # outside the table: no bound
        # sparse table for range minima
        self.tab = [self.cap]
        j = 1
        while (1 << j) <= len(self.cap):
            prev = self.tab[-1]
            self.tab.append(np.minimum(prev[:len(prev) - (1 << (j - 1))], prev[(1 << (j - 1)):]))
            j += 1

    def interval_min(self, zl, zh):
        lo = np.searchsorted(self.edges, zl, side="left")          # intervals lo..hi (inclusive) overlap or touch [zl, zh]
        hi = np.searchsorted(self.edges, zh, side="right")
        hi = np.maximum(hi, lo)
        span = hi - lo + 1
        j = np.floor(np.log2(span)).astype(int)
        out = np.empty(len(lo))
        for jj in np.unique(j):
            m = j == jj
            t = self.tab[jj]
            out[m] = np.minimum(t[lo[m]], t[hi[m] - (1 << jj) + 1])
        return out

    def at(self, z: float) -> float:
        # on a boundary both neighbours count
        return float(min(self.cap[int(np.searchsorted(self.edges, z, side="right"))], self.cap[int(np.searchsorted(self.edges, z, side="left"))]))


def _refine_tree(dim: int, R: float, sources_z, scale, h_axis, h_src, max_level: int, h_max: float, layer_cap=None):
    """Breadth-first graded quadtree / octree.  Returns leaves as (integer origin [m, dim], level[m])."""
    # root cells have side R; axis and symmetry plane lie on cell boundaries at every level.
    if dim == 2:
        roots = np.array([[0, -1], [0, 0]], dtype=np.int64)
    else:
        roots = np.array([[ix, 0, iz] for ix in (-1, 0) for iz in (-1, 0)], dtype=np.int64)
    src = np.asarray(list(sources_z), dtype=np.float64)
    leaves_o, leaves_l = [], []
    cells = roots
    for level in range(max_level + 1):
        if cells.shape[0] == 0:
            break
        s = R / (1 << level)
        lo = cells * s
        hi = lo + s
        # smallest value of the size field on the closed cell (h grows with both distances), times
        # 1.5: a body-centred cell of side s has edges s and 0.87 s and sides come in powers of two
        if dim == 2:
            rho_min = np.maximum(lo[:, 0], 0.0)
            zl, zh = lo[:, 1], hi[:, 1]
        else:
            dx = np.maximum(np.maximum(lo[:, 0], -hi[:, 0]), 0.0)
            dy = np.maximum(lo[:, 1], 0.0)
            rho_min = np.hypot(dx, dy)
            zl, zh = lo[:, 2], hi[:, 2]
        hmin = rho_min + h_axis
        for zs in src:
            dz = np.maximum(np.maximum(zl - zs, zs - zh), 0.0)
            hmin = np.minimum(hmin, 0.5 * (rho_min ** 2 + dz ** 2) + h_src)
        hmin = np.minimum(scale * hmin, h_max)
        if layer_cap is not None:
            hmin = np.minimum(hmin, layer_cap.interval_min(zl, zh))
        hmin = 1.5 * hmin
        # cells completely outside the ball are dropped
        near = np.where(np.abs(lo) < np.abs(hi), lo, hi)
        near = np.where((lo <= 0) & (hi >= 0), 0.0, near)
        outside = np.sqrt((near ** 2).sum(1)) > R
        split = (s > hmin) & (~outside) & (level < max_level)
        keep = (~split) & (~outside)
        leaves_o.append(cells[keep]); leaves_l.append(np.full(int(keep.sum()), level, dtype=np.int64))
        par = cells[split]
        if par.shape[0] == 0:
            cells = par
            continue
        offs = np.array(np.meshgrid(*[[0, 1]] * dim, indexing="ij")).reshape(dim, -1).T
        cells = (par[:, None, :] * 2 + offs[None, :, :]).reshape(-1, dim)
    return np.concatenate(leaves_o), np.concatenate(leaves_l)


def _lattice_points(dim, R, origins, levels, max_level):
    """Leaf corners + centres on an integer grid of resolution R / 2^(max_level+1); deduplicated.
    Returns (points float64, local cell size per point, is_centre)."""
    shift = (max_level + 1) - levels                     # cell side in grid units = 2^shift
    side = (1 << shift).astype(np.int64)
    base = origins * side[:, None]
    offs = np.array(np.meshgrid(*[[0, 1]] * dim, indexing="ij")).reshape(dim, -1).T
    corners = (base[:, None, :] + offs[None, :, :] * side[:, None, None]).reshape(-1, dim)
    csize = np.repeat(side, offs.shape[0])
    centres = base + (side // 2)[:, None]
    allp = np.concatenate([corners, centres])
    alls = np.concatenate([csize, side])
    isc = np.concatenate([np.zeros(corners.shape[0], bool), np.ones(centres.shape[0], bool)])
    # dedupe keeping the smallest local size
    order = np.lexsort((alls,) + tuple(allp[:, k] for k in range(dim - 1, -1, -1)))
    allp, alls, isc = allp[order], alls[order], isc[order]
    first = np.ones(allp.shape[0], bool)
    first[1:] = np.any(allp[1:] != allp[:-1], axis=1)
    unit = R / (1 << (max_level + 1))
    return allp[first].astype(np.float64) * unit, alls[first].astype(np.float64) * unit, isc[first]


def _boundary_facets(conn: np.ndarray):
    """Facets that belong to exactly one element, vertices ascending, facets in lexicographic order."""
    nb = conn.shape[1]
    if conn.size and int(conn.max()) < (1 << 21):
        # the sorted vertex tuple of a facet as ONE 63-bit key: sorting the keys is the lexicographic sort, the facets that occur once are
        # read back out of their keys - no gather through a permutation, no row-wise sort (3-5 x faster than the lexsort form below, same result)
        c = [conn[:, k].astype(np.int64) for k in range(nb)]
        keys = []
        for k in range(nb):
            v = [c[j] for j in range(nb) if j != k]
            if nb == 3:
                lo, hi = np.minimum(v[0], v[1]), np.maximum(v[0], v[1])
                keys.append((lo << 21) | hi)
            else:
                lo, hi = np.minimum(v[0], v[1]), np.maximum(v[0], v[1])
                mid = np.maximum(lo, np.minimum(hi, v[2]))
                lo, hi = np.minimum(lo, v[2]), np.maximum(hi, v[2])
                keys.append((lo << 42) | (mid << 21) | hi)
        ks = np.sort(np.concatenate(keys))
        same_next = np.zeros(ks.shape[0], bool)
        same_next[:-1] = ks[1:] == ks[:-1]
        same_prev = np.zeros(ks.shape[0], bool)
        same_prev[1:] = same_next[:-1]
        bk = ks[~(same_next | same_prev)]
        mask = (1 << 21) - 1
        cols = [(bk >> (21 * (nb - 2 - k))) & mask for k in range(nb - 1)]
        return np.stack(cols, axis=1).astype(conn.dtype)
    faces = []
    for k in range(nb):
        idx = [j for j in range(nb) if j != k]
        faces.append(conn[:, idx])
    faces = np.sort(np.concatenate(faces), axis=1)
    order = np.lexsort(tuple(faces[:, k] for k in range(faces.shape[1] - 1, -1, -1)))
    fs = faces[order]
    same_next = np.zeros(fs.shape[0], bool)
    same_next[:-1] = np.all(fs[1:] == fs[:-1], axis=1)
    same_prev = np.zeros(fs.shape[0], bool)
    same_prev[1:] = same_next[:-1]
    return fs[~(same_next | same_prev)]


def make_mesh(dim: int, R: float = 50.0, sources_z: Sequence[float] = (0.0,), scale: float = 1.0,
              material_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None, seed: int = 0,
              h_axis: float = 0.1, h_src: float = 0.01, max_level: int = 18,
              snap_z: Sequence[float] = (), h_max: Optional[float] = None, jitter: float = 0.12,
              improve_passes: int = 6, interfaces: Sequence[np.ndarray] = (), layer_cap: Optional["LayerCap"] = None) -> Mesh:
    """Graded Delaunay mesh of the reference's half disc (dim=2) or half ball (dim=3).

    sources_z : axis positions of current electrodes (refinement centres, snapped to vertices)
    scale     : global multiplier on the size field (scale < 1 -> finer mesh)
    material_fn(centroids[m, dim]) -> int array of material numbers (0-based); default all 0
    snap_z    : further axis positions to snap onto vertices (e.g. measuring electrodes)
    h_max     : cap on the size field (default R/5; keeps the polyhedral outer boundary round)
    jitter    : seeded displacement of interior lattice points, fraction of the local cell size
    improve_passes : sliver-removal passes (3D): perturb vertices of elements with quality < 0.15, re-triangulate
    layer_cap  : LayerCap bounding the size by the thickness of the layers (thin beds)
    interfaces : (2D only) polylines [(r, z), ...] that must be unions of mesh edges (material
                 interfaces).  They are sampled at ~0.6 h and lattice points closer than 0.8 of that
                 spacing are removed, which makes every sub-segment a Gabriel - hence Delaunay - edge.
    """
    from scipy.spatial import Delaunay, cKDTree

    rng = np.random.default_rng(seed)
    if h_max is None:
        h_max = 0.2 * R
    origins, levels = _refine_tree(dim, R, sources_z, scale, h_axis, h_src, max_level, h_max, layer_cap)
    pts, hs, isc = _lattice_points(dim, R, origins, levels, max_level)

    rad = np.sqrt((pts ** 2).sum(1))
    on_axis = (pts[:, 0] == 0.0) if dim == 2 else ((pts[:, 0] == 0.0) & (pts[:, 1] == 0.0))
    # points near / beyond the outer sphere: corners are projected onto it, centres dropped
    band = 0.5 * hs
    proj = (~isc) & (rad > R - band) & (rad < R + band) & (rad > 0)
    drop = (rad >= R + band) | (isc & (rad > R - 0.6 * hs)) | ((~isc) & (~proj) & (rad > R - band))
    pts[proj] *= (R / rad[proj])[:, None]
    keep = ~drop
    pts, hs, isc, proj, on_axis = pts[keep], hs[keep], isc[keep], proj[keep], on_axis[keep]
    # exact poles / rim
    if dim == 2:
        extra = np.array([[0.0, -R], [0.0, R], [R, 0.0]])
    else:
        extra = np.array([[0.0, 0.0, -R], [0.0, 0.0, R], [R, 0.0, 0.0], [-R, 0.0, 0.0], [0.0, R, 0.0]])
    pts = np.concatenate([pts, extra]); hs = np.concatenate([hs, np.full(len(extra), hs.max())])
    proj = np.concatenate([proj, np.ones(len(extra), bool)])
    on_axis = np.concatenate([on_axis, np.zeros(len(extra), bool)])
    # merge near-duplicates created by the projection
    tree = cKDTree(pts)
    dd, ii = tree.query(pts, k=2)
    dup = (dd[:, 1] < 0.3 * np.minimum(hs, hs[ii[:, 1]])) & (ii[:, 1] < np.arange(len(pts))) & proj
    pts, hs, proj, on_axis = pts[~dup], hs[~dup], proj[~dup], on_axis[~dup]

    # conforming interfaces (2D): sample the polylines, clear a corridor around them
    n_iface = 0
    if dim == 2 and len(interfaces):
        ip, isp, isv = [], [], []          # points, local spacing, 1 for polyline vertices / junctions
        src_list = [float(v) for v in sources_z]
        import math

        def h_at(r_, z_):                    # scalar copy of size_field (these loops are hot: pure Python floats)
            rho = abs(r_)
            hq = rho + h_axis
            for zs in src_list:
                hq = min(hq, 0.5 * (rho * rho + (z_ - zs) ** 2) + h_src)
            hq = min(scale * hq, h_max)
            if layer_cap is not None:
                hq = min(hq, layer_cap.at(z_))
            return 0.6 * hq

        for poly in interfaces:
            pl = [(float(p_[0]), float(p_[1])) for p_ in np.asarray(poly, dtype=np.float64)]
            for (ar, az), (br, bz) in zip(pl[:-1], pl[1:]):
                ra, rb_ = math.hypot(ar, az), math.hypot(br, bz)
                if ra >= R and rb_ >= R:
                    continue
                if ra > R or rb_ > R:   # clip the segment at the outer circle
                    (pr_, pz_), (qr_, qz_) = ((ar, az), (br, bz)) if ra < R else ((br, bz), (ar, az))
                    dr_, dz_ = qr_ - pr_, qz_ - pz_
                    A_, B_, C_ = dr_ * dr_ + dz_ * dz_, 2 * (pr_ * dr_ + pz_ * dz_), pr_ * pr_ + pz_ * pz_ - R * R
                    t = (-B_ + math.sqrt(B_ * B_ - 4 * A_ * C_)) / (2 * A_)
                    hr_, hz_ = pr_ + t * dr_, pz_ + t * dz_
                    sc_ = R / math.hypot(hr_, hz_)
                    (ar, az), (br, bz) = (pr_, pz_), (hr_ * sc_, hz_ * sc_)
                L = math.hypot(br - ar, bz - az)
                if L == 0:
                    continue
                pos = 0.0
                hq = L
                while pos < L:
                    qr_, qz_ = ar + (pos / L) * (br - ar), az + (pos / L) * (bz - az)
                    hq = h_at(qr_, qz_)
                    ip.append((qr_, qz_)); isp.append(min(hq, L)); isv.append(1 if pos == 0.0 else 0)
                    pos += hq
                    if L - pos < 0.5 * hq:      # avoid a short last piece: the end point closes the segment
                        break
                ip.append((br, bz)); isp.append(min(hq, L)); isv.append(1)
        ip = np.array(ip, dtype=np.float64); isp = np.array(isp); isv = np.array(isv)
        # merge coincident / very close interface points (junctions are listed by both polylines):
        # of two points closer than 0.3 of their spacing a marching point yields to a polyline vertex,
        # otherwise the later one goes
        ti = cKDTree(ip)
        dd2, ii2 = ti.query(ip, k=2)
        nn = ii2[:, 1]
        close = dd2[:, 1] < 0.3 * np.minimum(isp, isp[nn])
        me = np.arange(len(ip))
        loses = (isv < isv[nn]) | ((isv == isv[nn]) & (me > nn))
        keep_i = ~(close & loses)
        ip, isp = ip[keep_i], isp[keep_i]
        ti = cKDTree(ip)
        dnn, inn = ti.query(pts)
        clear = (dnn < 0.8 * isp[inn]) & (~on_axis)
        pts, hs, proj, on_axis = pts[~clear], hs[~clear], proj[~clear], on_axis[~clear]
        n_iface = len(ip)
        iface_on_rim = np.hypot(ip[:, 0], ip[:, 1]) >= R * (1 - 1e-12)
        pts = np.concatenate([pts, ip]); hs = np.concatenate([hs, isp])
        proj = np.concatenate([proj, np.ones(n_iface, bool)])          # interface points are pinned like boundary points
        on_axis = np.concatenate([on_axis, ip[:, 0] == 0.0])

    # Coincident points: lattice points pulled onto the sphere can land on the same spot (the poles and the equator points of the
    # half ball: two lattice points on one ray).  Qhull takes minutes over an exact duplicate that happens to be a hull point,
    # seconds otherwise - a sweep of 40 batches met three such meshes.  Of a coincident pair the later point goes, unless it is
    # an interface point (those come last and are pinned).
    pairs = cKDTree(pts).query_pairs(1e-9 * R, output_type="ndarray")
    if len(pairs):
        first_iface = len(pts) - n_iface
        lose = np.where(pairs[:, 1] >= first_iface, pairs[:, 0], pairs[:, 1])
        lose = lose[lose < first_iface]
        keep_p = np.ones(len(pts), bool)
        keep_p[lose] = False
        pts, hs, proj, on_axis = pts[keep_p], hs[keep_p], proj[keep_p], on_axis[keep_p]
    is_iface = np.zeros(len(pts), bool)
    if n_iface:
        is_iface[len(pts) - n_iface:] = True
    # jitter strictly interior, off-plane points to break lattice degeneracies (seeded)
    on_plane = np.zeros(len(pts), bool) if dim == 2 else (pts[:, 1] == 0.0)
    if dim == 2:
        on_plane = pts[:, 0] == 0.0
    free = (~proj) & (~on_plane)
    pts[free] += (rng.random((int(free.sum()), dim)) - 0.5) * (jitter * hs[free])[:, None]
    if dim == 3:  # in-plane jitter for symmetry-plane points that are not on the axis
        pl = on_plane & (~on_axis) & (~proj)
        j = (rng.random((int(pl.sum()), 3)) - 0.5) * (jitter * hs[pl])[:, None]
        j[:, 1] = 0.0
        pts[pl] += j

    # snap electrodes onto the nearest axis vertex
    ax_idx = np.nonzero(on_axis & (~proj))[0]
    zc = dim - 1
    for zs in list(sources_z) + list(snap_z):
        if abs(zs) >= R or ax_idx.size == 0:
            continue
        k = ax_idx[np.argmin(np.abs(pts[ax_idx, zc] - zs))]
        pts[k, zc] = zs

    on_sph = np.sqrt((pts ** 2).sum(1)) >= R * (1 - 1e-12)

    def triangulate(pts):
        # Qhull is ~10x slower on exactly coplanar hull points (symmetry plane / axis): triangulate
        # a copy lifted off the plane by ~1e-9 R, then use the exact coordinates; the flat hull
        # elements this creates have exactly zero measure afterwards and are removed.
        lifted = pts.copy()
        lifted[on_plane, 0 if dim == 2 else 1] += (0.5 + rng.random(int(on_plane.sum()))) * 1e-9 * R
        conn = Delaunay(lifted).simplices.astype(np.int64)
        P = pts[conn]
        if dim == 2:
            a = P[:, 1] - P[:, 0]; b = P[:, 2] - P[:, 0]
            vol = 0.5 * np.abs(a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0])
        else:
            a = P[:, 1] - P[:, 0]; b = P[:, 2] - P[:, 0]; c = P[:, 3] - P[:, 0]
            vol = np.abs(np.einsum("ij,ij->i", a, np.cross(b, c))) / 6.0
        edges = np.stack([np.sqrt(((P[:, i] - P[:, j]) ** 2).sum(1)) for i in range(dim + 1) for j in range(i + 1, dim + 1)], 1)
        # drop flat elements (coplanar hull points) and elements with every vertex on the sphere
        ok = (vol > 1e-9 * edges.max(1) ** dim) & (~np.all(on_sph[conn], axis=1))
        conn, vol, edges = conn[ok], vol[ok], edges[ok]
        rms = np.sqrt((edges ** 2).mean(1))
        qual = vol / rms ** dim * (4 / np.sqrt(3) if dim == 2 else 6 * np.sqrt(2))   # 1 for the regular simplex
        return conn, vol, edges, qual

    conn, vol, edges, qual = triangulate(pts)
    # Sliver removal.  A plain Delaunay tetrahedralisation keeps a few almost flat elements
    # ("slivers"); production meshers (Gmsh's optimiser, which the reference relies on) remove them,
    # and they cost the Jacobi-PCG ~1.5x in steps.  Perturb the movable vertices of poor elements by a
    # fraction of their shortest edge and re-triangulate, a few passes (seeded, deterministic).
    pinned = on_sph | on_axis
    for zs in list(sources_z) + list(snap_z):
        if ax_idx.size:
            pinned[ax_idx[np.argmin(np.abs(pts[ax_idx, zc] - zs))]] = True
    for _ in range(improve_passes if dim == 3 else 0):
        bad = qual < 0.15
        if not bad.any():
            break
        vb = conn[bad]; hb = edges[bad].min(1)
        for k in range(dim + 1):
            v = vb[:, k]
            mv = ~pinned[v]
            step = (rng.random((int(mv.sum()), dim)) - 0.5) * 0.5 * hb[mv][:, None]
            step[on_plane[v[mv]], 1] = 0.0     # symmetry-plane vertices move in their plane
            cand = pts[v[mv]] + step
            # a move must keep the vertex strictly inside the half ball (a point pushed through the
            # symmetry plane or the sphere would become a hull vertex and wreck the triangulation)
            inside = (np.sqrt((cand ** 2).sum(1)) < R - 0.5 * hb[mv]) & (on_plane[v[mv]] | (cand[:, 1] > 0.5 * hb[mv]))
            pts[v[mv][inside]] = cand[inside]
        conn, vol, edges, qual = triangulate(pts)
    emax = edges.max(1)

    # remove unused points, renumber along a Morton-like order for locality (sort by tree cell)
    used = np.zeros(len(pts), bool); used[conn.ravel()] = True
    q = np.floor((pts - pts.min(0)) / (pts.max(0) - pts.min(0) + 1e-300) * 1023).astype(np.int64)
    key = np.zeros(len(pts), dtype=np.int64)
    for bit in range(10):
        for k in range(dim):
            key |= ((q[:, k] >> bit) & 1) << (bit * dim + k)
    order = np.argsort(key, kind="stable")
    order = order[used[order]]
    new_id = np.full(len(pts), -1, dtype=np.int64); new_id[order] = np.arange(order.size)
    pts = pts[order]; conn = new_id[conn]
    node_h, node_iface = hs[order], is_iface[order]

    valence = np.bincount(conn.ravel(), minlength=len(pts)).max()
    if valence > 400:   # a vertex shared by hundreds of elements means the hull was broken
        raise RuntimeError("mesh generation produced a degenerate triangulation (vertex valence %d)" % valence)
    bf = _boundary_facets(conn)
    rad = np.sqrt((pts ** 2).sum(1))
    bdir = np.all(rad[bf] >= R * (1 - 1e-9), axis=1)
    cent = pts[conn].mean(1)
    mat = np.zeros(len(conn), dtype=np.int32) if material_fn is None else np.asarray(material_fn(cent), dtype=np.int32)
    exact = (np.pi * R * R / 2) if dim == 2 else (2.0 / 3.0 * np.pi * R ** 3)
    meta = dict(R=R, scale=scale, seed=seed, sources_z=[float(s) for s in sources_z],
                volume=float(vol.sum()), volume_exact=float(exact), min_quality=float(qual.min()),
                max_valence=int(valence), n_interface_points=int(n_iface), node_h=node_h, node_iface=node_iface)
    return Mesh(dim, np.ascontiguousarray(pts), np.ascontiguousarray(conn.astype(np.int32)),
                np.ascontiguousarray(mat), np.ascontiguousarray(bf.astype(np.int32)),
                np.ascontiguousarray(bdir.astype(np.uint8)), meta)


# ---------------------------------------------------------------------------------------------
# on-disk cache of generated meshes
#
# A lattice mesh depends on the electrode pattern of a batch in the batch-centred frame, the size multiplier and the seed -
# not on the depth: the 40 batches of the bench's sweep use TWO distinct meshes (one per tool), and so do all ranks of a
# multi-GPU run, all legs of one bench run and a rerun.  A size-L mesh takes 17-22 s of one core, so the meshes are kept as
# .npz files keyed by a hash of those arguments (and of this file, so that a change of the mesher invalidates them).
# One process builds, the others wait on the file's lock and load.  REMO_MESH_CACHE = directory (default: a per-user directory
# under /dev/shm, else the temporary directory), "0" = no cache.

_CODE_TAG = None


def mesh_cache_dir() -> Optional[str]:
    import os
    import tempfile
    d = os.environ.get("REMO_MESH_CACHE")
    if d in ("0", ""):
        return None
    if d is None:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        d = os.path.join(base, "remo3d_mesh_cache_%d" % os.getuid())
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        return None
    return d


def cached_mesh(key: tuple, build: Callable[[], Mesh]) -> Mesh:
    """build() through the on-disk cache: key = everything the mesh depends on (plain numbers / tuples)."""
    import hashlib
    import os
    global _CODE_TAG
    d = mesh_cache_dir()
    if d is None:
        return build()
    if _CODE_TAG is None:
        with open(__file__, "rb") as f:
            _CODE_TAG = hashlib.sha1(f.read()).hexdigest()[:12]
    name = hashlib.sha1(repr((key, _CODE_TAG, np.__version__)).encode()).hexdigest()[:24]
    path = os.path.join(d, name + ".npz")

    def load():
        with np.load(path, allow_pickle=False) as z:
            meta = {k[5:]: (z[k] if z[k].ndim else z[k].item()) for k in z.files if k.startswith("meta_")}
            if "sources_z" in meta:
                meta["sources_z"] = [float(v) for v in np.atleast_1d(meta["sources_z"])]
            return Mesh(int(z["dim"]), z["coords"], z["conn"], z["mat"], z["bconn"], z["bdirichlet"], meta)

    if os.path.exists(path):
        try:
            return load()
        except Exception:
            pass                     # unreadable (e.g. a crashed writer of an older layout): rebuild below
    try:
        import fcntl
        lock = open(path + ".lock", "w")
        fcntl.flock(lock, fcntl.LOCK_EX)      # the first process builds, the others wait here and find the file
    except Exception:
        lock = None
    try:
        if os.path.exists(path):
            try:
                return load()
            except Exception:
                pass
        mesh = build()
        try:
            tmp = "%s.%d.tmp.npz" % (path, os.getpid())
            meta = {"meta_" + k: np.asarray(v) for k, v in mesh.meta.items() if isinstance(v, (int, float, list, tuple, np.ndarray, np.generic))}
            np.savez(tmp, dim=np.int64(mesh.dim), coords=mesh.coords, conn=mesh.conn, mat=mesh.mat, bconn=mesh.bconn, bdirichlet=mesh.bdirichlet, **meta)
            os.replace(tmp, path)
        except OSError:
            pass                     # a full or read-only cache directory is not an error
        return mesh
    finally:
        if lock is not None:
            lock.close()


# ---------------------------------------------------------------------------------------------
# material models (centroid classification) for the reference's benchmark inputs


def layered_material_fn(dim: int, local_formation_geometry: np.ndarray, local_borehole_geometry: np.ndarray,
                        dip_rad: float = 0.0):
    """Material classifier for a windowed model as produced by geometry.select_gmsh_data_range.

    Material order follows the reference (gmsh_functions.py:172, 452-480 / 591-624):
    0 = borehole mud, then for every layer from top (smallest z in the local frame) to bottom:
    its flushed zone (if FZ_RADIUS is not NaN) followed by its undisturbed zone.
    local_formation_geometry: [L, 3] = (top, bottom, fz_radius) relative to the batch centre.
    local_borehole_geometry : [B, 2] = (z, radius) relative to the batch centre.
    The local frame has z increasing with depth, as in the reference's Gmsh models.
    """
    fg = np.asarray(local_formation_geometry, dtype=np.float64)
    bg = np.asarray(local_borehole_geometry, dtype=np.float64)
    first = []
    k = 1
    for i in range(fg.shape[0]):
        first.append(k)
        k += 1 if np.isnan(fg[i, 2]) else 2
    first = np.asarray(first)
    tan_d = np.tan(dip_rad)

    def fn(c: np.ndarray) -> np.ndarray:
        if dim == 2:
            rho = np.abs(c[:, 0]); z = c[:, 1]; zl = z
        else:
            rho = np.hypot(c[:, 0], c[:, 1]); z = c[:, 2]
            zl = z + tan_d * c[:, 0]     # dipping planes: slab rotated about y (gmsh_functions.py:610-611)
        rb = np.interp(z, bg[:, 0], bg[:, 1])
        li = np.clip(np.searchsorted(fg[:, 1], zl, side="right"), 0, fg.shape[0] - 1)
        m = first[li].copy()
        fz = fg[li, 2]
        undisturbed = (~np.isnan(fz)) & (rho >= np.where(np.isnan(fz), 0.0, fz))
        m[undisturbed] += 1
        m[rho < rb] = 0
        return m

    return fn


def layer_interfaces_2d(local_formation_geometry: np.ndarray, local_borehole_geometry: np.ndarray, R: float):
    """Material interfaces of a windowed axisymmetric model as polylines of (r, z) points: the
    borehole wall, the layer boundaries outside the borehole and the flushed-zone radii - the curves
    the reference hands to its mesher as geometry (gmsh_functions.py:404-480, netgen_functions.py:129-311).
    Junction points are shared between the polylines that meet there."""
    fg = np.asarray(local_formation_geometry, dtype=np.float64)
    bg = np.asarray(local_borehole_geometry, dtype=np.float64)
    rb = lambda z: float(np.interp(z, bg[:, 0], bg[:, 1]))
    bounds = [float(fg[i, 1]) for i in range(fg.shape[0] - 1) if abs(fg[i, 1]) < R]
    # borehole wall with the junctions of the layer boundaries inserted
    zs = sorted(set(list(bg[:, 0]) + [b for b in bounds if bg[0, 0] < b < bg[-1, 0]]))
    wall = np.array([[rb(z), z] for z in zs])
    polys = [wall]
    for i, b in enumerate(bounds):
        r0, r1 = rb(b), float(np.sqrt(max(R * R - b * b, 0.0)))
        if r1 <= r0:
            continue
        stops = [r0, r1]
        for layer in (i, i + 1):                       # flushed-zone radii ending on this boundary
            fz = fg[layer, 2]
            if not np.isnan(fz) and r0 < fz < r1:
                stops.append(float(fz))
        stops = sorted(set(stops))
        polys.append(np.array([[r, b] for r in stops]))
    for i in range(fg.shape[0]):
        fz = fg[i, 2]
        if np.isnan(fz) or fz >= R:
            continue
        zlim = float(np.sqrt(R * R - fz * fz))
        z0, z1 = max(float(fg[i, 0]), -zlim), min(float(fg[i, 1]), zlim)
        if z1 > z0:
            polys.append(np.array([[fz, z0], [fz, z1]]))
    return polys


# ---------------------------------------------------------------------------------------------
# interface-conforming 3D meshes of dipping models


def _revolve_triangulation(p2: np.ndarray, tri: np.ndarray, m: int, kappa: float):
    """Revolve a 2D (r, z) triangulation about the axis r = 0 over the half circle in m sectors.
    Returns (points [N3, 3] as (x, y, z), tets [T3, 4], parent triangle of every tet, 3D node ids of
    the 2D nodes on plane j as an [N2, m + 1] table).  Nodes on the axis are shared by all planes.
    Prisms / pyramids are cut with the smallest-vertex-number rule (Dompierre et al. 1999), which makes
    the diagonals of the quadrilateral faces agree between neighbours without any search."""
    n2 = len(p2)
    on_axis = p2[:, 0] == 0.0
    ids = np.empty((n2, m + 1), dtype=np.int64)
    n_ax = int(on_axis.sum())
    ids[on_axis, :] = np.arange(n_ax)[:, None]
    off = np.nonzero(~on_axis)[0]
    ids[off, :] = n_ax + np.arange(off.size)[:, None] * (m + 1) + np.arange(m + 1)[None, :]
    th = np.arange(m + 1) * (np.pi / m)
    pts = np.empty((n_ax + off.size * (m + 1), 3))
    pts[:n_ax] = np.stack([np.zeros(n_ax), np.zeros(n_ax), p2[on_axis, 1]], 1)
    r = kappa * p2[off, 0]
    c, sn = np.cos(th), np.sin(th)
    sn[0] = 0.0; sn[-1] = 0.0; c[0] = 1.0; c[-1] = -1.0            # exact symmetry plane
    pts[n_ax:] = np.stack([(r[:, None] * c[None, :]).ravel(), (r[:, None] * sn[None, :]).ravel(),
                           np.repeat(p2[off, 1], m + 1)], 1)
    tets, parent = [], []
    nax = on_axis[tri].sum(1)
    for j in range(m):
        lo, hi = ids[:, j], ids[:, j + 1]
        # --- no axis vertex: prism V1..V6 = a0 b0 c0 a1 b1 c1
        t0 = np.nonzero(nax == 0)[0]
        if t0.size:
            T = tri[t0]
            V = np.stack([lo[T[:, 0]], lo[T[:, 1]], lo[T[:, 2]], hi[T[:, 0]], hi[T[:, 1]], hi[T[:, 2]]], 1)
            # rotate so that the smallest id sits at V1 (prism symmetries: cyclic shifts of (abc), swap of the two caps)
            k = np.argmin(V, axis=1)
            perms = np.array([[0, 1, 2, 3, 4, 5], [1, 2, 0, 4, 5, 3], [2, 0, 1, 5, 3, 4],
                              [3, 5, 4, 0, 2, 1], [4, 3, 5, 1, 0, 2], [5, 4, 3, 2, 1, 0]])
            V = np.take_along_axis(V, perms[k], axis=1)
            first = np.minimum(V[:, 1], V[:, 5]) < np.minimum(V[:, 2], V[:, 4])
            A = np.where(first[:, None, None],
                         np.stack([V[:, [0, 1, 2, 5]], V[:, [0, 1, 5, 4]], V[:, [0, 4, 5, 3]]], 1),
                         np.stack([V[:, [0, 1, 2, 4]], V[:, [0, 4, 2, 5]], V[:, [0, 4, 5, 3]]], 1))
            tets.append(A.reshape(-1, 4)); parent.append(np.repeat(t0, 3))
        # --- one axis vertex: pyramid apex a, base b0 c0 c1 b1
        t1 = np.nonzero(nax == 1)[0]
        if t1.size:
            T = tri[t1]
            sh = np.argmax(on_axis[T], axis=1)                        # rotate the axis vertex to the front
            T = np.take_along_axis(T, (sh[:, None] + np.arange(3)[None, :]) % 3, axis=1)
            ap, b0, c0, c1, b1 = lo[T[:, 0]], lo[T[:, 1]], lo[T[:, 2]], hi[T[:, 2]], hi[T[:, 1]]
            base = np.stack([b0, c0, c1, b1], 1)
            kmin = np.argmin(base, axis=1)
            diag_b0c1 = (kmin == 0) | (kmin == 2)
            A = np.where(diag_b0c1[:, None, None],
                         np.stack([np.stack([ap, b0, c0, c1], 1), np.stack([ap, b0, c1, b1], 1)], 1),
                         np.stack([np.stack([ap, b0, c0, b1], 1), np.stack([ap, c0, c1, b1], 1)], 1))
            tets.append(A.reshape(-1, 4)); parent.append(np.repeat(t1, 2))
        # --- two axis vertices: one tetrahedron
        t2 = np.nonzero(nax == 2)[0]
        if t2.size:
            T = tri[t2]
            sh = np.argmin(on_axis[T], axis=1)                        # the off-axis vertex to the front
            T = np.take_along_axis(T, (sh[:, None] + np.arange(3)[None, :]) % 3, axis=1)
            tets.append(np.stack([lo[T[:, 1]], lo[T[:, 2]], lo[T[:, 0]], hi[T[:, 0]]], 1)); parent.append(t2)
    return pts, np.concatenate(tets), np.concatenate(parent), ids


def make_mesh_3d_conforming(R: float, local_formation_geometry: np.ndarray, local_borehole_geometry: np.ndarray, dip_rad: float,
                            sources_z: Sequence[float] = (0.0,), snap_z: Sequence[float] = (), scale: float = 1.0, seed: int = 0,
                            layer_cap: Optional["LayerCap"] = None, sectors: int = 6, exact_radius: float = 0.5) -> Mesh:
    """Half-ball tetrahedral mesh whose faces follow the material interfaces of a dipping model, the
    geometry the reference builds with OpenCASCADE (gmsh_functions.py:543-628): borehole = body of
    revolution of the wall polyline, layer boundaries = planes z + x tan(dip) = const through the
    boundary depths on the axis, flushed zones = coaxial cylinders cut by those planes.

    In the sheared coordinates (x, y, z' = z + x tan(dip)) all of that is axisymmetric: the planes are
    horizontal and the vertical cylinders stay what they are.  So the interface-conforming 2D mesh of the
    dip-0 model (make_mesh(2, ..., interfaces=...), the mesher validated against the reference's 2D
    logs) is revolved about the axis in `sectors` sectors over the half space y >= 0 (the reference's
    size field h ~ rho asks for about three elements around the half circle at every radius), every
    tetrahedron inherits the material of its parent triangle, and the vertices are sheared back.
    Conformity holds by construction and is affine invariant.

    * circles become regular polygons whose circumradius is scaled so that the polygon AREA equals the
      circle's: every volume of revolution (mud column, flushed zones) is preserved exactly;
    * a caliper that varies with depth is revolved as it is seen on the axis' own z' (exact for a
      cylindrical hole, first order in radius * tan(dip) otherwise);
    * the sheared ball is inscribed in the physical sphere |x| = R and its outer part (beyond
      `exact_radius` of the way to the boundary) is stretched radially onto the sphere, so the Dirichlet
      boundary is the reference's; within that fraction the dipping geometry is exact.
    """
    a = float(np.tan(dip_rad))
    fg = np.asarray(local_formation_geometry, dtype=np.float64)
    bg = np.asarray(local_borehole_geometry, dtype=np.float64)
    m = int(sectors)
    kappa = float(np.sqrt(np.pi / (m * np.sin(np.pi / m))))      # equal-area regular 2m-gon
    # shear S: (x, z') -> (x, z' - a x); its singular values bound the radius of the sheared ball
    smax = float(np.sqrt(1.0 + 0.5 * a * a + a * np.sqrt(1.0 + 0.25 * a * a)))
    Rs = R / (smax * kappa)                                       # inscribed: the stretch below only ever pushes outwards
    polys = layer_interfaces_2d(fg, bg, Rs)
    inside_src = [z for z in list(sources_z) + list(snap_z) if abs(z) < Rs]
    fn2 = layered_material_fn(2, fg, bg)
    m2 = make_mesh(2, Rs, sources_z=inside_src, scale=scale, seed=seed, interfaces=polys, layer_cap=layer_cap, material_fn=fn2,
                   h_max=0.2 * R)
    p2 = m2.coords.copy()
    p2[np.abs(p2[:, 0]) < 1e-12 * R, 0] = 0.0
    pts, conn, parent, ids = _revolve_triangulation(p2, m2.conn.astype(np.int64), m, kappa)
    mat = m2.mat[parent]
    # Dirichlet surface: revolved rim nodes of the 2D mesh
    rim2 = np.zeros(len(p2), bool)
    rim2[np.unique(m2.bconn[m2.bdirichlet == 1])] = True
    on_rim = np.zeros(len(pts), bool)
    on_rim[np.unique(ids[rim2])] = True
    # back to physical coordinates, then the outer shell onto the sphere
    pts[:, 2] -= a * pts[:, 0]
    rho = np.sqrt((pts ** 2).sum(1))
    # boundary radius B of the sheared (and polygonally revolved) ball in every node's direction: a rim node IS the
    # boundary in its own direction; for inner nodes take it from the same 2D ray: p2 scaled to |p2| = Rs
    n2 = len(p2)
    s2 = np.hypot(p2[:, 0], p2[:, 1])
    t2 = np.clip(s2 / Rs, 0.0, 1.0)                                # fraction of the way to the boundary (sheared space)
    t3 = np.zeros(len(pts))
    for j in range(m + 1):
        t3[ids[:, j]] = t2
    safe = t3 > 0
    B = np.where(safe, rho / np.where(safe, t3, 1.0), R)          # affine map: radius scales linearly along a ray
    t1 = float(exact_radius)
    w = np.clip((t3 - t1) / (1.0 - t1), 0.0, 1.0)
    w = w * w * (3.0 - 2.0 * w)                                    # smoothstep: 0 inside, 1 on the boundary
    stretch = 1.0 + (R / B - 1.0) * w
    pts *= stretch[:, None]
    pts[on_rim] *= (R / np.sqrt((pts[on_rim] ** 2).sum(1)))[:, None]
    Q = pts[conn]
    e1 = Q[:, 1] - Q[:, 0]; e2 = Q[:, 2] - Q[:, 0]; e3 = Q[:, 3] - Q[:, 0]
    vol = np.abs(np.einsum("ij,ij->i", e1, np.cross(e2, e3))) / 6.0
    edges = np.stack([np.sqrt(((Q[:, i] - Q[:, j]) ** 2).sum(1)) for i in range(4) for j in range(i + 1, 4)], 1)
    qual = vol / np.sqrt((edges ** 2).mean(1)) ** 3 * (6 * np.sqrt(2))
    if not (vol > 1e-12 * edges.max(1) ** 3).all():
        raise RuntimeError("revolved mesh has a degenerate element")
    # locality: Morton order of the nodes
    q = np.floor((pts - pts.min(0)) / (pts.max(0) - pts.min(0) + 1e-300) * 1023).astype(np.int64)
    key = np.zeros(len(pts), dtype=np.int64)
    for bit in range(10):
        for k in range(3):
            key |= ((q[:, k] >> bit) & 1) << (bit * 3 + k)
    order = np.argsort(key, kind="stable")
    new_id = np.empty(len(pts), dtype=np.int64); new_id[order] = np.arange(len(pts))
    pts = pts[order]; conn = new_id[conn]; on_rim = on_rim[order]
    bf = _boundary_facets(conn)
    bdir = np.all(on_rim[bf], axis=1)
    valence = int(np.bincount(conn.ravel(), minlength=len(pts)).max())
    meta = dict(R=R, scale=scale, seed=seed, sources_z=[float(s) for s in sources_z], volume=float(vol.sum()),
                volume_exact=float(2.0 / 3.0 * np.pi * R ** 3), min_quality=float(qual.min()), max_valence=valence,
                n_interface_points=int(m2.meta["n_interface_points"]), dip_rad=float(dip_rad), sectors=m, kappa=kappa,
                sheared_radius=Rs, exact_radius=float(t1 * Rs / smax), n_triangles_2d=int(m2.n_elems))
    return Mesh(3, np.ascontiguousarray(pts), np.ascontiguousarray(conn.astype(np.int32)), np.ascontiguousarray(mat.astype(np.int32)),
                np.ascontiguousarray(bf.astype(np.int32)), np.ascontiguousarray(bdir.astype(np.uint8)), meta)


def interface_straddlers(mesh: Mesh, local_formation_geometry, local_borehole_geometry, dip_rad: float, tol: float = 1e-7):
    """Number of elements whose vertices lie strictly on both sides of a material interface (layer
    plane, flushed-zone cylinder or borehole wall), i.e. elements a conforming mesh must not have.
    For revolved 3D meshes (meta has "sectors") circles are the equal-area polygons of the mesher, and only
    the part of the mesh inside meta["exact_radius"] is looked at (beyond it the geometry is stretched)."""
    fg = np.asarray(local_formation_geometry, dtype=np.float64)
    bg = np.asarray(local_borehole_geometry, dtype=np.float64)
    a = np.tan(dip_rad)
    c = mesh.coords
    R = mesh.meta.get("R", np.inf)
    limit = 0.9 * R
    if mesh.dim == 2:
        rho, z, zl = np.abs(c[:, 0]), c[:, 1], c[:, 1]
    else:
        rho, z, zl = np.hypot(c[:, 0], c[:, 1]), c[:, 2], c[:, 2] + a * c[:, 0]
        if "sectors" in mesh.meta:      # nominal radius of the polygonal "circle" through the node
            m, kappa = mesh.meta["sectors"], mesh.meta["kappa"]
            th = np.arctan2(c[:, 1], c[:, 0])
            phi = np.mod(th, np.pi / m) - np.pi / (2 * m)
            rho = rho * np.cos(phi) / (kappa * np.cos(np.pi / (2 * m)))
            limit = 0.98 * mesh.meta["exact_radius"]
    count = np.zeros(mesh.n_elems, bool)
    conn = mesh.conn

    def straddle(f):
        fe = f[conn]
        return (fe.min(1) < -tol) & (fe.max(1) > tol)

    rbn = np.interp(zl if "sectors" in mesh.meta else z, bg[:, 0], bg[:, 1])
    count |= straddle(rho - rbn)
    outside_bh = (rho[conn] >= rbn[conn] - tol).all(1)
    for i in range(fg.shape[0] - 1):
        b = fg[i, 1]
        if abs(b) < R:
            count |= straddle(zl - b) & outside_bh
    for i in range(fg.shape[0]):
        fz = fg[i, 2]
        if not np.isnan(fz):
            inlayer = ((zl[conn] >= fg[i, 0] - tol) & (zl[conn] <= fg[i, 1] + tol)).all(1)
            count |= straddle(rho - fz) & inlayer
    far = (np.sqrt((c ** 2).sum(1))[conn] > limit).any(1)     # interfaces are not tracked onto the far boundary
    return int((count & ~far).sum())
