"""Windowing of the global borehole / formation model to the simulation sphere of one batch:
the build's own restatement of remo3d/gmsh_functions.py:10-174 (SelectGmshDataRange and its two
helpers).  Output frame: depth relative to the batch's combined depth, z positive downwards.

Quirks of the reference that are kept because they define its results (pinned by
tests/golden/windows_*.json, generated from the reference in this container):
  * the borehole polyline keeps one extra sample on each side of the window (3-tap dilation);
  * in dipping models the borehole window is a slab |z| < R, not a sphere;
  * layers are kept when either boundary is closer than 0.99 R to the window centre, measured
    perpendicular to the (dipping) boundary; flushed zones that do not reach into that radius are
    merged into the undisturbed zone;
  * the first / last layer is stretched to +-1.01 R (times sqrt(1 + tan^2 dip) when dipping).
"""
from __future__ import annotations

import numpy as np


def _segment_circle_hit(p1, p2, radius):
    """Intersection of segment p1->p2 (rows are (z, r)) with the circle z^2 + r^2 = radius^2 that
    lies strictly inside the segment (gmsh_functions.py:12-25)."""
    x1, y1 = p1[1], p1[0]
    x2, y2 = p2[1], p2[0]
    dx, dy = x2 - x1, y2 - y1
    dr2 = dx * dx + dy * dy
    D = x1 * y2 - x2 * y1
    disc = radius ** 2 * dr2 - D ** 2
    for sign in (-1, 1):
        x = (D * dy + sign * np.sign(dy) * dx * np.sqrt(disc)) / dr2
        y = (-D * dx + sign * np.abs(dy) * np.sqrt(disc)) / dr2
        p = np.array([y, x])
        t = np.dot(p1 - p2, p1 - p)
        if 0 < t < np.dot(p1 - p2, p1 - p2):
            return p
    return None


def window_borehole(borehole_geometry, dip, depth, R):
    """Local borehole wall polyline [(z, radius)] clipped to the domain (gmsh_functions.py:10-90)."""
    bg = np.asarray(borehole_geometry, dtype=float)
    if bg.shape[0] == 2:
        loc = bg.copy()
    else:
        if dip == 0:
            inside = (bg[:, 0] - depth) ** 2 + bg[:, 1] ** 2 < R ** 2
        else:
            inside = np.abs(bg[:, 0] - depth) < R
        grown = inside.copy()
        grown[:-1] |= inside[1:]
        grown[1:] |= inside[:-1]
        loc = bg[grown, :].copy()
    loc[:, 0] -= depth

    def on_or_in(z, r):
        if dip == 0:
            q = z * z + r * r
            return (q == R * R), (q < R * R)
        return (abs(z) == R), (abs(z) < R)

    for end in (0, -1):
        nxt = 1 if end == 0 else -2
        z, r = loc[end]
        on, inn = on_or_in(z, r)
        sgn = -1.0 if end == 0 else 1.0
        if on:
            continue
        if inn:  # extend straight up / down to the domain boundary at the same radius
            if dip == 0:
                omega = np.arccos(r / R)
                new = np.array([sgn * np.sin(omega) * R, r])
            else:
                new = np.array([sgn * R, r])
            loc = np.vstack((new, loc)) if end == 0 else np.vstack((loc, new))
        else:    # pull the outside point back onto the boundary
            if dip == 0:
                loc[end, :] = _segment_circle_hit(loc[end, :], loc[nxt, :], R)
            else:
                a = abs(loc[end, 0]) - R
                b = R - sgn * loc[nxt, 0]
                loc[end, :] = [sgn * R, (b * loc[end, 1] + a * loc[nxt, 1]) / (a + b)]
    return loc


def window_formation(formation_parameters, dip, depth, R, active_geometry_window=0.99):
    """Local layer table [(top, bottom, fz_radius)] and the resistivity list in material order
    (gmsh_functions.py:92-165)."""
    fp = np.asarray(formation_parameters, dtype=float)
    active = R * active_geometry_window
    loc = fp.copy()
    loc[:, :2] -= depth
    if dip == 0:
        a = 0.0
        dist = np.abs(loc[:, :2])
    else:
        a = np.tan(dip)
        dist = np.abs(loc[:, :2]) / np.sqrt(a * a + 1.0)
    layers = loc[np.any(dist < active, axis=1), :]

    has_fz = ~np.isnan(layers[:, 2])
    if dip == 0:
        xs = np.repeat(layers[has_fz, 2][:, None], 2, axis=1)
        ys = layers[has_fz, :2]
    else:
        xs = np.repeat(layers[has_fz, 2][:, None], 4, axis=1)
        xs[:, :2] *= -1
        ys = a * xs + np.hstack([layers[has_fz, :2], layers[has_fz, :2]])
    reach = np.any(np.sqrt(xs ** 2 + ys ** 2) < active, axis=1)
    drop = has_fz.copy()
    drop[has_fz] = ~reach

    model = layers.copy()
    with_res = fp.shape[1] == 5
    if with_res:
        model[drop, 4] = model[drop, 3]
        model[drop, 2:4] = np.nan
    else:
        model[drop, 2] = np.nan
    stretch = R * 1.01 if dip == 0 else R * np.sqrt(a * a + 1.0) * 1.01
    if model[0, 0] > -stretch:
        model[0, 0] = -stretch
    if model[-1, 1] < stretch:
        model[-1, 1] = stretch
    if not with_res:
        return model
    res = model[:, 3:5].ravel()
    return model[:, :3], res[~np.isnan(res)]


def select_data_range(borehole_geometry, formation_parameters, dip, mud_resistivity, depth, R, active_geometry_window=0.99):
    """(local_formation_geometry, local_borehole_geometry, sigma) with sigma = [1/Rm] + 1/R_zones
    (gmsh_functions.py:168-174); the order of sigma is the material numbering of the mesh."""
    bh = window_borehole(borehole_geometry, dip, depth, R)
    fg, res = window_formation(formation_parameters, dip, depth, R, active_geometry_window)
    sigma = [1.0 / mud_resistivity] + list(1.0 / res)
    return fg, bh, sigma


# ---------------------------------------------------------------------------------------------
# Netgen path (2D only): remo3d/netgen_functions.py:12-118


def _segment_circle_hit_side(p1, p2, radius, side):
    """Like _segment_circle_hit, restricted to the upper (z < 0) or lower (z > 0) half
    (netgen_functions.py:14-29)."""
    x1, y1 = p1[1], p1[0]
    x2, y2 = p2[1], p2[0]
    dx, dy = x2 - x1, y2 - y1
    dr2 = dx * dx + dy * dy
    D = x1 * y2 - x2 * y1
    disc = radius ** 2 * dr2 - D ** 2
    for sign in (-1, 1):
        x = (D * dy + sign * np.sign(dy) * dx * np.sqrt(disc)) / dr2
        y = (-D * dx + sign * np.abs(dy) * np.sqrt(disc)) / dr2
        p = np.array([y, x])
        t = np.dot(p1 - p2, p1 - p)
        if ((side == "top" and y < 0) or (side == "bottom" and y > 0)) and 0 < t < np.dot(p1 - p2, p1 - p2):
            return p
    return None


def select_netgen_data_range(borehole_geometry, formation_parameters, mud_resistivity, depth, R, active_geometry_window=0.999):
    """Windowing of the reference's default 2D path (mesh_generator "netgen", remo3d.py:776-779):
    returns (local_formation_geometry [L, 5] = top, bottom, fz_radius, region numbers left / right,
    local_borehole_geometry [B, 2], sigma).  Quirks kept because they define the reference's inputs:
    the position of the borehole end points relative to the domain is tested with z^2 + r (radius NOT
    squared, netgen_functions.py:43-62); the active radius is 0.999 R; a flushed zone is dropped only if
    both inner corners AND the connecting line lie outside the active radius; the first / last layer is
    cut at the borehole polyline's end points instead of being stretched."""
    bg = np.asarray(borehole_geometry, dtype=float)
    fp = np.asarray(formation_parameters, dtype=float)
    if bg.shape[0] == 2:
        loc = bg.copy()
    else:
        inside = (bg[:, 0] - depth) ** 2 + bg[:, 1] ** 2 < R ** 2
        grown = inside.copy()
        grown[:-1] |= inside[1:]
        grown[1:] |= inside[:-1]
        loc = bg[grown, :].copy()
    loc[:, 0] -= depth
    for end, side in ((0, "top"), (-1, "bottom")):
        nxt = 1 if end == 0 else -2
        sgn = -1.0 if end == 0 else 1.0
        q = loc[end, 0] ** 2 + loc[end, 1]          # sic: radius not squared
        if np.isclose(q, R ** 2):
            continue
        if q < R ** 2:
            omega = np.arccos(loc[end, 1] / R)
            new = np.array([sgn * np.sin(omega) * R, loc[end, 1]])
            loc = np.vstack((new, loc)) if end == 0 else np.vstack((loc, new))
        else:
            loc[end, :] = _segment_circle_hit_side(loc[end, :], loc[nxt, :], R, side)

    active = R * active_geometry_window
    rel = fp[:, :2] - depth
    point_within = np.any(rel ** 2 <= active ** 2, axis=1)
    line_across = np.all(rel ** 2 > active ** 2, axis=1) & (fp[:, 0] < depth) & (fp[:, 1] > depth)
    model = fp[point_within | line_across, :].copy()
    model[:, :2] -= depth
    has_fz = ~np.isnan(model[:, 2])
    top_out = model[:, 0] ** 2 + model[:, 2] ** 2 >= active ** 2
    bot_out = model[:, 1] ** 2 + model[:, 2] ** 2 >= active ** 2
    line_out = ~((model[:, 0] < 0) & (model[:, 1] > 0) & (model[:, 2] < active))
    drop = has_fz & top_out & bot_out & line_out
    model[drop, 2] = np.nan
    model[drop, 4] = model[drop, 3]
    model[drop, 3] = np.nan
    if model[0, 0] != loc[0, 0]:
        model[0, 0] = loc[0, 0]
    if model[-1, 1] != loc[-1, 0]:
        model[-1, 1] = loc[-1, 0]
    grid = np.empty((model.shape[0], 2))
    region = 2
    for i in range(model.shape[0]):
        if np.isnan(model[i, 3]):
            grid[i, :] = region
            region += 1
        else:
            grid[i, 0] = region
            grid[i, 1] = region + 1
            region += 2
    res = model[:, 3:5].ravel()
    res = res[~np.isnan(res)]
    sigma = [1.0 / mud_resistivity] + list(1.0 / res)
    return np.hstack((model[:, :3], grid)), loc, sigma
