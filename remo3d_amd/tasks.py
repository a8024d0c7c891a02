"""Batching of measurement depths into units of work: the build's own restatement of
remo3d.py:602-692 (_prepare_simulation_depths_and_tasks).

A *batch* is one mesh around ``combined_depth`` plus up to ``batch_size`` *solves* (right-hand
sides: one per simulated current-electrode depth); every solve serves one or more *records*
(measurement depth, tool) that read the potential at that tool's measuring electrodes.  This is
the unit the GPU library consumes (remo_solve_batch: one mesh, n_rhs sources, eval points).

Pinned by tests/golden/tasks_*.json generated from the reference in this container.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np


@dataclass
class Record:
    depth_index: int     # index into measurement_depths
    tool_index: int      # index into the tools dict order
    offset: float        # simulated depth minus the batch's combined depth


@dataclass
class Solve:
    simulation_depth_index: int
    electrodes: np.ndarray            # [2, m]: positions (batch frame) and source terms, sorted by position
    records: List[Record] = field(default_factory=list)


@dataclass
class Batch:
    index: int
    combined_depth: float
    electrodes: np.ndarray            # [2, m] union over the batch (what the mesher refines around)
    solves: List[Solve] = field(default_factory=list)


def _merge(potential, current):
    cur = np.unique(current)
    pot = np.unique(potential)
    pot = pot[~np.isin(pot, cur)]
    e = np.hstack([np.vstack([pot, np.zeros_like(pot)]), np.vstack([cur, np.ones_like(cur)])])
    return e[:, e[0, :].argsort()]


def build_batches(tools: Dict[str, np.ndarray], sec: bool, measurement_depths: np.ndarray, batch_size: int):
    """Returns (combined_depths[n_batches], batches)."""
    md = np.asarray(measurement_depths, dtype=float)
    names = list(tools.keys())
    per_tool = {t: np.round(md + tools[t][1, 3], decimals=4) for t in names}
    if sec:
        sim = np.unique(np.hstack(list(per_tool.values())))
        sim_tool = None
    else:
        sim = np.hstack(list(per_tool.values()))
        sim_tool = [ti for ti in range(len(names)) for _ in range(len(md))]
        order = np.argsort(sim)
        sim = sim[order]
        sim_tool = [sim_tool[i] for i in order]
    nb = int(np.ceil(sim.size / batch_size))
    grid = np.pad(sim.astype(float), (0, nb * batch_size - sim.size), mode="constant", constant_values=np.nan).reshape(nb, batch_size)
    combined = np.round(np.nanmean(grid, axis=1), decimals=4)
    offsets = np.round(grid - combined[:, None], decimals=4)

    batches: List[Batch] = []
    for bi in range(nb):
        b_cur, b_pot = [], []
        batch = Batch(bi, float(combined[bi]), np.zeros((2, 0)))
        for di in range(batch_size):
            depth = grid[bi, di]
            if np.isnan(depth):
                break
            off = offsets[bi, di]
            sdi = bi * batch_size + di
            if sec:
                cur, pot, records = [], [], []
                for ti, t in enumerate(names):
                    if np.any(np.isclose(per_tool[t], depth)):
                        mdi = int(np.argwhere(np.isclose(md + tools[t][1, 3], depth))[0][0])
                        records.append(Record(mdi, ti, float(off)))
                        el = tools[t][:, :3].copy()
                        el[0, :] += off
                        el = np.round(el, 4)
                        c = list(el[0, el[1, :] != 0]); p = list(el[0, el[1, :] == 0])
                        cur += c; pot += p; b_cur += c; b_pot += p
                electrodes = _merge(pot, cur)
            else:
                ti = sim_tool[sdi]
                t = names[ti]
                mdi = int(np.argwhere(np.isclose(md + tools[t][1, 3], depth))[0][0])
                records = [Record(mdi, ti, float(off))]
                el = tools[t][:, :3].copy()
                el[0, :] += off
                el = np.round(el, 4)
                b_cur += list(el[0, el[1, :] != 0]); b_pot += list(el[0, el[1, :] == 0])
                electrodes = el[:, el[0, :].argsort()]
            batch.solves.append(Solve(sdi, electrodes, records))
        batch.electrodes = _merge(b_pot, b_cur)
        batches.append(batch)
    return combined, batches


def to_reference_layout(batches: Sequence[Batch]):
    """The nested-list task layout of remo3d.py:690 (for comparison with the golden vectors)."""
    out = []
    for b in batches:
        out.append([b.index, b.electrodes.tolist(),
                    [[s.simulation_depth_index, s.electrodes.tolist(), [[r.depth_index, r.tool_index, r.offset] for r in s.records]]
                     for s in b.solves]])
    return out


def batch_rhs(batch: Batch, tools: Dict[str, np.ndarray]):
    """Sources / evaluation points of a batch in the form remo_solve_batch takes, plus the map
    back to records: returns (sources, evals, readers) with readers[k] = list of
    (depth_index, tool_index, K, slice into evals[k]) following worker.py:113-131."""
    names = list(tools.keys())
    sources, evals, readers = [], [], []
    for s in batch.solves:
        z = s.electrodes[0, :]
        I = s.electrodes[1, :]
        sources.append((z[I != 0].copy(), I[I != 0].copy()))
        pts, rd = [], []
        for r in s.records:
            t = tools[names[r.tool_index]]
            geo = t[0, :3] + r.offset
            meas = geo[t[1, :3] == 0]
            rd.append((r.depth_index, r.tool_index, float(t[0, 3]), len(pts), len(meas)))
            pts += list(meas)
        evals.append(np.asarray(pts, dtype=float))
        readers.append(rd)
    return sources, evals, readers


def apparent_resistivity(u: np.ndarray, n_meas: int, K: float, dim: int) -> float:
    """Ra from potentials at the measuring electrodes (worker.py:124-131): |K (u2 - u1)| or
    |K u1|, halved in 3D because only a half-space is meshed."""
    if n_meas == 2:
        ra = abs(K * (u[1] - u[0]))
    elif n_meas == 1:
        ra = abs(K * u[0])
    else:
        return float("nan")
    return ra / 2 if dim == 3 else ra
