"""Logging-tool tables: the build's own restatement of remo3d.py:178-321.

A tool name such as "A2.0M0.5N" lists three electrodes from top to bottom (A/B current, M/N
measuring) and the two spacings in metres.  The table of a tool is the reference's 2x4 array

    [[z_1, z_2, z_3, K],            positions relative to the simulation depth (sorted ascending)
     [s_1, s_2, s_3, depth_shift]]  source terms (+1 / -1 / 0) and the shift of the record point

with K the geometric factor 4 pi AM AN / (AN - AM) (remo3d.py:285-306).  Pinned by the golden
vectors in tests/golden/tools.json, generated from the reference in this container.
"""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict, Sequence, Tuple

import numpy as np

_NAME = re.compile(r"^([A-Za-z]+)([0-9.eE+-]+)([A-Za-z]+)([0-9.eE+-]+)([A-Za-z]+)$")
_SWAP = str.maketrans("ABMN", "MNAB")


class ToolError(ValueError):
    pass


def _split(name: str):
    m = _NAME.match(name)
    if not m:
        raise ToolError(f"{name} logging tool specification is uncorrect")
    try:
        d = [float(m.group(2)), float(m.group(4))]
    except ValueError:
        raise ToolError(f"{name} logging tool specification is uncorrect")
    return (m.group(1), m.group(3), m.group(5)), d


def tool_table(name: str, force_single_electrode_configuration: bool = True) -> np.ndarray:
    """2x4 table of one tool (remo3d.py:231-321)."""
    spec = name
    if force_single_electrode_configuration and "A" in name and "B" in name:
        spec = name.translate(_SWAP)  # reciprocity: swap current and measuring roles (remo3d.py:211-214)
    electrodes, dist = _split(spec)
    if min(dist) <= 0 or sorted(electrodes) not in (sorted(p) for p in _PERMS):
        raise ToolError(f"{name} logging tool specification is uncorrect")
    if dist[0] == dist[1]:
        raise ToolError(f"{name} logging tool specification is uncorrect")
    # record point: centre of the closer pair (remo3d.py:259-264)
    z_mp = dist[0] / 2 if dist[0] < dist[1] else dist[0] + dist[1] / 2
    pos = np.array([0.0, dist[0], dist[0] + dist[1]])
    z = {e: pos[i] - z_mp for i, e in enumerate(electrodes)}
    if "A" not in z:      # B M N
        bm, bn = abs(z["B"] - z["M"]), abs(z["B"] - z["N"])
        k = abs(4 * np.pi * bm * bn / (bn - bm)); shift = z["B"]
        avail = np.array([z["B"], z["M"], z["N"]]); src = np.array([1, 0, 0])
    elif "B" not in z:    # A M N
        am, an = abs(z["A"] - z["M"]), abs(z["A"] - z["N"])
        k = abs(4 * np.pi * am * an / (an - am)); shift = z["A"]
        avail = np.array([z["A"], z["M"], z["N"]]); src = np.array([1, 0, 0])
    elif "M" not in z:    # A B N
        an, bn = abs(z["A"] - z["N"]), abs(z["B"] - z["N"])
        k = abs(4 * np.pi * an * bn / (an - bn)); shift = (z["A"] + z["B"]) / 2
        avail = np.array([z["A"], z["B"], z["N"]]); src = np.array([1, -1, 0])
    else:                 # A B M
        am, bm = abs(z["A"] - z["M"]), abs(z["B"] - z["M"])
        k = abs(4 * np.pi * am * bm / (bm - am)); shift = (z["A"] + z["B"]) / 2
        avail = np.array([z["A"], z["B"], z["M"]]); src = np.array([1, -1, 0])
    order = np.argsort(avail)
    table = np.zeros((2, 4))
    table[0, :3] = avail[order] - shift   # centred on the current electrode(s) (remo3d.py:319)
    table[1, :3] = src[order]
    table[0, 3] = k
    table[1, 3] = shift
    return table


_PERMS = [p for p in __import__("itertools").permutations("ABMN", 3)]


def tool_tables(tools: Sequence[str], force_single_electrode_configuration: bool = True) -> Tuple[Dict[str, np.ndarray], bool]:
    """Tables of all tools (insertion-ordered) and the single-electrode-computation flag
    (remo3d.py:178-228)."""
    if type(tools) != list or not all(isinstance(s, str) for s in tools):
        raise ValueError("Tools names have to be provided in the form of list of strings")
    if type(force_single_electrode_configuration) != bool:
        raise ValueError("The value of parameter force_single_electrode_configuration can be set only to True or False")
    out = OrderedDict()
    for name in tools:
        out[name] = tool_table(name, force_single_electrode_configuration)
    sec = all(not np.isclose(np.sum(t[1, :3]), 0) for t in out.values())
    return out, sec
