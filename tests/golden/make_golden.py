"""Generates the golden vectors under tests/golden/ by importing the REFERENCE's pure-numpy host
bookkeeping (tool parser, task builder, model loaders, Gmsh-path windowing) in this container.

Run here only:   python tests/golden/make_golden.py
The reference's FEM path (NGSolve / Netgen / Gmsh / mpi4py) is not installed, so those modules are
replaced by inert stubs before import; nothing numerical is taken from them.  The outputs are
data (inputs + expected outputs); the GPU box never sees /root/reference.
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

for name in ["mpi4py", "mpi4py.MPI", "gmsh", "netgen", "netgen.meshing", "netgen.csg", "netgen.geom2d", "ngsolve",
             "matplotlib", "matplotlib.pyplot", "matplotlib.patches", "matplotlib.lines", "matplotlib.collections"]:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = MagicMock()
sys.path.insert(0, os.path.join(REF, "remo3d"))
sys.path.insert(0, REF)
import remo3d.remo3d as ref_main  # noqa: E402
import gmsh_functions as ref_gmf  # noqa: E402

EX = os.path.join(REF, "Examples")


def tolist(x):
    if isinstance(x, np.ndarray):
        return [tolist(v) for v in x.tolist()] if x.dtype == object else x.tolist()
    if isinstance(x, (list, tuple)):
        return [tolist(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


# ---- tools ------------------------------------------------------------------------------------
tool_sets = {
    "example_01": ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"],
    "thin_bedded": ["A0.4M6.0N", "A1.62M6.0N", "A4.0M0.5N", "A8.0M1.0N"],
    "bm3": ["A0.4M6.0N", "A2.0M0.5N"],
    "mixed": ["N2.5M0.25A", "B5.7A0.4M", "M0.5N2.0B", "A1.0B3.0N"],
}
tools_out = {}
for key, names in tool_sets.items():
    for force in (True, False):
        m = ref_main.Model(names, force_single_electrode_configuration=force)
        tools_out[f"{key}|{force}"] = dict(names=names, force=force, sec=bool(m.sec),
                                           tables={k: v.tolist() for k, v in m.tools.items()})
bad = []
for name in ["A2.0M2.0N", "A2.0M", "AM2.0N1.0", "A-1.0M0.5N", "A1.0A0.5N", "X1.0M0.5N"]:
    try:
        ref_main.Model([name])
        bad.append([name, False])
    except ValueError:
        bad.append([name, True])
    except Exception as e:  # any other failure type of the reference
        bad.append([name, type(e).__name__])
tools_out["rejects"] = bad
dump("tools.json", tools_out)


# ---- tasks ------------------------------------------------------------------------------------
def tasks_case(names, depths, batch_size, force=True):
    m = ref_main.Model(names, force_single_electrode_configuration=force)
    sim, tasks = m._prepare_simulation_depths_and_tasks(depths, batch_size)
    return dict(names=names, force=force, depths=depths.tolist(), batch_size=batch_size, sec=bool(m.sec),
                simulation_depths=sim.tolist(), tasks=tolist(tasks))


dump("tasks_example_01.json", tasks_case(tool_sets["example_01"], np.arange(0, 25.1, 0.1), 5))
dump("tasks_bm3.json", tasks_case(tool_sets["bm3"], np.linspace(5, 20, 100, endpoint=False), 5))
dump("tasks_bm1_single.json", tasks_case(["A0.4M6.0N"], np.linspace(5, 55, 100), 5))
dump("tasks_nonsec.json", tasks_case(["B5.7A0.4M", "A2.0M0.5N"], np.arange(2.0, 8.0, 0.2), 4, force=False))
dump("tasks_example_02.json", tasks_case(tool_sets["example_01"], np.arange(0, 25.1, 0.1), 10))
dump("tasks_thin_bedded.json", tasks_case(tool_sets["thin_bedded"], np.arange(0, 20.01, 0.25), 5))


# ---- model loading + windowing ------------------------------------------------------------------
def window_case(formation, borehole, dip_deg, depths, R, names):
    m = ref_main.Model(names)
    m.set_model_parameters(formation, borehole, dip=dip_deg)
    out = dict(formation_file=os.path.relpath(formation, EX), borehole_file=os.path.relpath(borehole, EX), dip_deg=dip_deg, R=R,
               formation_model=m.formation_model.tolist(), borehole_model_loaded=m.borehole_model.tolist())
    if m.dip_deg != 0:
        m.borehole_model = m._add_points_to_borehole()
    out["borehole_model"] = m.borehole_model.tolist()
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    mud = np.interp(depths, m.borehole_model[:, 0], m.borehole_model[:, 2])
    cases = []
    for d, rm in zip(depths, mud):
        fg, bh, sigma = ref_gmf.SelectGmshDataRange(bg, m.formation_model, m.dip_rad, rm, d, R)
        cases.append(dict(depth=float(d), rm=float(rm), formation_geometry=fg.tolist(), borehole_geometry=bh.tolist(), sigma=[float(s) for s in sigma]))
    out["cases"] = cases
    return out


bm = os.path.join(EX, "Benchmark models")
dump("windows_example_01.json", window_case(os.path.join(EX, "Example_01/Input/Formation.txt"), os.path.join(EX, "Example_01/Input/Borehole.txt"),
                                             0, np.array([0.5, 3.0, 8.3, 12.45, 20.0, 24.9]), 50.0, tool_sets["bm3"]))
dump("windows_example_01_r5.json", window_case(os.path.join(EX, "Example_01/Input/Formation.txt"), os.path.join(EX, "Example_01/Input/Borehole.txt"),
                                                0, np.array([0.5, 3.0, 8.3, 12.45, 20.0, 24.9]), 5.0, tool_sets["bm3"]))
dump("windows_bm2.json", window_case(os.path.join(bm, "Benchmark model 2/Formation_BM2.txt"), os.path.join(bm, "Benchmark model 2/Borehole_BM2.txt"),
                                     0, np.array([4.8, 15.0, 30.1, 55.0]), 50.0, tool_sets["bm3"]))
dump("windows_bm2_r8.json", window_case(os.path.join(bm, "Benchmark model 2/Formation_BM2.txt"), os.path.join(bm, "Benchmark model 2/Borehole_BM2.txt"),
                                        0, np.array([4.8, 15.0, 30.1, 55.0]), 8.0, tool_sets["bm3"]))
dump("windows_bm3_30.json", window_case(os.path.join(bm, "Benchmark model 3/Formation_BM3_30.txt"), os.path.join(bm, "Benchmark model 3/Borehole_BM3.txt"),
                                        30, np.array([2.75, 5.0, 12.5, 19.8]), 50.0, tool_sets["bm3"]))
dump("windows_bm3_60_r6.json", window_case(os.path.join(bm, "Benchmark model 3/Formation_BM3_60.txt"), os.path.join(bm, "Benchmark model 3/Borehole_BM3.txt"),
                                           60, np.array([2.75, 9.0, 12.5, 19.8]), 6.0, tool_sets["bm3"]))


# ---- Netgen-path windowing (2D default of the reference) -----------------------------------------
import netgen_functions as ref_ngf  # noqa: E402


def netgen_window_case(formation, borehole, depths, R, names):
    m = ref_main.Model(names)
    m.set_model_parameters(formation, borehole, dip=0)
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    mud = np.interp(depths, m.borehole_model[:, 0], m.borehole_model[:, 2])
    cases = []
    for d, rm in zip(depths, mud):
        fg, bh, sigma = ref_ngf.SelectNetgenDataRange(bg.copy(), m.formation_model.copy(), rm, d, R)
        cases.append(dict(depth=float(d), rm=float(rm), formation_geometry=fg.tolist(), borehole_geometry=bh.tolist(), sigma=[float(s) for s in sigma]))
    return dict(formation_file=os.path.relpath(formation, EX), borehole_file=os.path.relpath(borehole, EX), R=R, cases=cases)


dump("netgen_windows_example_01.json", netgen_window_case(os.path.join(EX, "Example_01/Input/Formation.txt"), os.path.join(EX, "Example_01/Input/Borehole.txt"),
                                                          np.array([0.5, 3.0, 8.3, 12.45, 20.0, 24.9]), 50.0, tool_sets["bm3"]))
dump("netgen_windows_example_01_r5.json", netgen_window_case(os.path.join(EX, "Example_01/Input/Formation.txt"), os.path.join(EX, "Example_01/Input/Borehole.txt"),
                                                             np.array([0.5, 3.0, 8.3, 12.45, 20.0, 24.9]), 5.0, tool_sets["bm3"]))
dump("netgen_windows_bm2.json", netgen_window_case(os.path.join(bm, "Benchmark model 2/Formation_BM2.txt"), os.path.join(bm, "Benchmark model 2/Borehole_BM2.txt"),
                                                   np.array([4.8, 15.0, 30.1, 55.0]), 50.0, tool_sets["bm3"]))
dump("netgen_windows_bm2_r8.json", netgen_window_case(os.path.join(bm, "Benchmark model 2/Formation_BM2.txt"), os.path.join(bm, "Benchmark model 2/Borehole_BM2.txt"),
                                                      np.array([4.8, 15.0, 30.1, 55.0]), 8.0, tool_sets["bm3"]))
tb = os.path.join(bm, "Thin-bedded model")
# the formation model ends 2.6 m beyond the first / last measurement depth: the windows of the long tools reach past it
dump("netgen_windows_thin_bedded.json", netgen_window_case(os.path.join(tb, "Formation/Formation_model_1.txt"), os.path.join(tb, "Borehole/Borehole_model_correct_rm.txt"),
                                                           np.array([-4.25, 0.0, 2.5, 10.0, 19.5, 24.0]), 50.0, tool_sets["thin_bedded"]))
