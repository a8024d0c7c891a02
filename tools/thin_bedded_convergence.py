import os, sys
import numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from remo3d_amd.model import Model
base = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Thin-bedded model")
tools = ["A0.4M6.0N", "A1.62M6.0N", "A4.0M0.5N", "A8.0M1.0N"]
depths = np.arange(0, 20.01, 2.5)
gold = np.loadtxt(os.path.join(base, "Logs", "Logs 1", "Results_1.txt"), skiprows=2)
rows = np.rint(depths/0.25).astype(int)
for label, kw in [("scale 1.0", dict()), ("scale 0.5", dict(mesh_scale=0.5)), ("R=100", dict(domain_radius=100)), ("rtol 1e-11", dict(rtol=1e-11, maxsteps=5000))]:
    m = Model.compute_synthetic_logs(tools, depths, os.path.join(base, "Formation", "Formation_model_1.txt"), os.path.join(base, "Borehole", "Borehole_model_correct_rm.txt"), gpu_workers=1, verbose=False, **kw)
    rel = np.array([(m.logs[t][:, 1] - gold[rows, 1 + i]) / gold[rows, 1 + i] for i, t in enumerate(tools)])
    print(label, "median |rel| per tool", np.round(np.median(np.abs(rel),axis=1),5), " signed mean", np.round(rel.mean(1),5))
    print("   A8.0M1.0N ours", np.round(m.logs["A8.0M1.0N"][:,1],4), "ref", gold[rows,4])
