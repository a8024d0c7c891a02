#!/usr/bin/env python3
"""Mixed-precision behaviour on the small test meshes: steps / cycles / residuals per setting."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from remo3d_amd import solver  # noqa: E402
from remo3d_amd.meshgen import make_mesh  # noqa: E402

SRC = [([0.0], [1.0]), ([0.1], [1.0]), ([-0.1, 0.1], [1.0, -1.0])]
EVAL = [[0.4, 6.4, -2.0], [2.1, 2.6], [0.5, 3.0, 0.0]]
ctx = solver.Context(0)
for dim, scale in ((2, 2.0), (3, 8.0)):
    mesh = make_mesh(dim, 50.0, [0.0, 0.1, -0.1], scale=scale, material_fn=conftest._two_zone(dim), seed=0)
    for pre in ("local", "multigrid"):
        ref, st, rc = ctx.solve_batch(mesh, conftest.SIGMA3, SRC, EVAL, solver.make_opts(preconditioner=pre, rtol=1e-8))
        print(f"dim {dim} {pre:9s} fp64      rc {rc} its {st['iterations'][:3]} steps {st['pcg_steps']}")
        for digits in (2, 3, 4, 5, 6):
            outs, st, rc = ctx.solve_batch(mesh, conftest.SIGMA3, SRC, EVAL,
                                           solver.make_opts(preconditioner=pre, rtol=1e-8, precision="mixed", inner_digits=digits), raise_on_error=False) \
                if "raise_on_error" in solver.Context.solve_batch.__code__.co_varnames else \
                ctx.solve_batch(mesh, conftest.SIGMA3, SRC, EVAL, solver.make_opts(preconditioner=pre, rtol=1e-8, precision="mixed", inner_digits=digits))
            d = max(float(np.max(np.abs(g - r)) / np.max(np.abs(r))) for g, r in zip(outs, ref))
            print(f"dim {dim} {pre:9s} mixed d={digits} rc {rc} its {st['iterations'][:3]} steps {st['pcg_steps']} cycles {st['refinement_cycles']} "
                  f"relres {['%.1e' % v for v in st['relres'][:3]]} diff {d:.1e}")
