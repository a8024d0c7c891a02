#!/usr/bin/env python3
"""Cost of a grid-wide barrier (remo_debug_grid_barrier) at several grid sizes: python tools/probe_barrier.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd import _lib, solver

L = _lib.load()
with solver.Context(0) as ctx:
    for nblocks in (64, 256, 512, 1024):
        out = (C.c_double * 3)()
        rc = L.remo_debug_grid_barrier(ctx._h, nblocks, 200, out)
        print("workgroups %4d: rc %d  %.2f us per barrier  gave up %d  stale loads %d" % (nblocks, rc, out[0], int(out[1]), int(out[2])), flush=True)
        if rc != 0 or out[1] != 0:
            break
