#!/usr/bin/env python3
"""Run a tool of this repo against another build of the library (experiments with compile-time constants):
   python tools/run_with_lib.py remo3d_amd/libremo3d_hip_u10.so tools/ab_tune.py L 32 4 4"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import remo3d_amd._lib as lib
lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
