#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the Formulation A wave kernel (library built with -DISMPC_A_PROF).
usage: ISMPC_HIPCC_FLAGS=-DISMPC_A_PROF python -c 'from quadruped_gait_generation_ismpc_amd import build; build.build(force=True)'
       python scripts/prof_a.py walk_C100 4096"""
import ctypes as C, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quadruped_gait_generation_ismpc_amd import _lib
lib = _lib.load()
prof = (C.c_ulonglong * 32)()
lib.ismpc_a_debug_prof(prof, 1)
src = open(os.path.join(ROOT, "scripts", "bench_a.py")).read()
try:
    exec(compile(src, "bench_a.py", "exec"))
except SystemExit:
    pass
lib.ismpc_a_debug_prof(prof, 0)
names = ["search", "new row+neighbours+h", "small system", "(unused)", "comb/sv", "rho+ratio", "primal/dual step", "enter", "warm pass", "cold restarts", "QP setup", "QP solve (all of the above)", "QP verify + output"]
tot = prof[10] + prof[11] + prof[12]
for k in range(13):
    print(f"{names[k]:24s} {prof[k] / max(prof[16 + k], 1):9.0f} clk/visit  x{prof[16 + k]:9d}  {100.0 * prof[k] / tot:5.1f}%")
