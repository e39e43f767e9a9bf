#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the Formulation A wave kernel (library built with -DISMPC_A_PROF).
usage: python -c "from quadruped_gait_generation_ismpc_amd import build; build.build(out='build/variants/libismpc_prof.so', flags='-DISMPC_A_PROF')"
       ISMPC_LIB=build/variants/libismpc_prof.so python scripts/prof_a.py a_walk_C100"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import _lib
leg = sys.argv[1] if len(sys.argv) > 1 else "a_walk_C100"
name = {"a_walk_C100": "walk_C100", "config3_walk_C150": "walk_C150", "a_trot_C160": "trot_C160", "config4_mc_C200": "mc_C200"}[leg]
lib = _lib.load()
prof = (C.c_ulonglong * 32)()
R = bench.Ranks(1)
lib.ismpc_a_debug_prof(prof, 1)
bench.leg_a(R, q, leg, name, 4096, 3, 1, 1.0)
lib.ismpc_a_debug_prof(prof, 0)
names = ["search", "new row+neighbours+h", "small system", "(unused)", "comb/sv", "rho+ratio", "primal/dual step", "enter", "warm pass", "cold restarts", "QP setup", "QP solve (all of the above)", "QP verify + output"]
tot = prof[10] + prof[11] + prof[12]
for k in range(13):
    print(f"{names[k]:24s} {prof[k] / max(prof[16 + k], 1):9.0f} clk/visit  x{prof[16 + k]:9d}  {100.0 * prof[k] / tot:5.1f}%")
