#!/usr/bin/env python3
"""Diagnostic (not a test): HIP path vs committed golden vectors, prints error tables."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import TICK_IN, TICK_OUT

G = os.path.join(ROOT, "tests", "golden")
for N in (50, 100, 150, 200):
    z = np.load(os.path.join(G, f"formB_vectors_N{N}.npz"))
    tin = z["tick_in"].view(TICK_IN).reshape(-1); ref = z["tick_out"].view(TICK_OUT).reshape(-1)
    p = q.default_params(N=N)
    s = q.MPCSolver(q.reference_plan(params=p), params=p)
    out = s.solve_batch(tin)
    ok = (ref["status"] & q.ST_ERROR_MASK) == 0
    cp = np.abs(out["com_pos"] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
    cv = np.abs(out["com_vel"] - ref["com_vel"]).max(1)
    du = np.abs(out["u0"] - ref["u0"])
    print(f"N={N}: status mismatch {(out['status'] != ref['status']).sum()}  relCoM max {cp[ok].max():.3e}  vel {cv[ok].max():.3e}  du0 {du[ok].max(0)}")
    bad = np.where(out["status"] != ref["status"])[0]
    for b in bad[:5]:
        print("   inst", b, "gpu", out["status"][b], "ref", ref["status"][b], "rv", z["rv"][b], "u0", out["u0"][b], ref["u0"][b])
    worst = np.argsort(-np.where(ok, cp, 0))[:3]
    for b in worst:
        print("   worst", b, cp[b], "gpu", out["com_pos"][b], out["u0"][b], "ref", ref["com_pos"][b], ref["u0"][b], "it", hex(out["iters"][b]), "nwsr", z["nwsr"][b])
