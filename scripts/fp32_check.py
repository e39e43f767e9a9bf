#!/usr/bin/env python3
"""Diagnostic (not a test): the fp32 QP solve of Formulation A against the fp64 one on the bench workloads and on closed loops.
usage: python scripts/fp32_check.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload

def stats(a, b, name):
    d = np.abs(a - b)
    return {name + "_max": float(d.max()), name + "_p99": float(np.percentile(d, 99)), name + "_med": float(np.median(d))}

B = 8192
for wl in ("walk_C100", "walk_C150", "trot_C160", "mc_C200"):
    res = {"workload": wl}
    gens = {}
    for prec in ("f64", "f32"):
        if wl == "mc_C200":
            inst, push = workload.make_inst_mc(B)
            plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
            gen = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0], precision=prec); gen.add_plan(plans[1])
            d_inst = q.to_device(inst)
            d = q.to_device(gen.initial_state(0.88, batch=B))
            if prec == "f64":
                gen.rollout_inst_torch(d, d_inst, 60); st0 = d.clone()
            d = st0.clone()
            o = q.from_device(gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0")), FA.OUT_A)
        else:
            w = workload.make_batch_a(wl, B)
            g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
            gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
            d = q.to_device(w["state"])
            o = q.from_device(gen.tick_torch(d, torch.from_numpy(w["push"].copy()).to("cuda:0")), FA.OUT_A)
        torch.cuda.synchronize()
        gens[prec] = (o, q.from_device(d, FA.STATE_A))
    (o64, s64), (o32, s32) = gens["f64"], gens["f32"]
    res["status_nonzero_f64"] = int((o64["status"] != 0).sum()); res["status_nonzero_f32"] = int((o32["status"] != 0).sum())
    res["status_differs"] = int((o64["status"] != o32["status"]).sum())
    ok = (o64["status"] == 0) & (o32["status"] == 0)
    res.update(stats(o64["u0"][ok], o32["u0"][ok], "u0")); res.update(stats(o64["f0"][ok], o32["f0"][ok], "f0"))
    com64 = np.stack([s64["x"], s64["y"]], 1)[ok]; com32 = np.stack([s32["x"], s32["y"]], 1)[ok]
    rel = np.abs(com64 - com32).max(1) / np.maximum(np.abs(com64).max(1), 1e-3)
    res["com_rel_max"] = float(rel.max())
    res.update(stats(np.stack([s64["xd"], s64["yd"]], 1)[ok], np.stack([s32["xd"], s32["yd"]], 1)[ok], "vel"))
    res["iters_f64"] = float((o64["iters_x"] + o64["iters_y"]).mean() / 2); res["iters_f32"] = float((o32["iters_x"] + o32["iters_y"]).mean() / 2)
    res["active_differs"] = int((o64["active"] != o32["active"]).sum())
    print(json.dumps(res), flush=True)

# closed loops against the MATLAB fixtures and against the fp64 rollout
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "formA_matlab_meta.json")))
for name in ("walk_phipi4", "trot_phipi4", "walk_phi0", "trot_phipi2"):
    m = META[name]; kind = FA.WALK if m["gait"] == "walk" else FA.TROT
    g = FA.default_gait(kind, m["phi"], m["disp_A"]); _, ce = FA.plan(g)
    z = np.load(os.path.join(GOLD, f"formA_matlab_{name}.npz"))
    tr = {}
    for prec in ("f64", "f32"):
        gen = FA.GaitGenerator(FA.default_params(kind), ce, precision=prec)
        st = q.to_device(gen.initial_state(g.disp_C, batch=2))
        tr[prec] = q.from_device(gen.rollout_torch(st, 2000), FA.OUT_A)[:, 0]
    com = z["com"][:2000, :2]
    print(json.dumps({"fixture": name, "status32_nonzero": int((tr["f32"]["status"] != 0).sum()),
                      "f64_vs_matlab": float(np.abs(tr["f64"]["com_before"] - com).max()), "f32_vs_matlab": float(np.abs(tr["f32"]["com_before"] - com).max()),
                      "f32_vs_f64_com": float(np.abs(tr["f32"]["com_before"] - tr["f64"]["com_before"]).max()),
                      "f32_vs_f64_com_first100": float(np.abs(tr["f32"]["com_before"][:100] - tr["f64"]["com_before"][:100]).max()),
                      "f32_vs_f64_vel": float(np.abs(tr["f32"]["vel_after"] - tr["f64"]["vel_after"]).max()),
                      "iters32": float((tr["f32"]["iters_x"] + tr["f32"]["iters_y"]).mean() / 2), "iters64": float((tr["f64"]["iters_x"] + tr["f64"]["iters_y"]).mean() / 2)}), flush=True)
