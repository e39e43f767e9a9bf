#!/usr/bin/env python3
"""Diagnostic: heavily pushed walk_C100 instances, GPU (fp64) flags vs the oracle's, details of the disagreements."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload
from oracle import oracle_a as A
B = 600
w = workload.make_batch_a("walk_C100", 1500, seed=77)
rng = np.random.default_rng(5)
push = (w["push"] * rng.choice([3.0, 10.0, 30.0], 1500, p=[0.4, 0.4, 0.2])[:, None])[:B]
st = w["state"][:B]
g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
res = {}
for prec in ("f64", "f32"):
    gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
    d = q.to_device(st)
    res[prec] = q.from_device(gen.tick_torch(d, torch.from_numpy(push.copy()).to("cuda:0")), FA.OUT_A)
sim = A.SimA(A.gait(w["kind"], w["phi"], w["disp_A"]), A.params(w["kind"], C_=w["C"], P=w["P"], F=w["F"]), backend="ref")
n = 0
for i in range(B):
    sim.load_product_state(st[i]); r = sim.tick(tuple(push[i]))
    o = res["f64"][i]; o32 = res["f32"][i]
    for ax in (0, 1):
        gi = bool(o["status"] & (1 << ax)); ri = r["rv"][ax] != 0
        if gi != ri and n < 25:
            n += 1
            print(f"inst {i} ax {ax}: gpu64 status {o['status']:#x} iters {(o['iters_x'], o['iters_y'])[ax]} active {(o['active'] >> (16 * ax)) & 0xffff} | "
                  f"gpu32 status {o32['status']:#x} active {(o32['active'] >> (16 * ax)) & 0xffff} | ref rv {r['rv'][ax]} nwsr {r['nwsr'][ax]} | push {push[i]} "
                  f"u0 {o['u0'][ax]:.6f} {o32['u0'][ax]:.6f} {r['u0'][ax]:.6f}")
print("mismatches shown", n)
del sim
