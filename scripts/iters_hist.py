#!/usr/bin/env python3
"""Diagnostic: distribution of the work per QP (block passes + Goldfarb-Idnani steps) on the bench workloads."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload
B = 16384
for wl in ("walk_C150", "mc_C200", "walk_C100", "trot_C160"):
    for prec in ("f64", "f32"):
        if wl == "mc_C200":
            inst, push = workload.make_inst_mc(B)
            plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
            gen = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0], precision=prec); gen.add_plan(plans[1])
            d_inst = q.to_device(inst); d = q.to_device(gen.initial_state(0.88, batch=B)); gen.rollout_inst_torch(d, d_inst, 60)
            o = q.from_device(gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0")), FA.OUT_A)
        else:
            w = workload.make_batch_a(wl, B)
            g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
            gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
            d = q.to_device(w["state"])
            o = q.from_device(gen.tick_torch(d, torch.from_numpy(w["push"].copy()).to("cuda:0")), FA.OUT_A)
        it = np.concatenate([o["iters_x"], o["iters_y"]]); act = np.concatenate([o["active"] & 0xffff, o["active"] >> 16])
        edges = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 128, 10000]
        h, _ = np.histogram(it, bins=edges)
        big = it >= 32
        print(json.dumps({"workload": wl, "prec": prec, "mean": float(it.mean()), "hist_edges": edges[:-1], "hist": h.tolist(),
                          "share_of_work_in_ge32": float(it[big].sum() / it.sum()), "frac_ge32": float(big.mean()),
                          "active_mean_ge32": float(act[big].mean()) if big.any() else 0, "active_mean_lt32": float(act[~big].mean())}), flush=True)
