#!/usr/bin/env python3
"""Diagnostic: distribution of the work per QP (block passes + Goldfarb-Idnani steps) on the bench workloads.  With a
-DISMPC_A_DIAG build (ISMPC_LIB=build/variants/libismpc_diag.so) the iteration fields also carry: block solves, why a solve
started cold, the working-set size Goldfarb-Idnani took over with, and its partial steps."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload
B = 16384
for wl in (sys.argv[1:] or ("walk_C150", "mc_C200", "walk_C100", "trot_C160")):
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
        raw = np.concatenate([o["iters_x"], o["iters_y"]]).astype(np.int64) & 0xffffffff
        diag = bool((raw >> 10).any())
        it = raw & 1023 if diag else raw
        act = np.concatenate([o["active"] & 0xffff, o["active"] >> 16])
        edges = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 128, 10000]
        h, _ = np.histogram(it, bins=edges)
        big = it >= 32
        print(json.dumps({"workload": wl, "prec": prec, "mean": float(it.mean()), "hist_edges": edges[:-1], "hist": h.tolist(),
                          "share_of_work_in_ge32": float(it[big].sum() / it.sum()), "frac_ge32": float(big.mean()),
                          "active_mean_ge32": float(act[big].mean()) if big.any() else 0, "active_mean_lt32": float(act[~big].mean())}), flush=True)
        if diag:
            ns = (raw >> 10) & 15; cold = (raw >> 14) & 3; q0 = (raw >> 16) & 255; part = (raw >> 24) & 255
            gi = it - ns
            def grp(mask):
                return {"frac": float(mask.mean()), "work_share": float(it[mask].sum() / it.sum()), "gi_steps_mean": float(gi[mask].mean()) if mask.any() else 0,
                        "block_solves_mean": float(ns[mask].mean()) if mask.any() else 0, "q0_mean": float(q0[mask].mean()) if mask.any() else 0,
                        "final_active_mean": float(act[mask].mean()) if mask.any() else 0, "partial_mean": float(part[mask].mean()) if mask.any() else 0}
            top = np.argsort(-it)[:12]
            print(json.dumps({"workload": wl, "prec": prec, "top": [{"work": int(it[t]), "block_solves": int(ns[t]), "cold": int(cold[t]), "q0": int(q0[t]),
                                                                  "partial": int(part[t]), "final_active": int(act[t])} for t in top]}), flush=True)
            print(json.dumps({"workload": wl, "prec": prec, "warm_ok": grp(cold == 0), "cold_check": grp(cold == 1), "cold_budget": grp(cold == 2),
                              "warm_ok_ge32": grp((cold == 0) & big), "warm_ok_gi_hist": np.histogram(gi[cold == 0], bins=edges)[0].tolist()}), flush=True)
