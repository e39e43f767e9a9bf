#!/usr/bin/env python3
"""How evenly are the two per-axis QPs of one instance loaded?  (the costing of "two QPs per wavefront".)  Runs the bench's
pushed batches once and prints, from the work units the kernel reports per QP (ismpc_a_out.iters_x / iters_y): mean units per QP, the mean of
max(x, y) per instance and their ratio -- what two QPs iterating in lockstep in one wavefront would pay against two independent wavefronts.
usage (GPU box): python scripts/pair_imbalance.py [walk_C150 mc_C200 ...]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload

B = 16384
for wl in (sys.argv[1:] or ("walk_C150", "mc_C200", "walk_C100", "trot_C160")):
    for prec in ("f32", "f64"):
        if wl == "mc_C200":
            inst, push = workload.make_inst_mc(B)
            plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
            gen = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0], precision=prec); gen.add_plan(plans[1])
            d_inst = q.to_device(inst); d = q.to_device(gen.initial_state(0.88, batch=B))
            prep = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0]); prep.add_plan(plans[1])
            prep.rollout_inst_torch(d, d_inst, 60); torch.cuda.synchronize(); prep.close()
            o = gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0"))
        else:
            w = workload.make_batch_a(wl, B)
            g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
            gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
            d = q.to_device(w["state"])
            o = gen.tick_torch(d, torch.from_numpy(w["push"].copy()).to("cuda:0"))
        torch.cuda.synchronize()
        oo = q.from_device(o, FA.OUT_A)
        ix, iy = oo["iters_x"].astype(np.float64), oo["iters_y"].astype(np.float64)
        mean_qp = float((ix + iy).mean() / 2); mean_max = float(np.maximum(ix, iy).mean())
        print(json.dumps({"workload": wl, "prec": prec, "units_per_qp_mean": round(mean_qp, 3), "max_of_pair_mean": round(mean_max, 3),
                          "lockstep_over_independent": round(mean_max / mean_qp, 3), "corr_xy": round(float(np.corrcoef(ix, iy)[0, 1]), 3),
                          "units_p99": float(np.percentile(np.concatenate([ix, iy]), 99))}), flush=True)
        gen.close()
