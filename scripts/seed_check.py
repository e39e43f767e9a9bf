"""Diagnostic: the bench workloads drawn from other random streams and with harder pushes -- work per QP (mean, worst, 99.9 %)
and flags, both precisions.  usage: python scripts/seed_check.py [workload ...]   (ISMPC_LIB=<-DISMPC_A_DIAG build> decodes the worst QP)"""
import sys, os, json, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload
B = 16384
for stream in (1, 2, 3):
    for wl in (sys.argv[1:] or ("walk_C150", "trot_C160", "walk_C100", "mc_C200")):
        for prec in ("f64", "f32"):
            if wl == "mc_C200":
                inst, push = workload.make_inst_mc(B, stream=stream)
                plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
                gen = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0], precision=prec); gen.add_plan(plans[1])
                d_inst = q.to_device(inst); d = q.to_device(gen.initial_state(0.88, batch=B)); gen.rollout_inst_torch(d, d_inst, 60)
                o = q.from_device(gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0")), FA.OUT_A)
            else:
                w = workload.make_batch_a(wl, B, stream=stream, push_scale=1.0 + 0.5 * stream)
                g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
                gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
                d = q.to_device(w["state"])
                o = q.from_device(gen.tick_torch(d, torch.from_numpy(w["push"].copy()).to("cuda:0")), FA.OUT_A)
            raw = np.concatenate([o["iters_x"], o["iters_y"]]).astype(np.int64) & 0xffffffff
            it = raw & 1023 if (raw >> 10).any() else raw
            if (raw >> 10).any():
                t = int(np.argmax(it)); r = int(raw[t])
                print("worst QP:", dict(work=int(it[t]), block_solves=(r >> 10) & 15, cold=(r >> 14) & 3, q0=(r >> 16) & 255, partial=(r >> 24) & 255, inst=t % B, axis=t // B), flush=True)
            print(json.dumps({"stream": stream, "workload": wl, "prec": prec, "mean": round(float(it.mean()), 2), "max": int(it.max()), "p999": float(np.percentile(it, 99.9)),
                              "status_nonzero": int((o["status"] != 0).sum()), "unverified": int(((o["status"] & FA.ST_UNVERIFIED) != 0).sum())}), flush=True)
            gen.close()
