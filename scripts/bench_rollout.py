#!/usr/bin/env python3
"""Closed loops on the device, beside bench.py: ticks/s of ismpc_rollout_device (Formulation B; the tick
loop runs inside one launch, ISMPC_ROLLOUT=host gives one launch per tick for A/B) and of ismpc_a_rollout*_device
(Formulation A).  Numbers go to DESIGN.md; never the bench `value`.
usage: python scripts/bench_rollout.py [b8192|b65536|walk_C100|walk_C150|trot_C160 ...]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import workload, formulation_a as FA

legs = sys.argv[1:] or ["b8192", "b65536", "walk_C100", "walk_C150", "trot_C160"]
for leg in legs:
    if leg.startswith("b"):
        B, N, T = int(leg[1:]), 100, 400
        p = q.default_params(N=N)
        solver = q.MPCSolver(q.reference_plan(params=p), params=p, device=0)
        st0 = np.repeat(np.zeros(1, dtype=q.TICK_IN), B); st0["com_pos"][:, 2] = 0.69
        rng = np.random.default_rng(1)
        st0["com_pos"][:, :2] += rng.uniform(-0.004, 0.004, (B, 2)); st0["com_vel"][:, :2] += rng.uniform(-0.02, 0.02, (B, 2))
        d = q.to_device(st0)
        solver.rollout_torch(d.clone(), 0, 20, want_traj=False); torch.cuda.synchronize()
        res = {"leg": leg, "ticks": T, "mode": os.environ.get("ISMPC_ROLLOUT", "kernel"), "lpi": os.environ.get("ISMPC_LPI", "16")}
        for want in (False, True):
            s2 = d.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); solver.rollout_torch(s2, 0, T, want_traj=want); torch.cuda.synchronize(); el = time.perf_counter() - t0
            key = "with_traj" if want else "no_traj"
            res[key + "_ticks_per_s"] = B * T / el; res[key + "_us_per_tick"] = 1e6 * el / T
        print(json.dumps(res), flush=True)
        solver.close()
    else:
        B, T = 16384, 50
        w = workload.make_batch_a(leg, B)
        g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
        gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce)
        d0 = q.to_device(w["state"]); d = d0.clone(); dpush = torch.from_numpy(w["push"].copy()).to("cuda:0")
        gen.rollout_torch(d, 3); torch.cuda.synchronize()
        d.copy_(d0); gen.tick_torch(d, dpush)                     # one pushed tick, then T ticks of closed loop
        t0 = time.perf_counter(); traj = gen.rollout_torch(d, T); torch.cuda.synchronize(); el = time.perf_counter() - t0
        tr = q.from_device(traj, FA.OUT_A)
        raw = np.stack([tr["iters_x"].ravel(), tr["iters_y"].ravel()]).astype(np.int64) & 0xffffffff
        if (raw >> 10).any():                                     # -DISMPC_A_DIAG build: statistics packed into the iteration fields
            it = raw & 1023; act = np.stack([tr["active"].ravel() & 0xffff, (tr["active"].ravel() >> 16) & 0xffff])
            for tick in (5, 20):
                sl = slice(tick * B, (tick + 1) * B)
                top = np.argsort(-it[:, sl].ravel())[:8]
                print("tick", tick, [dict(axis=int(t // B), inst=int(t % B), work=int(it[:, sl].ravel()[t]), block_solves=int(((raw[:, sl].ravel()[t]) >> 10) & 15),
                                          cold=int((raw[:, sl].ravel()[t] >> 14) & 3), q0=int((raw[:, sl].ravel()[t] >> 16) & 255), partial=int((raw[:, sl].ravel()[t] >> 24) & 255),
                                          active=int(act[:, sl].ravel()[t])) for t in top], flush=True)
            tr = tr.copy(); tr["iters_x"] = it[0].reshape(tr["iters_x"].shape); tr["iters_y"] = it[1].reshape(tr["iters_y"].shape)
        print(json.dumps({"leg": leg, "batch": B, "ticks": T, "ticks_per_s": B * T / el, "ms_per_tick": 1e3 * el / T,
                          "status_nonzero": int((tr["status"] != 0).sum()), "iters_mean": float((tr["iters_x"] + tr["iters_y"]).mean() / 2),
                          "iters_max_per_tick": [int(v) for v in np.maximum(tr["iters_x"], tr["iters_y"]).reshape(T, B).max(1)[:12]],
                          "iters_max_per_tick_mean": float(np.maximum(tr["iters_x"], tr["iters_y"]).reshape(T, B).max(1).mean())}), flush=True)
        gen.close()
