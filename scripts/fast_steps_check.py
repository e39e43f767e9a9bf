#!/usr/bin/env python3
"""Round-4 check of the closed-form first / second exact steps of the Formulation A solver (csrc/ismpc_a_wave.hpp, ISMPC_A_FIRST_STEP /
ISMPC_A_SECOND_STEP) against a build without them (-DISMPC_A_FIRST_STEP=0 -DISMPC_A_SECOND_STEP=0): the bench workloads drawn from other random
streams and pushed 1.5x-2.5x harder, both precisions, 16 384 instances each.  Same flags instance by instance (a QP within rounding of the
feasibility boundary may flip: counted), and the returned u0 / f0 / next state agree to solver accuracy.
usage (GPU box):  python scripts/fast_steps_check.py run <out.npz>          (one library: $ISMPC_LIB or the in-tree one)
                  python scripts/fast_steps_check.py compare <variant.so>   (runs both in child processes and compares)"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = 16384
CASES = [(wl, prec, stream) for stream in (1, 2, 3) for wl in ("walk_C150", "trot_C160", "walk_C100", "mc_C200") for prec in ("f64", "f32")]


def run(path):
    import torch
    import quadruped_gait_generation_ismpc_amd as q
    from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload
    res = {}
    for wl, prec, stream in CASES:
        if wl == "mc_C200":
            inst, push = workload.make_inst_mc(B, stream=stream); push = push * (1.0 + 0.25 * stream)
            plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
            gen = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0], precision=prec); gen.add_plan(plans[1])
            prep = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0]); prep.add_plan(plans[1])
            d_inst = q.to_device(inst); d = q.to_device(gen.initial_state(0.88, batch=B)); prep.rollout_inst_torch(d, d_inst, 60); torch.cuda.synchronize(); prep.close()
            o = gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0"))
        else:
            w = workload.make_batch_a(wl, B, stream=stream, push_scale=1.0 + 0.5 * stream)
            g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
            gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
            d = q.to_device(w["state"])
            o = gen.tick_torch(d, torch.from_numpy(w["push"].copy()).to("cuda:0"))
        torch.cuda.synchronize()
        key = f"{wl}:{prec}:{stream}"
        res[key + ":out"] = q.from_device(o, FA.OUT_A).view(np.uint8).copy(); res[key + ":state"] = q.from_device(d, FA.STATE_A).view(np.uint8).copy()
        gen.close()
    np.savez(path, **res)


def compare(variant):
    from quadruped_gait_generation_ismpc_amd import formulation_a as FA
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "a.npz"), os.path.join(td, "b.npz")
        for path, lib in ((a, ""), (b, variant)):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "run", path], env=dict(os.environ, ISMPC_LIB=lib) if lib else {k: v for k, v in os.environ.items() if k != "ISMPC_LIB"})
        za, zb = np.load(a), np.load(b)
        worst = {}
        for wl, prec, stream in CASES:
            key = f"{wl}:{prec}:{stream}"
            oa, ob = za[key + ":out"].view(FA.OUT_A).reshape(-1), zb[key + ":out"].view(FA.OUT_A).reshape(-1)
            sa, sb = za[key + ":state"].view(FA.STATE_A).reshape(-1), zb[key + ":state"].view(FA.STATE_A).reshape(-1)
            flips = int((oa["status"] != ob["status"]).sum())
            ok = (oa["status"] == 0) & (ob["status"] == 0)
            du = float(np.abs(oa["u0"] - ob["u0"])[ok].max()); df = float(np.abs(oa["f0"] - ob["f0"])[ok].max())
            dx = float(max(np.abs(sa["x"] - sb["x"])[ok].max(), np.abs(sa["y"] - sb["y"])[ok].max()))
            print(json.dumps({"case": key, "status_nonzero": int((oa["status"] != 0).sum()), "flag_flips": flips, "du0": du, "df0": df, "dcom": dx,
                              "units_mean": [round(float((oa["iters_x"] + oa["iters_y"]).mean() / 2), 3), round(float((ob["iters_x"] + ob["iters_y"]).mean() / 2), 3)]}), flush=True)
            w = worst.setdefault(prec, dict(du0=0.0, df0=0.0, dcom=0.0, flips=0))
            w["du0"] = max(w["du0"], du); w["df0"] = max(w["df0"], df); w["dcom"] = max(w["dcom"], dx); w["flips"] += flips
        print(json.dumps({"worst": worst}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        compare(sys.argv[2])
