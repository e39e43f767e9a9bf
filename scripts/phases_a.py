#!/usr/bin/env python3
"""Where a Formulation A launch spends its wavefront time, by solver phase (-DISMPC_A_PHASES build: s_memtime deltas summed over
all wavefronts).  Build here:  python scripts/phases_a.py build      Run on the GPU box:  python scripts/phases_a.py [workload ...]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "build", "variants", "libismpc_phases.so")
if sys.argv[1:2] == ["build"]:
    from quadruped_gait_generation_ismpc_amd import build
    os.makedirs(os.path.dirname(VAR), exist_ok=True)
    build.build(out=VAR, flags="-DISMPC_A_PHASES", force=True); print(VAR); sys.exit(0)
os.environ["ISMPC_LIB"] = VAR
import numpy as np, torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload, _lib
NAMES = ["between QPs", "set-up", "row values", "small system", "bs links", "bs gram", "bs fold", "bs G+rhs", "bs defect", "bs slopes",
         "pass select", "gi search", "gi step", "final check", "outputs", "loop glue"]
B = 16384
lib = _lib.load()
for wl in (sys.argv[1:] or ("walk_C150", "mc_C200")):
    for prec in ("f32", "f64"):
        rl = 3 if wl == "walk_C150" or wl == "trot_C160" else (4 if wl == "mc_C200" else 2)
        dbg = getattr(lib, f"ismpc_a_debug_phases_rl{rl}"); dbg.argtypes = [C.c_void_p, C.c_int]; dbg.restype = C.c_int
        if wl == "mc_C200":
            inst, push = workload.make_inst_mc(B)
            plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
            gen = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0], precision=prec); gen.add_plan(plans[1])
            d_inst = q.to_device(inst); d = q.to_device(gen.initial_state(0.88, batch=B)); gen.rollout_inst_torch(d, d_inst, 60)
            tick = lambda: gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0"))
        else:
            w = workload.make_batch_a(wl, B)
            g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
            gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
            d = q.to_device(w["state"])
            dp = torch.from_numpy(w["push"].copy()).to("cuda:0")
            tick = lambda: gen.tick_torch(d, dp)
        torch.cuda.synchronize()
        assert dbg(None, 1) == 0
        d0 = d.clone(); o = tick(); torch.cuda.synchronize()
        ph = np.zeros(16, dtype=np.uint64); assert dbg(ph.ctypes.data_as(C.c_void_p), 1) == 0
        oo = q.from_device(o, FA.OUT_A)
        tot = float(ph.sum())
        print(json.dumps({"workload": wl, "prec": prec, "units_per_qp": float((oo["iters_x"] + oo["iters_y"]).mean() / 2),
                          "cycles_per_qp": tot / (2 * B), "share": {NAMES[k]: round(float(ph[k]) / tot, 4) for k in range(16)}}), flush=True)
        gen.close()
