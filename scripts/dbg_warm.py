#!/usr/bin/env python3
"""warm-started vs cold-started Formulation A tick on a heavily pushed batch: where do they differ?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA
name = sys.argv[1] if len(sys.argv) > 1 else "walk_C100"
z = np.load(os.path.join(ROOT, "tests", "golden", f"prerollA_{name}.npz"))
tab = z["state"].view(FA.STATE_A).reshape(-1)
kind = int(z["gait"]); g = FA.default_gait(kind, float(z["phi"]), float(z["disp_A"]))
_, ce = FA.plan(g)
p = FA.default_params(kind, C=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
os.environ.pop("ISMPC_A_WARM", None)
warm = FA.GaitGenerator(p, ce)
os.environ["ISMPC_A_WARM"] = "0"
cold = FA.GaitGenerator(p, ce)
rng = np.random.default_rng(11)
B = 4096
st0 = tab[rng.integers(0, len(tab), B)].copy()
scale = rng.choice([1.0, 3.0, 10.0, 30.0], B, p=[0.55, 0.25, 0.15, 0.05])
push = np.stack([rng.uniform(-0.03, 0.03, B), rng.uniform(-0.05, 0.05, B)], 1) * scale[:, None]
d_push = torch.from_numpy(push.copy()).cuda()
sw, sc = q.to_device(st0), q.to_device(st0)
ow = q.from_device(warm.tick_torch(sw, d_push), FA.OUT_A)
oc = q.from_device(cold.tick_torch(sc, d_push), FA.OUT_A)
bad = np.nonzero(ow["status"] != oc["status"])[0]
print("status mismatches", len(bad), "of", B, " infeasible cold", int((oc["status"] != 0).sum()), "warm", int((ow["status"] != 0).sum()))
for i in bad[:12]:
    print(i, "scale", scale[i], "push", push[i], "warm", ow["status"][i], ow["iters_x"][i], ow["iters_y"][i], "cold", oc["status"][i], oc["iters_x"][i], oc["iters_y"][i],
          "u0", ow["u0"][i], oc["u0"][i])
ok = (ow["status"] == 0) & (oc["status"] == 0)
print("max |du0|", np.abs(ow["u0"][ok] - oc["u0"][ok]).max(), "max |df0|", np.abs(ow["f0"][ok] - oc["f0"][ok]).max())
print("iters warm", ow["iters_x"][ok].mean(), "cold", oc["iters_x"][ok].mean())
dif = np.nonzero(ok & (np.abs(ow["u0"] - oc["u0"]).max(1) > 1e-6))[0]
print("ok-ok differing", len(dif))
for i in dif[:8]:
    print(i, "scale", scale[i], "push", push[i], "iters", ow["iters_x"][i], ow["iters_y"][i], oc["iters_x"][i], oc["iters_y"][i], "u0", ow["u0"][i], oc["u0"][i], "act", hex(ow["active"][i]), hex(oc["active"][i]))
sel = np.concatenate([bad, dif])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"dbg_warm_{name}.npz"), idx=sel, j=st0["j"][sel], push=push[sel], warm_status=ow["status"][sel], cold_status=oc["status"][sel],
         warm_u0=ow["u0"][sel], cold_u0=oc["u0"][sel], warm_f0=ow["f0"][sel], cold_f0=oc["f0"][sel])
np.savez(os.path.join(ROOT, "gpurun_out", f"dbg_warm_all_{name}.npz"), j=st0["j"], push=push, warm_status=ow["status"], cold_status=oc["status"], warm_u0=ow["u0"], warm_f0=ow["f0"])
print("status histogram warm", dict(zip(*np.unique(ow["status"], return_counts=True))), "cold", dict(zip(*np.unique(oc["status"], return_counts=True))))
for i in (57,):
    print(i, "warm", ow["status"][i], ow["iters_x"][i], ow["iters_y"][i], hex(ow["active"][i]), "cold", oc["status"][i], oc["iters_x"][i], oc["iters_y"][i], hex(oc["active"][i]), "u0", ow["u0"][i], oc["u0"][i])
