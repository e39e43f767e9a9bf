"""A/B of the two Formulation-A kernels on the same pushed instances (both on the GPU)."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import quadruped_gait_generation_ismpc_amd as q
    from quadruped_gait_generation_ismpc_amd import formulation_a as FA
    name = sys.argv[2]; nb = int(sys.argv[3])
    z = np.load(os.path.join(ROOT, "tests", "golden", f"prerollA_{name}.npz"))
    tab = z["state"].view(FA.STATE_A).reshape(-1)
    kind = int(z["gait"]); g = FA.default_gait(kind, float(z["phi"]), float(z["disp_A"]))
    fp, ce = FA.plan(g); p = FA.default_params(kind, C=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    gen = FA.GaitGenerator(p, ce)
    rng = np.random.default_rng(5)
    jj = rng.integers(0, len(tab), nb); st0 = tab[jj].copy()
    push = np.stack([rng.uniform(-0.03, 0.03, nb), rng.uniform(-0.05, 0.05, nb)], 1)
    push[: nb // 2] = 0.0
    d = q.to_device(st0); dp = torch.from_numpy(push.copy()).cuda()
    out = q.from_device(gen.tick_torch(d, dp), FA.OUT_A); torch.cuda.synchronize()
    np.save(sys.argv[4], out)
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "walk_C100"; nb = 64
outs = {}
for kern in ("block", "wave"):
    env = dict(os.environ, ISMPC_A_KERNEL=kern)
    f = f"/tmp/dbg_a_{kern}.npy"
    subprocess.check_call([sys.executable, __file__, "child", name, str(nb), f], env=env)
    outs[kern] = np.load(f)
a, b = outs["block"], outs["wave"]
du = np.abs(a["u0"] - b["u0"]).max(1); df = np.abs(a["f0"] - b["f0"]).max(1)
print("status block", np.unique(a["status"]), "wave", np.unique(b["status"]))
print("max du0 %.3e  max df0 %.3e" % (du.max(), df.max()))
bad = np.where((du > 1e-8) | (df > 1e-9) | (a["status"] != b["status"]))[0]
print("mismatching instances:", len(bad), "of", nb)
for i in bad[:12]:
    print(i, "block its", a["iters_x"][i], a["iters_y"][i], "act", hex(a["active"][i]), "u0", a["u0"][i], "| wave its", b["iters_x"][i], b["iters_y"][i], "act", hex(b["active"][i]), "u0", b["u0"][i], "st", b["status"][i])
