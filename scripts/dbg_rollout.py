import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
N = 100; ticks = 600
z = np.load(os.path.join(ROOT, "tests/golden", f"preroll_N{N}.npz"))
tin = z["tick_in"].view(q.TICK_IN).reshape(-1); ref = z["tick_out"].view(q.TICK_OUT).reshape(-1)
p = q.default_params(N=N); s = q.MPCSolver(q.reference_plan(params=p), params=p)
st0 = tin[:1].copy(); st0["simulation_time"] = 0.0; st0["footstep_counter"] = 0; st0["mpc_iter"] = 0; st0["control_iter"] = 0
d = q.to_device(st0)
traj = s.rollout_torch(d, 0, ticks); torch.cuda.synchronize()
out = q.from_device(traj, q.TICK_OUT)[:, 0]
err = np.abs(out["com_pos"] - ref["com_pos"][:ticks]).max(1)
bad = np.where(err > 1e-9)[0]
print("first bad ticks", bad[:5])
t = bad[0]
for k in range(max(0, t - 2), t + 3):
    print(k, "fc", tin["footstep_counter"][k], "mpc", tin["mpc_iter"][k], "gpu", out["com_pos"][k], out["u0"][k], out["status"][k], hex(out["iters"][k]), "ref", ref["com_pos"][k], ref["u0"][k], ref["status"][k])
# single-tick on the recorded inputs around t
o1 = s.solve_batch(tin[max(0,t-2):t+3])
print("single-tick u0", o1["u0"], o1["status"])
