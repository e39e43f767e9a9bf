#!/usr/bin/env python3
"""Formulation B beside the headline: (i) closed loop on the device (ismpc_rollout_device), (ii) the host-pointer entry
point ismpc_solve_batch (PCIe-inclusive).  Numbers go to DESIGN.md; never the bench `value`."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import workload
N = 100
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = q.default_params(N=N)
solver = q.MPCSolver(q.reference_plan(params=p), params=p, device=0)
tin = workload.make_batch(N, B)
res = {"batch": B}
# (i) closed loop: every instance from its (perturbed) state, T ticks on the device
T = 500
st = q.to_device(tin, "cuda:0")
solver.rollout_torch(st.clone(), int(tin["simulation_time"][0]) if "simulation_time" in tin.dtype.names else 0, 5, want_traj=False); torch.cuda.synchronize()
s2 = st.clone(); torch.cuda.synchronize()
t0 = time.perf_counter(); solver.rollout_torch(s2, 0, T, want_traj=False); torch.cuda.synchronize(); el = time.perf_counter() - t0
res["rollout_ticks_per_s"] = B * T / el; res["rollout_us_per_tick"] = 1e6 * el / T
# (ii) host buffers in and out (pageable numpy memory): H2D + kernel(s) + D2H per call
solver.solve_batch(tin[:64])
K = 20
t0 = time.perf_counter()
for _ in range(K):
    out = solver.solve_batch(tin)
el = time.perf_counter() - t0
res["host_path_ticks_per_s"] = B * K / el; res["host_path_ms_per_call"] = 1e3 * el / K
print(json.dumps(res))
