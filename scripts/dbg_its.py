#!/usr/bin/env python3
"""distribution of the knapsack Newton iteration counts (itx, ity) over the bench workload"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import workload
N = 100
p = q.default_params(N=N)
solver = q.MPCSolver(q.reference_plan(params=p), params=p, device=0)
tin = workload.make_batch(N, 8192)
out = q.from_device(solver.solve_batch_torch(q.to_device(tin, "cuda:0")), q.TICK_OUT)
print(out.dtype.names)
for name in out.dtype.names:
    if "it" in name:
        v = out[name]
        print(name, "mean", v.mean(), "hist", np.bincount(v.astype(np.int64).ravel())[:12])
print("status hist", dict(zip(*np.unique(out["status"], return_counts=True))))
it = out["iters"].astype(np.int64)
itx, ity = it & 255, (it >> 8) & 255
print("itx hist", np.bincount(itx)[:10], "ity hist", np.bincount(ity)[:10])
# lock-step cost: a wavefront iterates until its slowest row is done
m4 = np.maximum(itx, ity).reshape(-1, 4).max(1)
print("max over the 4 rows of a wavefront: hist", np.bincount(m4)[:10], "mean", m4.mean())
