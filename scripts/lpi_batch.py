#!/usr/bin/env python3
"""Lanes per instance of the Formulation B lane-group kernels against the batch size: us per step of ismpc_solve_batch_device
(HIP events over 200 steps) for ISMPC_LPI = 8, 16 and 32 and for the one-instance-per-wavefront kernel.  usage: python scripts/lpi_batch.py [batch ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import workload
N = 100
p = q.default_params(N=N)
solvers = {}
for lpi in (8, 16, 32):
    os.environ["ISMPC_LPI"] = str(lpi)
    solvers[lpi] = q.MPCSolver(q.reference_plan(params=p), params=p, device=0)
os.environ["ISMPC_PATH"] = "wave"                          # one instance per wavefront (ismpc_tick_affine)
solvers[64] = q.MPCSolver(q.reference_plan(params=p), params=p, device=0)
for B in [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 16384]:
    d_in = q.to_device(workload.make_batch(N, B), "cuda:0"); d_out = torch.empty((B, 80), dtype=torch.uint8, device="cuda:0")
    res = {"batch": B}
    for lpi, s in solvers.items():
        for _ in range(20): s.solve_batch_torch(d_in, d_out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): s.solve_batch_torch(d_in, d_out)
        e1.record(); torch.cuda.synchronize()
        res[f"lpi{lpi}_us"] = round(1e3 * e0.elapsed_time(e1) / 200, 2)
    print(json.dumps(res), flush=True)
