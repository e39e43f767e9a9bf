#!/usr/bin/env python3
"""Where a per-tick launch of the Formulation B lane-group kernel spends its time (diagnostic build with -DISMPC_STAMPS):
every wavefront stamps s_memrealtime (100 MHz) at 6 points; this prints, for one launch, the spread of wave start times
(dispatch ramp), the per-phase durations (median / p95 over wavefronts) and the tail (last wave end - median wave end).
usage: python -c "from quadruped_gait_generation_ismpc_amd import build; build.build(out='build/variants/libismpc_stamps.so', flags='-DISMPC_STAMPS')"
       ISMPC_LIB=build/variants/libismpc_stamps.so python scripts/stamps_b.py [batch ...]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import workload, _lib
lib = _lib.load()
lib.ismpc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
N = 100
p = q.default_params(N=N)
solver = q.MPCSolver(q.reference_plan(params=p), params=p, device=0)
lpi = 8 if os.environ.get("ISMPC_LPI") == "8" else 16
names = ["start->record", "record->tables+lambda+local products", "->scan+walk+reductions", "->knapsack", "->stores issued"]
for B in [int(a) for a in sys.argv[1:]] or [8192, 65536]:
    d_in = q.to_device(workload.make_batch(N, B), "cuda:0"); d_out = torch.empty((B, 80), dtype=torch.uint8, device="cuda:0")
    for _ in range(20):
        solver.solve_batch_torch(d_in, d_out)
    torch.cuda.synchronize()
    lib.ismpc_debug_stamps(None, 1)
    solver.solve_batch_torch(d_in, d_out); torch.cuda.synchronize()
    buf = np.zeros(16384 * 8, dtype=np.uint64)
    lib.ismpc_debug_stamps(buf.ctypes.data_as(C.c_void_p), 0)
    waves = min((B * lpi + 63) // 64, 16384)
    t = buf.reshape(16384, 8)[:waves, :6].astype(np.int64) * 10           # ns
    t0 = t[:, 0].min()
    res = {"batch": B, "waves_stamped": waves, "lpi": lpi,
           "wave_start_ns": {"p50": float(np.median(t[:, 0] - t0)), "p95": float(np.percentile(t[:, 0] - t0, 95)), "max": float((t[:, 0] - t0).max())},
           "wave_end_ns": {"p50": float(np.median(t[:, 5] - t0)), "p95": float(np.percentile(t[:, 5] - t0, 95)), "max": float((t[:, 5] - t0).max())},
           "wave_life_ns": {"p50": float(np.median(t[:, 5] - t[:, 0])), "p95": float(np.percentile(t[:, 5] - t[:, 0], 95)), "max": float((t[:, 5] - t[:, 0]).max())},
           "phases_ns": {names[k]: {"p50": float(np.median(t[:, k + 1] - t[:, k])), "p95": float(np.percentile(t[:, k + 1] - t[:, k], 95))} for k in range(5)}}
    print(json.dumps(res), flush=True)
