#!/usr/bin/env python3
"""Builds a parameter sweep of K sets (default 512) and prints what the table build took: the batched MFMA Newton-Schulz inverse is the
one dense contraction of the path.  Under rocprofv3 (scripts/profile_r03.sh) this gives the sweep_gemm kernel's
duration and MFMA counters at a size that fills the chip.  usage: python scripts/sweep_build_probe.py [K]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import workload
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ps = workload.make_sweep_params(K, N=100)
s = q.MPCSolver.sweep(q.reference_plan(params=ps[0]), ps)
info = s.sweep_info()
worst = max(max(s.sweep_verify_tables(k).values()) for k in (0, K // 2, K - 1))
flops = info["mfma_gemm_launches"] * 2.0 * 128 ** 3 * K
print(json.dumps(dict(info, mfma_flops=flops, build_tflops=flops / (info["build_ms"] * 1e-3) / 1e12, worst_rel_table_error_of_3_sets=worst)))
s.close()
