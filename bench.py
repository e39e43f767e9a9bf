#!/usr/bin/env python3
"""bench.py -- ISMPC tick throughput on MI355X (see the contract in DESIGN.md section "Measurement").

One "step" = one pass of the hot path (one MPCSolver::solve tick, reference MPCSolver.cpp:204-430)
over one batch of synthetic instances already resident in HBM.  Per-GPU work is fixed (weak
scaling): each rank owns a contiguous shard of `--batch-per-gpu` instances (default 8 192 = the
shard of BASELINE config 3, 65 536 instances over 8 GPUs); with more than one rank a step ends with
the single RCCL all-gather of the 80-byte output records the north star prescribes.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD spec; MI355X_MICROARCH.md lists the FP32 157.3)
HORIZON = 100


def algorithmic_flops_per_tick(N):
    """SURVEY.md 8d, vertical Hessian factor shared by the batch: 6 N^2 + 20 N."""
    return 6.0 * N * N + 20.0 * N


def measured_traffic(N, B):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01/pmc_quad_b8192.json:
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 wide reads, + WRITE_SIZE), scaled by
    instances per launch; None when no profile of this kernel/horizon is committed."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_quad_b8192.json")
    if not os.path.exists(path) or os.environ.get("ISMPC_PATH") == "dense":
        return None
    j = json.load(open(path))
    if j.get("horizon") != N:
        return None
    return j["derived"]["hbm_bytes_per_launch"] * B / j["batch"]


def one_launch(N, B, cus):
    """csrc/ismpc_hip.hip launch(): batches whose wavefronts are all resident at once (<= 2 per SIMD) take the variant that
    runs the inequality fallback inside the same launch."""
    path = os.environ.get("ISMPC_PATH")
    return (path not in ("dense", "wave") and N <= 128 and os.environ.get("ISMPC_ONE_LAUNCH") != "0"
            and os.environ.get("ISMPC_Z_FALLBACK") != "0" and (B + 3) // 4 <= 8 * cus)


def kernel_name(N, B, cus):
    """The dominant kernel of the step, as rocprofv3 names it (csrc/ismpc_hip.hip launch())."""
    path = os.environ.get("ISMPC_PATH")
    if path == "dense":
        return "ismpc_tick_dense<%d, 16>" % ((N + 63) // 64)
    if path == "wave" or N > 128:
        return "ismpc_tick_affine<%d>" % ((N + 63) // 64)
    if one_launch(N, B, cus):
        return "ismpc_tick_quad_inline<%d, %d>" % ((N + 15) // 16, (N + 63) // 64)
    return "ismpc_tick_quad<%d>" % ((N + 15) // 16)        # four instances per wavefront


def cpu_baseline(N, tick_in, budget_s=20.0):
    """The reference's single-thread qpOASES path on this box's host cores: the CPU restatement of
    MPCSolver::solve with every QP solved by the reference's own vendored qpOASES (oracle/_ref),
    cold start per QP exactly like utils.cpp:121-130.  Bounded sample of the same workload."""
    from oracle import oracle as O
    kind = "reference" if O.have_ref() else "port"
    orc = O.Oracle(O.default_params(N), backend="ref" if kind == "reference" else "gi")
    orc.solve(tick_in[:8])                       # warm-up
    t0 = time.perf_counter(); orc.solve(tick_in[:64]); per = (time.perf_counter() - t0) / 64
    n = int(max(64, min(len(tick_in), budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    out, info = orc.solve(tick_in[:n])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "ticks/s", "cores": 1, "kind": kind,
            "sample": f"first {n} instances of the same batch, {dt:.1f} s, single thread, "
                      f"{'reference vendored qpOASES 3.2 (setToMPC, nWSR=300, cold init per QP)' if kind == 'reference' else 'oracle Goldfarb-Idnani'}",
            "ms_per_tick": 1e3 * dt / n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=8192)
    ap.add_argument("--horizon", type=int, default=HORIZON)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--also-config1", action="store_true",
                    help="additionally time BASELINE configs[1] (1 024 instances, one GPU) and report it under other_configs; "
                         "off by default so that a rocprofv3 summary of the default command holds launches of one size only")
    args = ap.parse_args()

    import torch
    import quadruped_gait_generation_ismpc_amd as q
    from quadruped_gait_generation_ismpc_amd import workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X: the hot path has no CPU fallback"
    # ISMPC_BENCH_REHEARSE=1: rehearsal of the multi-rank control flow on a ONE-GPU box -- every rank uses cuda:0 and the
    # all-gather runs over gloo on host copies.  Not a measurement (the JSON line says so); the driver never sets it.
    rehearse = world > 1 and os.environ.get("ISMPC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    from quadruped_gait_generation_ismpc_amd.distributed import shard_range, gather_records
    N, B = args.horizon, args.batch_per_gpu
    first, count = shard_range(world * B, rank, world)
    assert (first, count) == (rank * B, B)
    counts = [B] * world
    p = q.default_params(N=N)
    solver = q.MPCSolver(q.reference_plan(params=p), params=p, device=local_rank)
    tick_in = workload.make_batch(N, B, first_instance=rank * B)         # this rank's shard, no communication
    d_in = q.to_device(tick_in, dev)
    d_out = torch.empty((B, 80), dtype=torch.uint8, device=dev)
    d_all = torch.empty((world * B, 80), dtype=torch.uint8, device=("cpu" if rehearse else dev)) if world > 1 else None
    _gather = gather_records
    if rehearse:
        def gather_records(local, world_, out=None, counts=None):          # host copies over gloo
            return _gather(local.cpu(), world_, out=out, counts=counts)

    def step():
        solver.solve_batch_torch(d_in, d_out)
        if world > 1:
            gather_records(d_out, world, out=d_all, counts=counts)        # the one collective of the path

    for _ in range(args.warmup):
        step()
    # Kernel duration for the roofline: HIP events on the launch stream (torch's current stream IS the stream
    # the C ABI launches on).  An event pair around a ~10 us kernel reads several us high, so on one GPU the
    # pair brackets the whole timed region (K back-to-back launches, nothing else on the stream) and the
    # average launch interval is reported; with a collective in the loop each launch gets its own pair.
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if world == 1:
        ev_region[0].record()
        for k in range(args.steps):
            solver.solve_batch_torch(d_in, d_out)
        ev_region[1].record()
    else:
        for k in range(args.steps):
            ev[k][0].record()
            solver.solve_batch_torch(d_in, d_out)
            ev[k][1].record()
            gather_records(d_out, world, out=d_all, counts=counts)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if rehearse else dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_interval_ms = None
    if world == 1:
        # Small batches: the step is ONE launch (ismpc_tick_quad_inline) and the region above already times it.  Large
        # batches: two launches (ismpc_tick_quad, then the normally idle inequality fallback); the roofline then prices the
        # dominant kernel alone: same inputs, same kernel, a handle created with the fallback launch switched off
        # (ISMPC_Z_FALLBACK=0), K back-to-back launches bracketed by one event pair on the launch stream.
        step_interval_ms = ev_region[0].elapsed_time(ev_region[1]) / args.steps
        kernel_ms = step_interval_ms
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if one_launch(N, B, cus):
            pass                                          # the step IS one launch of the dominant kernel
        elif os.environ.get("ISMPC_PATH") != "dense" and os.environ.get("ISMPC_Z_FALLBACK") != "0":
            os.environ["ISMPC_Z_FALLBACK"] = "0"
            try:
                solo = q.MPCSolver(q.reference_plan(params=p), params=p, device=local_rank)
            finally:
                del os.environ["ISMPC_Z_FALLBACK"]
            d_tmp = torch.empty_like(d_out)
            for _ in range(args.warmup):
                solo.solve_batch_torch(d_in, d_tmp)
            torch.cuda.synchronize()
            ev_region[0].record()
            for _ in range(args.steps):
                solo.solve_batch_torch(d_in, d_tmp)
            ev_region[1].record()
            torch.cuda.synchronize()
            kernel_ms = ev_region[0].elapsed_time(ev_region[1]) / args.steps
    else:
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # BASELINE configs[1] (1 024 instances on one GPU) beside the headline shard: same kernels, same inputs (first 1 024)
    small = None
    if world == 1 and B > 1024 and args.also_config1:
        d_in_s, d_out_s = d_in[:1024].contiguous(), torch.empty((1024, 80), dtype=torch.uint8, device=dev)
        for _ in range(args.warmup):
            solver.solve_batch_torch(d_in_s, d_out_s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); ts = time.perf_counter(); e0.record()
        for _ in range(args.steps):
            solver.solve_batch_torch(d_in_s, d_out_s)
        e1.record(); torch.cuda.synchronize(); el_s = time.perf_counter() - ts
        k_ms = e0.elapsed_time(e1) / args.steps
        small = {"workload": "BASELINE configs[1]: 1 024 instances, N=%d, one GPU" % N, "value": 1024 * args.steps / el_s, "unit": "ticks/s",
                 "ms_per_step": 1e3 * el_s / args.steps, "kernel_ms": k_ms,
                 "roofline_frac": algorithmic_flops_per_tick(N) * 1024 / (k_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS}

    out = q.from_device(d_out, q.TICK_OUT)
    if world > 1:
        allout = q.from_device(d_all, q.TICK_OUT) if not rehearse else np.frombuffer(d_all.numpy().tobytes(), dtype=q.TICK_OUT)
        assert allout[rank * B:(rank + 1) * B].tobytes() == out.tobytes(), "all-gather misplaced this rank's shard"
    st = out["status"]
    frac_flight = float(((st & q.ST_FLIGHT) != 0).mean())
    frac_infeasible = float(((st & (q.ST_X_INFEASIBLE | q.ST_Y_INFEASIBLE)) != 0).mean())

    if rank == 0:
        ticks = world * B * args.steps
        value = ticks / elapsed
        flops = algorithmic_flops_per_tick(N) * B
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "ISMPC QP solves/s (batch, N=100 horizon)", "value": value,
            "unit": "ticks/s (1 tick = one MPCSolver::solve = 3 QPs: vertical + x + y)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a measurement)",
            "config": {"workload": f"Formulation B (MPCSolver::solve), trot plan Controller.cpp:89-97, N={N}, S=35, F=10, "
                                   f"{B} instances/GPU (shard of BASELINE config 3: 65 536 over 8 GPUs), nominal pre-roll + perturbation (SURVEY 8d)",
                       "horizon": N, "batch_per_gpu": B, "global_batch": world * B,
                       "collective": "one RCCL all-gather of 80-byte output records per step" if world > 1 else "none (1 GPU)",
                       "flight_fraction": frac_flight, "infeasible_fraction": frac_infeasible},
            "qp_solves_per_s": 3.0 * value,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_TFLOPS, "traffic": measured_traffic(N, B),
                         "kernel": kernel_name(N, B, torch.cuda.get_device_properties(dev).multi_processor_count), "kernel_ms": kernel_ms, "step_interval_ms": step_interval_ms,
                         "algorithmic_flops_per_launch": flops,
                         "note": "FP64 compute roofline (vector = matrix peak 78.6 TF); algorithmic flops 6N^2+20N per tick, "
                                 "shared vertical factor; algorithmic HBM bytes 152 B/tick are ~1e-4 of the HBM roofline; "
                                 "traffic = HBM bytes/launch from rocprofv3 PMC (profiles/r01/pmc_quad_b8192.json), "
                                 "measured at 8192 instances/launch and scaled linearly to this batch"},
        }
        if small is not None:
            line["other_configs"] = [small]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, tick_in, args.cpu_budget)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
