#!/usr/bin/env python3
"""bench.py -- ISMPC tick throughput on MI355X (contract: DESIGN.md section 5 "Measurement").

One "step" = one pass of the hot path over one batch of synthetic instances already resident in HBM.

Headline (the JSON line itself): Formulation B = MPCSolver::solve (reference MPCSolver.cpp:204-430), N = 100, fp64,
GLOBAL batch 65 536 (BASELINE configs[2]).  `--gpus N` shards that batch over N ranks (strong scaling: N = 1 runs all
65 536 instances on one GPU, N = 8 runs 8 192 per GPU); with more than one rank a step ends with the single RCCL
all-gather of the 80-byte output records, issued on a side stream so that it overlaps the next step's kernel.

`other_configs` (always emitted at N = 1): BASELINE configs[1] (1 024 instances, Formulation B), configs[3] (walking gait,
Formulation A, C = 150, 16 384 instances) and the per-GPU shape of configs[4] (Monte-Carlo trot/walk, C = 200, 16 384
instances per GPU), each with its own roofline and cpu_baseline.  At N > 1 configs[4] runs 16 384 instances on every
rank (131 072 at N = 8) with the same all-gather.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Timing: W warm-up steps, then regions of EXACTLY K steps each, bracketed by barrier + synchronize on both sides, wall time
= max over ranks; regions repeat until at least --min-region-ms of timed work exists (K = 20 steps of a 56 us kernel is
1 ms) and the MEDIAN region is reported.  Rank 0 prints ONE JSON line.

cpu_baseline: the CPU oracle with the reference's own vendored qpOASES (oracle/_ref) on the box's host cores, in worker
processes that never touch the GPU (`bench.py --cpu-worker ...`).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# VALU peaks the executed flops are priced against (MI355X_MICROARCH.md / AMD spec): FP64 vector = matrix 78.6 TFLOP/s (1024 SIMDs x
# 16 lanes x 2 flop x 2.4 GHz: one wave64 v_fma_f64 per 4 cycles), FP32 vector 157.3 TFLOP/s (one wave64 v_fma_f32 per 2 cycles).
# Measured on the box with scripts/micro/valu_peak.hip (profiles/r04/valu_peak.json): v_fma_f64 69.3, UNPACKED v_fma_f32 123.1,
# v_pk_fma_f32 146.0 TFLOP/s -- the fp32 solve issues unpacked instructions (the library is built -fno-slp-vectorize) and they do run
# at the two-cycle rate.  A kernel that executes both types is priced against the blend: peak = flops / (f64/78.6 + f32/157.3).
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}
PEAK_NOTE = ("FP64 vector (= matrix) 78.6 TFLOP/s, FP32 vector 157.3 TFLOP/s; a kernel executing both is priced against the blend flops / (f64 / 78.6 + "
             "f32 / 157.3).  Measured FMA rates on the box (profiles/r04/valu_peak.json): v_fma_f64 69.3, unpacked v_fma_f32 123.1, v_pk_fma_f32 146.0 TFLOP/s")
CLOCK_GHZ = 2.4              # MI355X_MICROARCH.md "Max clock"
SIMDS = 256 * 4
PROFILES = os.path.join(ROOT, "profiles", "r04")
LINE_LIMIT = 6000            # bytes of the ONE JSON line (the driver keeps an 8 KB tail of stdout: round 3's 27 KB line was never parsed)
DETAIL_NAME = "bench_detail.json"   # every leg's full object (prose notes, counters, host-path statistics), written next to bench.py
GLOBAL_BATCH = 65536
HORIZON = 100
A_BATCH = 16384


# ======================================================================================================================
# CPU worker (test infrastructure on the host cores; never imports torch, never touches the GPU)
# ======================================================================================================================
def cpu_worker(spec):
    """spec: dict(leg, n, offset, cpu).  Solves n units of the leg's workload with the oracle, prints {"n", "seconds"}."""
    if spec.get("cpu") is not None:
        try:
            os.sched_setaffinity(0, {int(spec["cpu"])})
        except OSError:
            pass
    leg, n, off = spec["leg"], int(spec["n"]), int(spec["offset"])
    from quadruped_gait_generation_ismpc_amd import workload
    if leg == "B":
        from oracle import oracle as O
        N = int(spec.get("horizon", HORIZON))
        tin = workload.make_batch(N, n, first_instance=off)
        orc = O.Oracle(O.default_params(N), backend=spec.get("backend") or ("ref" if O.have_ref() else "gi"))
        orc.solve(tin[:2])
        t0 = time.perf_counter(); orc.solve(tin); dt = time.perf_counter() - t0
        done = n
    elif leg.startswith("A:"):
        from oracle import oracle_a as A
        name = leg[2:]
        backend = spec.get("backend") or ("ref" if A.O.have_ref() else "gi")
        if name == "mc_C200":
            inst, push = workload.make_inst_mc(off + n)
            phi, dA = np.pi / 4, 0.1
            t0 = time.perf_counter(); done = 0
            for i in range(off, off + n):
                kind = A.TROT if inst["plan"][i] == 0 else A.WALK
                p = A.params(kind, C_=200, P=400, F=int(inst["F"][i]), step=int(inst["step"][i]), ds=int(inst["ds"][i]), Qf=float(inst["Qf"][i]))
                p.height = float(inst["height"][i])
                sim = A.SimA(A.gait(kind, phi, dA), p, backend=backend)
                sim.run(MC_PREROLL); sim.tick(tuple(push[i])); done += MC_PREROLL + 1
            dt = time.perf_counter() - t0
        else:
            w = workload.make_batch_a(name, off + n)
            sim = A.SimA(A.gait(w["kind"], w["phi"], w["disp_A"]), A.params(w["kind"], C_=w["C"], P=w["P"], F=w["F"]), backend=backend)
            t0 = time.perf_counter()
            for i in range(off, off + n):
                sim.load_product_state(w["state"][i]); sim.tick(tuple(w["push"][i]))
            dt = time.perf_counter() - t0; done = n
    else:
        raise SystemExit(f"unknown cpu leg {leg}")
    print(json.dumps({"n": done, "seconds": dt}), flush=True)


MC_PREROLL = 60      # nominal closed-loop ticks that spread the gait phases of the Monte-Carlo instances before the timed tick


def _spawn_worker(spec):
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps(spec)], cwd=ROOT,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, close_fds=True)


def _collect(procs, timeout):
    res = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill(); o, e = p.communicate()
        line = [l for l in o.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            raise RuntimeError("cpu worker failed: " + (e or "")[-500:])
        res.append(json.loads(line[-1]))
    return res


def cpu_info():
    model = "unknown"
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip(); break
    except OSError:
        pass
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    return model, os.cpu_count(), aff


def cpu_baseline(leg, unit, budget_s, sample_desc, unit_per_n=1, horizon=HORIZON):
    """The oracle (reference qpOASES backend where oracle/_ref exists) timed on this box's host cores: one pinned core
    (`value`, `cores` = 1) and, beside it, the embarrassingly parallel figure on all cores of this process's CPU share
    (`all_cores`).  Bounded samples of the same workload, about `budget_s` seconds each."""
    from oracle import oracle as O
    kind = "reference" if O.have_ref() else "port"
    model, nproc, aff = cpu_info()
    share = aff[:16]                                         # the GPU box grants 16 CPUs per GPU; never oversubscribe
    probe_n = 1 if leg == "A:mc_C200" else 24
    pr = _collect([_spawn_worker(dict(leg=leg, n=probe_n, offset=0, cpu=share[0], horizon=horizon))], 600)[0]
    per = pr["seconds"] / max(pr["n"], 1) * unit_per_n      # seconds per worker item (instance)
    n1 = int(max(probe_n, min(4096, budget_s / max(per, 1e-6))))
    one = _collect([_spawn_worker(dict(leg=leg, n=n1, offset=0, cpu=share[0], horizon=horizon))], 900)[0]
    single = one["n"] / one["seconds"]
    allc = _collect([_spawn_worker(dict(leg=leg, n=n1, offset=0, cpu=c, horizon=horizon)) for c in share], 1800)
    multi = sum(r["n"] / r["seconds"] for r in allc)
    qp = "reference vendored qpOASES 3.2 (setToMPC, nWSR=300, cold init per QP)" if kind == "reference" else "oracle Goldfarb-Idnani"
    own = None
    if kind == "reference":
        # the build's own CPU restatement (oracle + its Goldfarb-Idnani QP) beside (ii) the reference's qpOASES
        o = _collect([_spawn_worker(dict(leg=leg, n=n1, offset=0, cpu=share[0], horizon=horizon, backend="gi"))], 900)[0]
        own = {"value": o["n"] / o["seconds"], "cores": 1, "kind": "port", "sample": f"{o['n']} {unit.split('/')[0]} in {o['seconds']:.1f} s, oracle with its own dense Goldfarb-Idnani QP"}
    return {"value": single, "unit": unit, "cores": 1, "kind": kind, "own": own,
            "sample": f"{sample_desc}: {one['n']} {unit.split('/')[0]} in {one['seconds']:.1f} s on one pinned core, {qp}",
            "ms_per_unit": 1e3 / single,
            "all_cores": {"value": multi, "cores": len(share),
                          "note": f"one pinned worker process per core, same sample each, rates summed; {len(share)} of the box's {len(aff)} CPUs by policy "
                                  "(the GPU pool grants 16 CPUs per leased GPU): the embarrassingly parallel figure of a whole node is this rate per core x its cores"},
            "cpu_model": model, "nproc": nproc, "cpus_in_share": len(aff)}


# ======================================================================================================================
# GPU side
# ======================================================================================================================
def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_ranks(n, argv, timeout_s=3000):
    """`python bench.py --gpus N` as typed (no torch.distributed.run around it): THIS process touches no GPU -- it starts one
    rank process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment, the same
    variables torch.distributed.run sets), forwards rank 0's JSON line and exits non-zero if any rank does.  A rank that
    dies takes the others with it (they would wait in a collective for ever): exact PIDs only."""
    import tempfile
    port = _free_port()
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ISMPC_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = tempfile.TemporaryFile(mode="w+") if r == 0 else subprocess.DEVNULL
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT, stdout=out))
    t0, rc = time.time(), 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    print(f"bench.py: rank {r} exited with {c}; stopping the other ranks", file=sys.stderr, flush=True)
        if (rc != 0 or time.time() - t0 > timeout_s) and live:
            if rc == 0:
                rc = 124; print("bench.py: ranks timed out", file=sys.stderr, flush=True)
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(20)
                except subprocess.TimeoutExpired:
                    procs[r].kill(); procs[r].wait()
            live.clear()
        if live:
            time.sleep(0.2)
    outs[0].seek(0)
    text = outs[0].read()
    sys.stdout.write(text); sys.stdout.flush()
    if rc == 0 and not any(l.startswith("{") for l in text.splitlines()):
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr); rc = 1
    return rc


class Ranks:
    """The launcher's view: rank / world from the environment, the device, and the CONTROL plane (barrier, max over ranks, the 128-byte
    RCCL unique id).  collective = "abi" (default): the data plane is the native library's own RCCL communicator (include/ismpc_group.h,
    ismpc_group_create_rank) and torch.distributed runs over gloo for the control plane only; "torch": torch.distributed's nccl backend
    (= RCCL) carries both (round 3's path, kept for A/B)."""

    def __init__(self, gpus, force_collective=False, collective="abi"):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={self.world}: start `python bench.py --gpus N` (it spawns its own ranks) or "
                             "torch.distributed.run --nproc-per-node N bench.py --gpus N")
        assert torch.cuda.is_available(), "bench.py needs an MI355X: the hot path has no CPU fallback"
        # ISMPC_BENCH_REHEARSE=1: rehearsal of the multi-rank control flow on a ONE-GPU box -- every rank uses cuda:0 and the
        # all-gather runs over gloo on host copies.  Not a measurement (the JSON line says so); the driver never sets it.
        self.rehearse = self.world > 1 and os.environ.get("ISMPC_BENCH_REHEARSE") == "1"
        if self.rehearse:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dist = None
        # force_collective: a ONE-rank RCCL group, so that the path's collective (communicator set-up, the side-stream
        # all-gather and its event order) runs on a one-GPU box too -- over a group of one rank it moves no bytes between GPUs
        self.collective = self.world > 1 or force_collective
        self.abi = collective == "abi"
        self.ctl_cpu = self.rehearse or self.abi       # the control plane's tensors live on the host (gloo)
        self.data_group = None                         # process group of the torch data plane (None: the default one)
        self.abi_fallback = None                       # why the native group was given up, if it was
        if self.collective:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                os.environ["MASTER_PORT"] = str(_free_port())
            if self.rehearse or self.abi:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)   # "nccl" is RCCL on ROCm

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=("cpu" if self.ctl_cpu else self.dev))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def agree(self, ok):
        """True when EVERY rank says ok (one all-reduce over the control plane)."""
        if self.world == 1:
            return bool(ok)
        t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64, device=("cpu" if self.ctl_cpu else self.dev))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return t.item() > 0.5

    def native_group(self, make):
        """make(unique_id) -> the native group of this rank (include/ismpc_group.h).  The group has never run on more than one GPU
        (DESIGN 6.1): if building it fails on ANY rank, every rank gives it up and torch.distributed's nccl backend (= RCCL) carries
        the data plane instead -- the round-3 path, measured by the same legs; the line says which one ran (multi_gpu.path)."""
        if not (self.collective and self.abi) or self.rehearse:
            return None
        uid = self.unique_id()
        grp, why = None, None
        try:
            if os.environ.get("ISMPC_BENCH_FAIL_NATIVE_GROUP") == "1":
                raise RuntimeError("ISMPC_BENCH_FAIL_NATIVE_GROUP=1 (test of the fallback)")
            with stdout_to_stderr():
                grp = make(uid)
        except Exception as e:                           # noqa: BLE001 -- whatever it is, the other ranks must hear about it
            why = f"{type(e).__name__}: {e}"
        if self.agree(grp is not None):
            return grp
        if grp is not None:
            try:
                grp.close()
            except Exception:                            # noqa: BLE001
                pass
        self.abi = False
        self.abi_fallback = why or "the native group failed on another rank"
        print(f"bench.py rank {self.rank}: native RCCL group unavailable ({self.abi_fallback}); torch.distributed nccl carries the gather", file=sys.stderr)
        self.data_group = self.dist.new_group(backend="nccl")
        return None

    def unique_id(self):
        """The communicator's 128 bytes: made by rank 0 (ismpc_group_unique_id), handed to every rank over the control plane."""
        from quadruped_gait_generation_ismpc_amd import group as G
        box = [G.unique_id() if self.rank == 0 else None]
        if self.world > 1:
            self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed_regions(R, step, K, W, min_ms, drain=None, ev=None, prep=None):
    """W warm-up steps; then regions of exactly K steps, barrier + synchronize on both sides, max over ranks; regions repeat
    until min_ms of timed work (at least 3).  ev = (record_start, record_end) hooks on the launch stream, per region.
    prep() runs BEFORE a region's opening barrier (untimed): it puts the region's inputs in place (HBM-resident when the clock starts)."""
    torch = R.torch
    if prep:
        prep()
    for k in range(W):
        step(k)
    if drain:
        drain()
    walls, n = [], 0
    total = 0.0
    while True:
        if prep:
            prep()
        R.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        if ev:
            ev[0](n)
        for k in range(K):
            step(k)
        if ev:
            ev[1](n)
        if drain:
            drain()
        torch.cuda.synchronize(); R.barrier()
        el = R.max(time.perf_counter() - t0)
        walls.append(el); total += el; n += 1
        if (n >= 3 and total >= min_ms * 1e-3) or n >= 400:
            break
    return walls


class RegionEvents:
    """One HIP-event pair per region on torch's current stream -- which IS the stream the C ABI launches on."""
    def __init__(self, torch):
        self.torch, self.pairs = torch, []

    def start(self, n):
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        self.pairs.append((a, b)); a.record()

    def end(self, n):
        self.pairs[n][1].record()

    def median_ms(self, K):
        return statistics.median(a.elapsed_time(b) for a, b in self.pairs) / K


def load_pmc(leg):
    """profiles/r04/pmc_<leg>.json (scripts/profile_r04.sh + scripts/pmc_summary.py): counters of the dominant kernel,
    mean per launch, collected on exactly this leg's batch -- never scaled from another batch."""
    base = os.environ.get("ISMPC_PROFILES_DIR") or PROFILES
    path = os.path.join(base, f"pmc_{leg}.json")
    if not os.path.exists(path) and leg.startswith("headline_b"):        # the headline sharded over N ranks: the per-GPU shard's own pass
        path = os.path.join(base, f"pmc_shard_b{leg[len('headline_b'):]}.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))
    except ValueError:
        return None


_LIB_SHA = {}


def lib_sha256(path=None):
    """sha256 of the HIP library the process loads (quadruped_gait_generation_ismpc_amd/libismpc_hip.so, or $ISMPC_LIB): what ties a committed
    counter summary to the binary it was collected from (scripts/pmc_summary.py writes the same hash into every pmc_*.json; the hipcc
    build is reproducible: the same sources give the same bytes)."""
    import hashlib
    if path is None:
        from quadruped_gait_generation_ismpc_amd import _lib
        path = _lib.LIB_PATH
    if path not in _LIB_SHA:
        try:
            h = hashlib.sha256()
            with open(path, "rb") as f:
                for blk in iter(lambda: f.read(1 << 20), b""):
                    h.update(blk)
            _LIB_SHA[path] = h.hexdigest()
        except OSError:
            _LIB_SHA[path] = None
    return _LIB_SHA[path]


def lib_src_sha256(path=None):
    """sha256 of the SOURCES the loaded library was built from (quadruped_gait_generation_ismpc_amd/build.py::source_sha256, written next to
    the library at build time).  The library's own bytes embed the paths of its sources, so the same sources built in another directory
    have another lib_sha256; this value is path-independent."""
    if path is None:
        from quadruped_gait_generation_ismpc_amd import _lib
        path = _lib.LIB_PATH
    try:
        return open(path + ".src_sha256").read().strip() or None
    except OSError:
        return None


def roofline(leg, kernel, kernel_ms, batch, dtype, alg_flops, alg_bytes, extra=None, note=""):
    """The contract's roofline object for one leg.  `achieved` / `frac` are a MEASUREMENT of what the kernel executes: the
    floating-point wave-instructions the SQ counted for this kernel on this batch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_{F64,F32},
    committed under profiles/r04/) x 64 lanes, FMA = 2 flop, over the kernel's own launch duration (HIP events, live), against
    the VALU peak of the arithmetic type.  The SURVEY 8d figure (a dense solve the kernels do not run) is kept beside it as
    `algorithmic_credit` and carries no fraction.  `issue` is the time the vector pipes spend issuing ALL VALU instructions
    (4 cycles per wave64 instruction per SIMD) over the same duration -- also <= 1 by construction."""
    j = load_pmc(leg)
    peak = PEAK_TFLOPS[dtype]
    rf = {"bound": "valu", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None,
          "kernel": kernel, "kernel_ms": kernel_ms,
          "algorithmic_credit": {"flops_per_launch": alg_flops, "tflops": alg_flops / (kernel_ms * 1e-3) / 1e12, "bytes_per_launch": alg_bytes,
                                 "note": "SURVEY 8d per-tick figure x instances per launch / kernel_ms; not a fraction of anything the kernel executes"},
          "executed": None, "peak_note": PEAK_NOTE, "note": note}
    if extra:
        rf.update(extra)
    if not j or j.get("batch") != batch or j.get("kernel", "") not in kernel:
        rf["note"] += "  (no PMC summary for this leg / batch under profiles/r04: achieved, frac, traffic left null)"
        return rf
    sha, src = lib_sha256(), lib_src_sha256()
    rf["lib_sha256"] = sha; rf["pmc_lib_sha256"] = j.get("lib_sha256"); rf["pmc_git_head"] = j.get("git_head")
    rf["src_sha256"] = src; rf["pmc_src_sha256"] = j.get("src_sha256")
    same_binary = sha is not None and j.get("lib_sha256") == sha
    same_sources = src is not None and j.get("src_sha256") == src          # the same code built in another directory (paths are embedded in the binary)
    rf["pmc_matches_lib"] = bool(same_binary or same_sources)
    if not (same_binary or same_sources):
        # the counters were collected from another build of the library than the one loaded now: no executed-work figure is derived from them
        rf["note"] += "  (STALE PMC summary: neither its lib_sha256 nor its src_sha256 is the loaded library's; achieved, frac, traffic left null -- rerun scripts/profile_r04.sh)"
        return rf
    c, d = j.get("counters_mean_per_launch", {}), j.get("derived", {})
    per_step = float(j.get("launches_per_step", 1))
    fl64, fl32 = d.get("flops_f64_per_launch"), d.get("flops_f32_per_launch")
    ex = {"source": "profiles/r04/pmc_" + str(j.get("leg", leg)) + ".json"}
    if fl64 is not None and fl32 is not None and fl64 + fl32 > 0:
        flops = (fl64 + fl32) * per_step
        rf["achieved"] = flops / (kernel_ms * 1e-3) / 1e12
        rf["peak"] = peak = (fl64 + fl32) / (fl64 / PEAK_TFLOPS["f64"] + fl32 / PEAK_TFLOPS["f32"])      # blend by executed type
        rf["frac"] = rf["achieved"] / peak
        ex.update({"flops_f64_per_launch": fl64 * per_step, "flops_f32_per_launch": fl32 * per_step,
                   "fp_wave_instructions_per_launch": (d.get("fp_insts_f64_per_launch", 0.0) + d.get("fp_insts_f32_per_launch", 0.0)) * per_step,
                   "fp_share_of_valu_instructions": d.get("fp_share_of_valu_insts")})
    if "SQ_INSTS_VALU" in c:
        valu = c["SQ_INSTS_VALU"] * per_step
        issue_ms = valu * 4.0 / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
        ex.update({"valu_insts_per_launch": valu, "valu_issue_ms": issue_ms, "valu_issue_frac": min(1.0, issue_ms / kernel_ms),
                   "valu_issue_note": "VALU wave-instructions x 4 cycles / 1024 SIMDs / 2.4 GHz over kernel_ms: an upper estimate (32-bit instructions can issue every 2 cycles, and the chip clocks below 2.4 GHz under load)",
                   "mfma_f64_mops_per_launch": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64")})
        if "SQ_ACTIVE_INST_VALU" in c and c.get("SQ_BUSY_CYCLES"):
            # counters only (same pass, clock-independent): VALU-active quad-cycles x 4 per SIMD over the busy cycles of a shader engine (32 of them)
            ex["valu_busy_frac"] = min(1.0, c["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / (c["SQ_BUSY_CYCLES"] / 32.0))
    rf["executed"] = ex
    hbm = d.get("hbm_bytes_per_launch")
    rf["traffic"] = hbm * per_step if hbm is not None else None
    return rf


def isolated_ms(torch, fn, reps=40):
    """One launch alone: synchronize, event, launch, event, synchronize; median over reps.  This is what rocprofv3 reports per
    dispatch.  In a train of back-to-back launches (what `value` measures) the tail of launch k and the head of launch k+1
    overlap, so the per-launch interval of a train is SHORTER than an isolated launch -- the two numbers are reported apart."""
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def flops_b(N):
    """SURVEY.md 8d, vertical Hessian factor shared by the batch: 6 N^2 + 20 N per tick."""
    return 6.0 * N * N + 20.0 * N


def flops_a(C, F, w):
    """flops_A = 2 [nv w^2 + w^3/3 + 4 nv w + 6 nv], nv = C + F_A, w = measured mean working-set size."""
    nv = C + F
    return 2.0 * (nv * w * w + w ** 3 / 3.0 + 4.0 * nv * w + 6.0 * nv)


def _lpi(B):
    """csrc/ismpc_hip.hip pick_layout(): lanes per instance of the lane-group kernels (ISMPC_LPI, else 32 for batches <= 2 048, 16 up to
    8 192, 8 above)."""
    forced = {"8": 8, "16": 16, "32": 32}.get(os.environ.get("ISMPC_LPI"))
    return forced if forced else (32 if B <= 2048 else (16 if B <= 8192 else 8))


def _quad_r(N, lpi):
    """csrc/ismpc_hip.hip quad_R(): samples per lane of the lane-group kernels (smallest instantiated value that covers N)."""
    need = (N + lpi - 1) // lpi
    if lpi == 32:
        return 4
    if lpi == 16:
        return 4 if need <= 4 else (7 if need <= 7 else 8)
    return 8 if need <= 8 else (13 if need <= 13 else 16)


def one_launch(N, B, cus):
    """csrc/ismpc_hip.hip launch(): does a step of batch B consist of ONE kernel launch?"""
    path = os.environ.get("ISMPC_PATH")
    if path == "dense" or os.environ.get("ISMPC_Z_FALLBACK") == "0":
        return True
    mode = os.environ.get("ISMPC_ONE_LAUNCH", "2")                       # 2 (default): every batch size; 1: only batches resident at once; 0: never
    if path == "wave" or N > 128 or mode == "0":
        return False
    return mode != "1" or _resident(B, cus)


def _resident(B, cus):
    """Every wavefront of the batch on the chip at once at two wavefronts per SIMD (the latency kernel ismpc_tick_quad_inline)."""
    return (B * _lpi(B) + 63) // 64 <= 8 * cus


def kernel_name_b(N, B, cus, sweep=False, deferring=False):
    path = os.environ.get("ISMPC_PATH")
    if path == "dense":
        return "ismpc_tick_dense<%d, 16>" % ((N + 63) // 64)
    if path == "wave" or N > 128:
        return "ismpc_tick_affine<%d>" % ((N + 63) // 64)
    lpi = _lpi(B)
    slpi = 8 if (B > 8192 and os.environ.get("ISMPC_LPI") != "16") else 16  # sweep handles: 16 lanes per instance up to 8 192 instances, 8 beyond (ISMPC_LPI=16: 16 at every size)
    if one_launch(N, B, cus) and os.environ.get("ISMPC_Z_FALLBACK") != "0":
        # beyond the resident size the library goes back to two launches while instances are being deferred (launch() in csrc/ismpc_hip.hip)
        big_one = not deferring or os.environ.get("ISMPC_ONE_LAUNCH") == "3"
        if sweep and big_one:
            return "ismpc_tick_quad_one<%d, %d, %d, true>" % (_quad_r(N, slpi), slpi, (N + 63) // 64)
        if not sweep and _resident(B, cus):
            return "ismpc_tick_quad_inline<%d, %d, %d>" % (_quad_r(N, lpi), lpi, (N + 63) // 64)
        if not sweep and big_one:
            return "ismpc_tick_quad_one<%d, %d, %d, false>" % (_quad_r(N, lpi), lpi, (N + 63) // 64)
    if sweep:
        return "ismpc_tick_quad<%d, %d, true>" % (_quad_r(N, slpi), slpi)
    return "ismpc_tick_quad<%d, %d, false>" % (_quad_r(N, lpi), lpi)


def leg_b(R, q, leg, N, global_batch, K, W, min_ms, extras, sweep_sets=0):
    """Formulation B: `global_batch` instances sharded over the ranks; returns the result dict (rank 0) or None.
    sweep_sets > 0: a parameter sweep (ismpc_create_sweep) -- instance i runs with parameter set i % sweep_sets."""
    torch = R.torch
    from quadruped_gait_generation_ismpc_amd import workload
    from quadruped_gait_generation_ismpc_amd.distributed import shard_range, GatherPipeline
    world, rank = R.world, R.rank
    first, B = shard_range(global_batch, rank, world)
    assert global_batch % world == 0, "the bench shards the global batch evenly"
    p = q.default_params(N=N)
    pins = None
    if extras and world == 1:
        # the caller's page-locked record buffers of the host-path measurement: allocated first, as a service allocates its I/O buffers
        # at start-up (page-locked memory taken late in a long-lived process comes in scattered pages; the kernel's in-place PCIe
        # accesses then run at half the rate -- measured 0.39 ms against 0.19 ms per call of 65 536 records)
        pins = (q.PinnedRecords(B, q.TICK_IN), q.PinnedRecords(B, q.TICK_OUT))
    if sweep_sets > 0:
        solver = q.MPCSolver.sweep(q.reference_plan(params=p), workload.make_sweep_params(sweep_sets, N=N), device=R.local_rank)
    else:
        solver = q.MPCSolver(q.reference_plan(params=p), params=p, device=R.local_rank)
    tick_in = workload.make_batch(N, B, first_instance=first)            # this rank's shard, no communication
    if sweep_sets > 0:
        tick_in["reserved"] = (first + np.arange(B)) % sweep_sets
    d_in = q.to_device(tick_in, R.dev)
    sweep_sorted = sweep_sets > 0 and os.environ.get("ISMPC_BENCH_SWEEP_BIND", "1") != "0"
    if sweep_sorted:
        solver.sweep_bind(d_in)        # once: a sweep's instance -> set assignment is static, the launches then run sorted by set (ismpc_sweep_bind)
    d_out = [torch.empty((B, 80), dtype=torch.uint8, device=R.dev) for _ in range(2)]
    cus = torch.cuda.get_device_properties(R.dev).multi_processor_count
    pipe, grp = None, None
    if R.collective and R.abi and not R.rehearse and sweep_sets == 0:
        # the product's multi-GPU path: one native group per rank (its own handle, launch stream, side stream, RCCL communicator); the
        # shard's kernel writes into the gathered buffer in place and ONE ncclAllGather completes it on every GPU (csrc/ismpc_group.hip)
        from quadruped_gait_generation_ismpc_amd import group as G
        grp = R.native_group(lambda uid: G.Group.from_rank(q.reference_plan(params=p), p, R.local_rank, uid, rank, world))
    if grp is not None:
        assert grp.world == world and grp.shard(global_batch) == (first, B)
        grp.reserve(global_batch)
        torch.cuda.synchronize()                                         # d_in is complete before the group's own streams read it
    elif R.collective:
        pipe = GatherPipeline(world, B, 80, device=("cpu" if R.rehearse else R.dev), host_copies=R.rehearse, group=R.data_group)

    def step(k):
        if grp is not None:
            grp.step_device(global_batch, [d_in.data_ptr()], k & 1)
        elif pipe is None:
            solver.solve_batch_torch(d_in, d_out[0])
        else:
            b = pipe.before_launch(k)                                    # gather k-2 has left d_out[b]
            solver.solve_batch_torch(d_in, d_out[b])
            pipe.after_launch(k, d_out[b])                               # the one collective of the path, on the side stream

    ev = RegionEvents(torch)
    walls = timed_regions(R, step, K, W, min_ms, drain=(grp.sync if grp else (pipe.drain if pipe else None)), ev=(ev.start, ev.end))
    wall = statistics.median(walls)
    step_interval_ms = ev.median_ms(K) if grp is None else 1e3 * wall / K      # (the group launches on its own streams: events on torch's stream do not see them)
    if grp is not None:
        solver.solve_batch_torch(d_in, d_out[0])                         # this rank's shard through the plain handle: what the gathered buffer must hold at [first, first + B)

    # ---- dominant kernel alone (roofline): same inputs.  Two durations: a train of K back-to-back launches (per-launch
    # interval, event pair per region) and ONE launch between synchronisations (what rocprofv3 reports per dispatch)
    solo = solver
    deferring = bool(((q.from_device(d_out[0], q.TICK_OUT)["status"] & q.ST_Z_INEQ_ACTIVE) != 0).any())
    two_launches = not one_launch(N, B, cus) or (deferring and not _resident(B, cus) and os.environ.get("ISMPC_ONE_LAUNCH") != "3")
    if two_launches:
        os.environ["ISMPC_Z_FALLBACK"] = "0"                             # the tick kernel alone: the second launch (fallback) switched off
        try:
            solo = (q.MPCSolver.sweep(q.reference_plan(params=p), workload.make_sweep_params(sweep_sets, N=N), device=R.local_rank) if sweep_sets > 0
                    else q.MPCSolver(q.reference_plan(params=p), params=p, device=R.local_rank))
        finally:
            del os.environ["ISMPC_Z_FALLBACK"]
        if sweep_sorted:
            solo.sweep_bind(d_in)
    d_tmp = torch.empty_like(d_out[0])
    ev2 = RegionEvents(torch)
    timed_regions(R, lambda k: solo.solve_batch_torch(d_in, d_tmp), K, W, min_ms, ev=(ev2.start, ev2.end))
    kernel_ms_train = ev2.median_ms(K)
    kernel_ms = isolated_ms(torch, lambda: solo.solve_batch_torch(d_in, d_tmp))
    collective_ms = None
    if grp is not None:
        # the collective alone: steps whose kernels are already done do not exist in this API, so time a train of full steps against the
        # kernel train measured above -- the difference is what the all-gather adds when it does NOT hide behind the next kernel
        collective_ms = max(0.0, 1e3 * wall / K - kernel_ms_train)
    if pipe is not None:
        ev3 = RegionEvents(torch)
        timed_regions(R, lambda k: pipe.gather_blocking(d_out[0]), K, W, min_ms, ev=(ev3.start, ev3.end))
        collective_ms = ev3.median_ms(K) if not R.rehearse else None

    out = q.from_device(d_out[0], q.TICK_OUT)
    if grp is not None:
        allout = q.from_device(grp.result_torch(global_batch, 0, (K - 1) & 1).clone(), q.TICK_OUT)
        assert allout[first:first + B].tobytes() == out.tobytes(), "group all-gather misplaced or altered this rank's shard"
    if pipe is not None:
        allout = pipe.result_numpy((K - 1) & 1, q.TICK_OUT)
        assert allout[rank * B:(rank + 1) * B].tobytes() == q.from_device(d_out[(K - 1) & 1], q.TICK_OUT).tobytes(), "all-gather misplaced this rank's shard"
    st = out["status"]
    stage3 = (st & (q.ST_FLIGHT | q.ST_BAD_INDEX | q.ST_TICK_SKIPPED)) == 0
    itx, ity = out["iters"] & 255, (out["iters"] >> 8) & 255
    nqp = 2 * int(stage3.sum())
    active_box = (int(((itx >= 2) & stage3).sum()) + int(((ity >= 2) & stage3).sum())) / max(nqp, 1)

    res = None
    if rank == 0:
        value = global_batch / (wall / K)
        flops = flops_b(N) * B
        kname = kernel_name_b(N, B, cus, sweep=sweep_sets > 0, deferring=deferring)
        res = {
            "value": value, "unit": "ticks/s (1 tick = one MPCSolver::solve = 3 QPs: vertical + x + y)",
            "ms_per_step": 1e3 * wall / K, "dtype": "f64", "qp_solves_per_s": 3.0 * value,
            "regions": len(walls), "region_ms": {"median": 1e3 * wall, "min": 1e3 * min(walls), "max": 1e3 * max(walls)},
            "config": {"workload": f"Formulation B (MPCSolver::solve), trot plan Controller.cpp:89-97, N={N}, S=35, F=10, fp64, "
                                   f"global batch {global_batch} ({B} instances/GPU), nominal pre-roll + perturbation (SURVEY 8d)",
                       "horizon": N, "global_batch": global_batch, "batch_per_gpu": B,
                       "collective": (f"one RCCL all-gather of 80-byte output records per step over {world} rank(s), side stream, double-buffered "
                                      "(overlaps the next step's kernel); " + ("issued by the native library (ismpc_group_step_device, ncclAllGather in place)" if grp is not None
                                                                               else "issued through torch.distributed")) if (pipe is not None or grp is not None) else "none (1 GPU)",
                       "flight_fraction": float(((st & q.ST_FLIGHT) != 0).mean()),
                       "infeasible_fraction": float(((st & (q.ST_X_INFEASIBLE | q.ST_Y_INFEASIBLE)) != 0).mean()),
                       "z_inequality_active_fraction": float(((st & q.ST_Z_INEQ_ACTIVE) != 0).mean()),
                       "active_box_fraction": active_box},
            "roofline": roofline(leg, kname, kernel_ms, B, "f64", flops, 152.0 * B,
                                 extra={"kernel_ms_train": kernel_ms_train, "step_interval_ms": step_interval_ms},
                                 note="kernel_ms = ONE launch between synchronisations (HIP events; agrees with the rocprofv3 per-dispatch average under "
                                      "profiles/r04); kernel_ms_train = per-launch interval of back-to-back launches of the same kernel (consecutive "
                                      "launches overlap head to tail, so it is shorter and is what `value` is made of); step_interval_ms = the same for the "
                                      "whole step (one launch while no instance has active vertical inequality rows; a batch beyond the resident size that "
                                      "does defer instances takes two: the tick kernel, then one wavefront per deferred instance).  No MFMA on this "
                                      "path: bound = FP64 vector issue."),
        }
        if grp is not None:
            from quadruped_gait_generation_ismpc_amd import group as G
            res["multi_gpu"] = {"rccl_world": grp.world, "rccl_version": G.rccl_version(), "path": "abi", "kernel_ms": kernel_ms_train,
                                "exposed_collective_ms": collective_ms, "overlapped_step_ms": 1e3 * wall / K}
        if pipe is not None:
            res["multi_gpu"] = {"rccl_world": (R.dist.get_world_size(R.data_group) if not R.rehearse else None), "path": "torch", "kernel_ms": kernel_ms_train,
                                "collective_ms": collective_ms, "overlapped_step_ms": 1e3 * wall / K}
            if R.abi_fallback:
                res["multi_gpu"]["native_group_failed"] = R.abi_fallback[:48]
        if extras and world == 1:
            # PCIe-inclusive rate through the host-pointer entry point (H2D of the inputs and D2H of the outputs
            # inside the metric) -- never `value`.  Page-locked caller buffers: zero copy (the kernel reads and writes them in place
            # over PCIe); pageable buffers: staged through device memory
            host_stats = []

            def host_rate(tin_h, out_h):
                solver.solve_batch(tin_h[:64])
                solver.solve_batch(tin_h, out=out_h)
                reps = max(5, int(0.05 / max(1e-4, B * 2.5e-9 + 1e-4)))
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter(); solver.solve_batch(tin_h, out=out_h); ts.append(time.perf_counter() - t0)
                if os.environ.get("ISMPC_BENCH_DEBUG"):
                    print("host_rate", len(ts), "median", 1e3 * statistics.median(ts), "mean", 1e3 * sum(ts) / len(ts), "first", [round(1e3 * x, 3) for x in ts[:6]], file=sys.stderr)
                el = statistics.median(ts)                       # the median call: single calls of the in-place path stall for milliseconds now and
                srt = sorted(ts)                                  # then (host memory system of a shared box); mean and p90 are reported beside it
                host_stats.append({"calls": len(ts), "median_ms": 1e3 * el, "mean_ms": 1e3 * sum(ts) / len(ts), "p90_ms": 1e3 * srt[int(0.9 * len(srt))], "max_ms": 1e3 * srt[-1]})
                return B / el, 1e3 * el
            pin_in, pin_out = pins
            pin_in.array[:] = tick_in
            res["value_incl_pcie"], res["ms_per_step_incl_pcie"] = host_rate(pin_in.array, pin_out.array)
            assert pin_out.array.tobytes() == out.tobytes(), "zero-copy host path differs from the device-pointer path"
            res["value_incl_pcie_pageable"], res["ms_per_step_incl_pcie_pageable"] = host_rate(tick_in, np.zeros(B, dtype=q.TICK_OUT))
            res["incl_pcie_calls"] = {"page_locked": host_stats[0], "pageable": host_stats[1]}
            res["incl_pcie_note"] = ("ismpc_solve_batch, host records in and out: value_incl_pcie with page-locked caller buffers (ismpc_host_alloc: zero copy, "
                                     "the kernel reads and writes the caller's records in place over PCIe; records bit-identical to the device path), "
                                     "value_incl_pcie_pageable with pageable numpy buffers (H2D -> kernel -> D2H through device staging)")
            pin_in.free(); pin_out.free()
            # batch of ONE through ismpc_solve_batch = the body of MPCSolver::solve in include/MPCSolver.hpp (Controller.cpp:346-348 shape)
            import ctypes as C
            one_in = np.ascontiguousarray(tick_in[:1]); one_out = np.zeros(1, dtype=q.TICK_OUT)
            pi, po = one_in.ctypes.data_as(C.c_void_p), one_out.ctypes.data_as(C.c_void_p)
            lat = []
            for k in range(320):
                t0 = time.perf_counter()
                rc = solver._lib.ismpc_solve_batch(solver._h, 1, pi, po)
                lat.append(time.perf_counter() - t0)
                assert rc == 0
            lat = sorted(lat[20:])
            res["latency_batch1_us"] = 1e6 * lat[len(lat) // 2]
            res["latency_batch1_us_p99"] = 1e6 * lat[int(0.99 * len(lat))]
    if res is not None and sweep_sets > 0:
        info = solver.sweep_info()
        ng = (N + 63) // 64 * 64
        flops = info["mfma_gemm_launches"] * 2.0 * ng ** 3 * sweep_sets
        res["config"]["workload"] = res["config"]["workload"].replace("Formulation B (MPCSolver::solve)", f"Formulation B PARAMETER SWEEP ({sweep_sets} parameter sets: mass, h_des, q_p, q_u, q_v, foot width; instance i -> set i % {sweep_sets})")
        res["config"]["sorted_by_set"] = bool(sweep_sorted)
        res["sweep"] = dict(info, mfma_flops=flops, build_tflops=flops / (info["build_ms"] * 1e-3) / 1e12 if info["build_ms"] > 0 else None,
                            note="tables of every set built on the device: Newton-Schulz inverse of the vertical Hessians as batched v_mfma_f64_16x16x4_f64 products "
                                 "(csrc/ismpc_sweep.hip); build_ms = the whole build (all kernels), build_tflops = the MFMA products' flops over it")
    if grp is not None:
        grp.close()
    solver.close()
    return res


class stdout_to_stderr:
    """RCCL prints a version banner on file descriptor 1 when a process creates its first communicator; rank 0's stdout carries the ONE
    JSON line and nothing else, so communicators are created with fd 1 pointing at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1); os.close(self.saved)
        return False


def leg_group_one(R, q, N, batch, K, W, min_ms):
    """At --gpus 1: the multi-GPU entry points of the C ABI on the one GPU there is -- a group of one device (ncclCommInitAll), the
    double-buffered ismpc_group_step_device with its in-place ncclAllGather on the side stream.  Reports what RCCL itself says about the
    communicator (ncclCommCount) and the step time of the pipelined path; NOT a scaling measurement."""
    torch = R.torch
    from quadruped_gait_generation_ismpc_amd import group as G, workload
    p = q.default_params(N=N)
    with stdout_to_stderr():
        g = G.Group(q.reference_plan(params=p), p, devices=[R.local_rank])
    d_in = q.to_device(workload.make_batch(N, batch), R.dev)
    g.reserve(batch); torch.cuda.synchronize()
    walls = timed_regions(R, lambda k: g.step_device(batch, [d_in.data_ptr()], k & 1), K, W, min_ms, drain=g.sync)
    wall = statistics.median(walls)
    res = {"rccl_world": g.world, "rccl_version": G.rccl_version(), "path": "abi", "mode": "one process, ncclCommInitAll over 1 device (no scaling curve measured)",
           "group_step_ms": 1e3 * wall / K, "group_value": batch / (wall / K)}
    g.close()
    return res


def leg_sustained(R, q, N, batch, seconds, ticks=2000):
    """SURVEY 8 row f1 as a throughput figure, and the one leg long enough for an outside utilisation sampler to see: the closed loop
    on the device (ismpc_rollout_device: Controller.cpp:297-310,346-348,503-504 around solve(), the tick loop inside one launch),
    `ticks` ticks x `batch` instances per call from the same perturbed start, calls back to back for at least `seconds` of GPU time."""
    torch = R.torch
    p = q.default_params(N=N)
    rows = 8 + (ticks + 2 * N) // (p.S + p.F) + 1                      # the plan covers first_frame + ticks + 2 N samples
    solver = q.MPCSolver(q.reference_plan(rows=rows, params=p), params=p, device=R.local_rank)
    st0 = np.repeat(np.zeros(1, dtype=q.TICK_IN), batch); st0["com_pos"][:, 2] = p.h_des
    rng = np.random.default_rng(1)
    st0["com_pos"][:, :2] += rng.uniform(-0.004, 0.004, (batch, 2)); st0["com_vel"][:, :2] += rng.uniform(-0.02, 0.02, (batch, 2))
    d0 = q.to_device(st0, R.dev); d = d0.clone()
    solver.rollout_torch(d, 0, 20, want_traj=False); torch.cuda.synchronize()
    calls, t0 = 0, time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while True:
        for _ in range(8):
            d.copy_(d0); solver.rollout_torch(d, 0, ticks, want_traj=False); calls += 1
        torch.cuda.synchronize()
        if time.perf_counter() - t0 >= seconds:
            break
    b.record(); torch.cuda.synchronize()
    gpu_s = a.elapsed_time(b) * 1e-3
    fin = q.from_device(d, q.TICK_IN)
    # one tick of the final state through the per-tick entry point: its status word says whether the loop stayed inside the plan
    st = q.from_device(solver.solve_batch_torch(d), q.TICK_OUT)["status"]
    res = {"name": "closed loop on the device (ismpc_rollout_device), N=%d, %d instances x %d ticks per call" % (N, batch, ticks),
           "value": batch * ticks * calls / gpu_s, "unit": "ticks/s", "gpu_seconds": gpu_s, "calls": calls, "ticks_per_call": ticks, "batch": batch,
           "us_per_tick": 1e6 * gpu_s / (calls * ticks), "final_control_iter_max": int(fin["control_iter"].max()),
           "final_status_bad_index": int(((st & q.ST_BAD_INDEX) != 0).sum()), "final_status_error": int(((st & q.ST_ERROR_MASK) != 0).sum())}
    solver.close()
    return res


def a_kernel_name(C, F, per_inst, dtype):
    rl = max(2, (C + 63) // 64)
    real = "double" if dtype == "f64" else "float"
    if per_inst and os.environ.get("ISMPC_A_BUCKET") == "1" and F > 3:
        return f"ismpc_a_tick_wave<{real}, {rl}, F, true>, F = 3..{F}: one launch per footstep count (instances grouped by F_i)"
    return f"ismpc_a_tick_wave<{real}, {rl}, {F}, {'true' if per_inst else 'false'}>"


def leg_a(R, q, leg, name, batch, K, W, min_ms, dtype="f64"):
    """Formulation A: `batch` instances PER RANK (weak: every rank draws its own instances); one tick per step from the
    same pushed states (state restored by a device copy inside the step, as a caller replaying a Monte-Carlo draw does)."""
    torch = R.torch
    from quadruped_gait_generation_ismpc_amd import formulation_a as FA, workload
    from quadruped_gait_generation_ismpc_amd.distributed import gather_records
    world, rank = R.world, R.rank
    prec = {"f64": None, "f32": "f32"}[dtype]
    kw = {} if prec is None else {"precision": prec}
    if name == "mc_C200":
        Cn, Pn, Fn = 200, 400, 6
        inst, push = workload.make_inst_mc(batch, stream=rank)
        plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
        gen = FA.GaitGenerator(FA.default_params(0, C=Cn, P=Pn, F=Fn), plans[0], device=R.local_rank, **kw); gen.add_plan(plans[1])
        d_inst = q.to_device(inst, R.dev)
        d0 = q.to_device(gen.initial_state(0.88, batch=batch), R.dev)
        # data preparation, not the timed path: a nominal closed loop spreads the gait phases.  It runs on a second handle
        # with the OTHER arithmetic type, so that its 60 cheap launches carry another kernel name and a rocprofv3 --stats
        # summary of this process averages only the timed launches under the leg's kernel
        # (an fp32 handle follows every launch with a small fp64 launch (up to 64 workgroups) for the QPs it could not solve: switched off on
        # the preparation handle, or those would carry the fp64 leg's kernel name)
        saved = os.environ.get("ISMPC_A_F32_RESOLVE"); os.environ["ISMPC_A_F32_RESOLVE"] = "0"
        prep = FA.GaitGenerator(FA.default_params(0, C=Cn, P=Pn, F=Fn), plans[0], device=R.local_rank,
                                precision=("f32" if dtype == "f64" else "f64")); prep.add_plan(plans[1])
        if saved is None: del os.environ["ISMPC_A_F32_RESOLVE"]
        else: os.environ["ISMPC_A_F32_RESOLVE"] = saved
        prep.rollout_inst_torch(d0, d_inst, MC_PREROLL); torch.cuda.synchronize(); prep.close()
        tick = lambda st, pu: gen.tick_inst_torch(st, d_inst, pu)
        desc = (f"Formulation A Monte-Carlo (BASELINE configs[4] per-GPU shape): trot / walk by instance parity, C={Cn}, P={Pn}, per-instance CoM height "
                f"U(0.50,0.62), step U{{40..100}}, ds=round(0.6 step), F=ceil(C/step)+1 <= 6, Qf 1e7/1e9; {MC_PREROLL} nominal ticks then ONE pushed tick")
        per_inst = True
    else:
        w = workload.make_batch_a(name, batch, stream=rank)
        Cn, Pn, Fn = w["C"], w["P"], w["F"]
        g = FA.default_gait(w["kind"], w["phi"], w["disp_A"])
        _, ce = FA.plan(g)
        gen = FA.GaitGenerator(FA.default_params(w["kind"], C=Cn, P=Pn, F=Fn), ce, device=R.local_rank, **kw)
        d0 = q.to_device(w["state"], R.dev); push = w["push"]
        tick = lambda st, pu: gen.tick_torch(st, pu)
        desc = (f"Formulation A (MATLAB ISMPC tick with footstep adaptation), {'walk' if w['kind'] == 1 else 'trot'} plan phi=pi/4, C={Cn}, P={Pn}, F={Fn}, "
                f"step/ds {gen.params.step}/{gen.params.ds}, Qf={gen.params.Qf:g}; nominal state at a random tick + push (SURVEY 8d config 4)")
        per_inst = False
    dpush = torch.from_numpy(push.copy()).to(R.dev)
    d = d0.clone()
    ga = None
    if R.collective and R.abi and not R.rehearse:
        # the native group (include/ismpc_group.h): this rank's `batch` instances are shard `rank` of a global batch of world x batch
        from quadruped_gait_generation_ismpc_amd import group as G

        def make_group(uid):
            if name == "mc_C200":
                g_ = G.GroupA.from_rank(FA.default_params(0, C=Cn, P=Pn, F=Fn), plans[0], R.local_rank, uid, rank, world); g_.add_plan(plans[1])
                return g_
            return G.GroupA.from_rank(FA.default_params(w["kind"], C=Cn, P=Pn, F=Fn), ce, R.local_rank, uid, rank, world)
        ga = R.native_group(make_group)
    if ga is not None:
        ga.set_precision(dtype == "f32")
        assert ga.world == world and ga.shard(world * batch) == (rank * batch, batch)
        ga.reserve(world * batch)
    all_out = torch.empty((world * batch, 80), dtype=torch.uint8, device=("cpu" if R.rehearse else R.dev)) if (R.collective and ga is None) else None
    last = [None]
    evs = []
    ts = torch.cuda.current_stream(R.dev).cuda_stream

    # A tick updates its state records in place, and every step must solve the SAME pushed tick: step k of a region works on its own
    # copy of the pushed state, all of them put back before the region's clock starts (inputs resident in HBM when the timed region
    # begins; rounds 1-3 replayed the state inside the step -- a 1.5 MB device copy and its launch gap charged to every tick)
    nring = max(K, W, 1)
    ring = [d0.clone() for _ in range(nring)] if nring * d0.numel() * d0.element_size() <= (8 << 30) else None

    def put_back():
        if ring is not None:
            for r_ in ring:
                r_.copy_(d0)

    def state_of(k):
        if ring is not None:
            return ring[k % nring]
        d.copy_(d0)
        return d

    def group_step(k):
        dk = state_of(k)
        if ring is None:
            ga.order_after(0, ts)                                        # the replay on torch's stream, which the group's launch stream waits for
        ga.step_device(world * batch, [dk.data_ptr()], [d_inst.data_ptr()] if name == "mc_C200" else None, [dpush.data_ptr()], k & 1)
        if ring is None:
            ga.wait_on(0, k & 1, ts)                                     # the next copy into `d` comes after this step's kernel (and collective)

    def step(k):
        o = tick(state_of(k), dpush)
        last[0] = o
        if all_out is not None:
            gather_records(o.cpu() if R.rehearse else o, world, out=all_out, counts=[batch] * world, force=True, group=R.data_group)

    def step_timed(k):                                                   # the same tick between two events: kernel_ms (untimed pass)
        dk = state_of(k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); o = tick(dk, dpush); b.record(); evs.append((a, b))
        last[0] = o

    if ga is not None:
        walls = timed_regions(R, group_step, K, W, min_ms, drain=ga.sync, prep=put_back)
        gathered = q.from_device(ga.result_torch(world * batch, 0, (K - 1) & 1).clone(), FA.OUT_A)
    else:
        walls = timed_regions(R, step, K, W, min_ms, prep=put_back)
    wall = statistics.median(walls)
    torch.cuda.synchronize()
    timed_regions(R, step_timed, min(K, 5), 1, 0.0, prep=put_back)       # the tick through the plain handle between events: kernel_ms (and, for a group, the bytes to compare with)
    torch.cuda.synchronize()
    kernel_ms = statistics.median(a.elapsed_time(b) for a, b in evs[1:])
    o = q.from_device(last[0], FA.OUT_A)
    if ga is not None:
        assert gathered[rank * batch:(rank + 1) * batch].tobytes() == o.tobytes(), "group all-gather misplaced or altered this rank's shard"
    res = None
    if rank == 0:
        act = ((o["active"] & 0xffff) + (o["active"] >> 16)) / 2.0
        wmean = float(act.mean())
        flops = flops_a(Cn, Fn, wmean) * batch
        kname = a_kernel_name(Cn, Fn, per_inst, dtype)
        legkey = leg if dtype == "f64" else f"{leg}_{dtype}"
        value = world * batch / (wall / K)
        res = {
            "value": value, "unit": "ticks/s (1 tick = 2 per-axis QPs of C+F variables)", "ms_per_step": 1e3 * wall / K, "dtype": dtype,
            "qp_solves_per_s": 2.0 * value, "n_gpus": world, "scaling": "weak",
            "regions": len(walls),
            "config": {"workload": desc, "horizon": Cn, "batch_per_gpu": batch, "global_batch": world * batch,
                       "status_nonzero": int((o["status"] != 0).sum()),
                       "iterations_per_qp_mean": float((o["iters_x"] + o["iters_y"]).mean() / 2),
                       "iterations_per_qp_max": int(max(o["iters_x"].max(), o["iters_y"].max())),
                       "working_set_mean": wmean, "working_set_max": int(max((o["active"] & 0xffff).max(), (o["active"] >> 16).max())),
                       "active_box_fraction": float(((o["active"] & 0xffff) > 1).mean() / 2 + ((o["active"] >> 16) > 1).mean() / 2)},
            "roofline": roofline(legkey, kname, kernel_ms, batch, dtype, flops, 136.0 * batch,
                                 note="kernel_ms = HIP events around every tick call of the timed regions, median (the state copy precedes the first event; "
                                      "prologue launch, the wave kernel and, behind an fp32 launch, the fp64 re-solve launch inside; the wave kernel is 97-99 % "
                                      "of it: compare the rocprofv3 average under profiles/r04).  algorithmic_credit = SURVEY 8d flops_A = 2[nv w^2 + w^3/3 + "
                                      "4 nv w + 6 nv], nv = C+F, w = measured mean working-set size (a dense active-set solve; the structured solver executes "
                                      "less).  fp32 legs execute fp64 instructions too (right-hand sides, prefix sums, the LIP update): both types are counted."),
        }
        if dtype == "f32":
            res["config"]["deferred_to_fp64"] = int(gen.last_deferred())
        if ga is not None:
            res["multi_gpu"] = {"rccl_world": ga.world, "path": "abi", "kernel_ms": kernel_ms, "overlapped_step_ms": 1e3 * wall / K}
    if ga is not None:
        ga.close()
    gen.close()
    return res


# ======================================================================================================================
# The ONE line: numbers only, <= LINE_LIMIT bytes; everything else goes to bench_detail.json
# ======================================================================================================================
def _sig(x, n=5):
    """n significant digits (the line is size-limited; bench_detail.json keeps full precision)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{n}g}")
    except (TypeError, ValueError):
        return x


def compact_roofline(rf, full=True):
    if not rf:
        return None
    ex = rf.get("executed") or {}
    out = {"frac": _sig(rf.get("frac"), 4), "kernel": rf.get("kernel"), "kernel_ms": _sig(rf.get("kernel_ms")), "traffic": _sig(rf.get("traffic"))}
    if full:
        out = {"bound": rf.get("bound"), "achieved": _sig(rf.get("achieved")), "peak": _sig(rf.get("peak")), "unit": rf.get("unit"),
               "frac": _sig(rf.get("frac"), 4), "traffic": _sig(rf.get("traffic")), "kernel": rf.get("kernel"), "kernel_ms": _sig(rf.get("kernel_ms")),
               "kernel_ms_train": _sig(rf.get("kernel_ms_train")), "algorithmic_credit": {"tflops": _sig((rf.get("algorithmic_credit") or {}).get("tflops")),
                                                                                         "bytes_per_launch": _sig((rf.get("algorithmic_credit") or {}).get("bytes_per_launch"))},
               "valu_busy_frac": _sig(ex.get("valu_busy_frac"), 3), "fp_share": _sig(ex.get("fp_share_of_valu_instructions"), 3),
               "pmc_matches_lib": bool(rf.get("pmc_matches_lib"))}
    return out


def compact_cpu(cb, full=True):
    if not cb:
        return None
    out = {"value": _sig(cb.get("value"), 4)}
    if full:
        s = str(cb.get("sample", ""))
        out = {"value": _sig(cb.get("value"), 4), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
               "sample": s if len(s) <= 200 else s[:197] + "...",
               "all_cores": {"value": _sig((cb.get("all_cores") or {}).get("value"), 4), "cores": (cb.get("all_cores") or {}).get("cores")},
               "own": ({"value": _sig(cb["own"]["value"], 4), "kind": "port", "cores": 1} if cb.get("own") else None),
               "cpu_model": cb.get("cpu_model"), "nproc": cb.get("nproc")}
    return out


def compact_line(full, detail_path):
    """The driver's line from the full result: the contract's keys, numeric roofline and cpu_baseline, other_configs cut to numbers."""
    cfg = full.get("config", {})
    wl = str(cfg.get("workload", ""))
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"] = _sig(full.get("value"), 7); line["ms_per_step"] = _sig(full.get("ms_per_step"), 6)
    line["qp_solves_per_s"] = _sig(full.get("qp_solves_per_s"), 7)
    line["config"] = {"workload": wl if len(wl) <= 220 else wl[:217] + "...", "horizon": cfg.get("horizon"), "global_batch": cfg.get("global_batch"),
                      "batch_per_gpu": cfg.get("batch_per_gpu"), "active_box_fraction": _sig(cfg.get("active_box_fraction"), 3)}
    line["regions"] = full.get("regions"); line["region_ms_median"] = _sig((full.get("region_ms") or {}).get("median"))
    line["roofline"] = compact_roofline(full.get("roofline"))
    if full.get("cpu_baseline"):
        line["cpu_baseline"] = compact_cpu(full["cpu_baseline"])
    for k in ("value_incl_pcie", "value_incl_pcie_pageable", "latency_batch1_us"):
        if k in full:
            line[k] = _sig(full[k], 4)
    if full.get("multi_gpu"):
        line["multi_gpu"] = {k: _sig(v) for k, v in full["multi_gpu"].items() if not isinstance(v, (dict, list)) and not (isinstance(v, str) and len(v) > 48)}
    if full.get("sustained"):
        su = full["sustained"]
        line["sustained"] = {"leg": "ismpc_rollout_device closed loop", "value": _sig(su["value"], 6), "unit": su["unit"], "gpu_seconds": _sig(su["gpu_seconds"], 4),
                             "batch": su["batch"], "ticks_per_call": su["ticks_per_call"], "calls": su["calls"]}
    oc = []
    for o in full.get("other_configs", []):
        c = {"name": (o["name"] if len(o["name"]) <= 96 else o["name"][:93] + "..."), "value": _sig(o.get("value"), 6), "ms_per_step": _sig(o.get("ms_per_step")),
             "dtype": o.get("dtype"), "roofline": compact_roofline(o.get("roofline"), full=False)}
        if o.get("cpu_baseline"):
            c["cpu_baseline"] = compact_cpu(o["cpu_baseline"], full=False)
        oc.append(c)
    if oc:
        line["other_configs"] = oc
    line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:               # never over the limit: shed the optional parts, most dispensable first
        for k in ("other_configs", "sustained", "multi_gpu"):
            if k == "other_configs" and "other_configs" in line:
                line["other_configs"] = [{"name": c["name"][:60], "value": c["value"], "ms_per_step": c["ms_per_step"], "dtype": c["dtype"],
                                          "roofline": {"frac": (c.get("roofline") or {}).get("frac")}} for c in line["other_configs"]]
            elif k in line:
                del line[k]
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    while len(text) > LINE_LIMIT and line.get("other_configs"):
        line["other_configs"].pop(); line["other_configs_dropped"] = line.get("other_configs_dropped", 0) + 1      # (they are all in the detail file)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def write_detail(full):
    """bench_detail.json next to bench.py (and a copy under gpurun_out/ when that directory exists: it is what travels back from a GPU box)."""
    path = os.path.join(ROOT, DETAIL_NAME)
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        go = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(go):
            with open(os.path.join(go, DETAIL_NAME), "w") as f:
                json.dump(full, f, indent=1)
    except OSError as e:
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    return DETAIL_NAME


def emit(full):
    print(compact_line(full, write_detail(full)), flush=True)


# ======================================================================================================================
def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-worker":
        cpu_worker(json.loads(sys.argv[2]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH)
    ap.add_argument("--horizon", type=int, default=HORIZON)
    ap.add_argument("--min-region-ms", type=float, default=50.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds of oracle work per cpu_baseline measurement")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and batch-1 latency measurements (profiling runs: "
                                                              "every launch of the process then has the leg's own shape)")
    ap.add_argument("--only", default=None, help="run ONE leg and print it as the line: headline | config1_b1024 | config3_walk_C150 | "
                                                  "config4_mc_C200 | shard_b8192 | shard_b16384 | shard_b32768 | sweep_k64_b65536 | a_walk_C100 | a_trot_C160 (profiling runs: one kernel shape per process)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="with --only <Formulation A leg>: arithmetic type of the QP solve")
    ap.add_argument("--sustained-seconds", type=float, default=8.0, help="GPU seconds of the sustained closed-loop leg (0: skip it)")
    ap.add_argument("--full-line", action="store_true", help="print the full result object (what bench_detail.json holds) instead of the compact line "
                                                              "(profiling scripts that keep one leg's complete object)")
    ap.add_argument("--spawn", action="store_true", help="start the rank processes from this process even for --gpus 1 (the launcher path of `--gpus N` as typed)")
    ap.add_argument("--collective", default="abi", choices=["abi", "torch"], help="who issues the all-gather at N > 1: the native library's own RCCL communicator "
                                                                                   "(include/ismpc_group.h; default) or torch.distributed's nccl backend (A/B)")
    ap.add_argument("--force-collective", action="store_true", help="build the RCCL group and run the path's all-gather even with ONE rank "
                                                                     "(exercises communicator set-up and the side-stream pipeline on a one-GPU box; moves no bytes between GPUs)")
    ap.add_argument("--launcher-selftest", type=int, default=None, help=argparse.SUPPRESS)   # rank that fails (-1: none); no GPU is touched
    args = ap.parse_args()
    if os.environ.get("ISMPC_BENCH_CHILD") != "1" and (args.spawn or (args.gpus > 1 and "WORLD_SIZE" not in os.environ)):
        # the launcher never touches the GPU (a process that did must not be replaced or forked around): it only starts the ranks
        sys.exit(spawn_ranks(args.gpus, [a for a in sys.argv[1:] if a != "--spawn"]))

    if args.launcher_selftest is not None:
        # what a rank process sees, without touching a GPU (tests/test_distributed.py::test_bench_launcher_*)
        r = int(os.environ.get("RANK", "0"))
        if r == 0:
            print(json.dumps({"rank": r, "world": int(os.environ.get("WORLD_SIZE", "1")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                              "master": os.environ.get("MASTER_ADDR"), "port": int(os.environ.get("MASTER_PORT", "0")), "gpus": args.gpus}), flush=True)
        if r == args.launcher_selftest:
            sys.exit(7)
        time.sleep(0.5 if args.launcher_selftest < 0 else 30)
        return

    R = Ranks(args.gpus, force_collective=args.force_collective, collective=args.collective)
    world, rank = R.world, R.rank
    import quadruped_gait_generation_ismpc_amd as q
    K, W, M = args.steps, args.warmup, args.min_region_ms
    a_steps = max(3, min(K, 10))                                        # Formulation A steps take milliseconds

    LEGS = {
        "headline":         lambda: leg_b(R, q, "headline_b%d" % (args.global_batch // world), args.horizon, args.global_batch, K, W, M, extras=not args.no_extras),
        "config1_b1024":    lambda: leg_b(R, q, "config1_b1024", args.horizon, 1024, K, W, M, extras=False),
        "shard_b8192":      lambda: leg_b(R, q, "shard_b8192", args.horizon, 8192, K, W, M, extras=False),
        "shard_b16384":     lambda: leg_b(R, q, "shard_b16384", args.horizon, 16384, K, W, M, extras=False),    # the per-GPU shards of the headline at N = 4, 2
        "shard_b32768":     lambda: leg_b(R, q, "shard_b32768", args.horizon, 32768, K, W, M, extras=False),
        "sweep_k64_b65536": lambda: leg_b(R, q, "sweep_k64_b65536", args.horizon, args.global_batch, K, W, M, extras=False, sweep_sets=64),
        "config3_walk_C150": lambda dt="f64": leg_a(R, q, "config3_walk_C150", "walk_C150", A_BATCH, a_steps, 2, M, dt),
        "config4_mc_C200":  lambda dt="f64": leg_a(R, q, "config4_mc_C200", "mc_C200", A_BATCH, a_steps, 2, M, dt),
        "a_walk_C100":      lambda dt="f64": leg_a(R, q, "a_walk_C100", "walk_C100", A_BATCH, a_steps, 2, M, dt),
        "a_trot_C160":      lambda dt="f64": leg_a(R, q, "a_trot_C160", "trot_C160", A_BATCH, a_steps, 2, M, dt),
    }
    CPU = {
        "B": lambda: cpu_baseline("B", "ticks/s", args.cpu_budget, "first instances of the same batch", horizon=args.horizon),
        "config3_walk_C150": lambda: cpu_baseline("A:walk_C150", "ticks/s", args.cpu_budget, "first instances of the same pushed batch"),
        "config4_mc_C200": lambda: cpu_baseline("A:mc_C200", "ticks/s", args.cpu_budget,
                                                f"first instances of the same per-instance draw, {MC_PREROLL} nominal closed-loop ticks + the pushed tick each",
                                                unit_per_n=MC_PREROLL + 1),
        "a_walk_C100": lambda: cpu_baseline("A:walk_C100", "ticks/s", args.cpu_budget, "first instances of the same pushed batch"),
        "a_trot_C160": lambda: cpu_baseline("A:trot_C160", "ticks/s", args.cpu_budget, "first instances of the same pushed batch"),
    }
    a_legs = ("config3_walk_C150", "config4_mc_C200", "a_walk_C100", "a_trot_C160")
    from quadruped_gait_generation_ismpc_amd import formulation_a as FA
    have_f32 = bool(getattr(FA, "HAVE_F32", False))

    def with_cpu(res, key):
        if res is not None and world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = CPU[key]()
        return res

    common = {"n_gpus": world, "steps": K, "warmup": W, "higher_is_better": True, "vs_baseline": None,
              "data": "synthetic" if not R.rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a measurement)"}
    if args.only:
        if args.only not in LEGS:
            raise SystemExit(f"--only: unknown leg {args.only}")
        res = LEGS[args.only](args.dtype) if args.only in a_legs else LEGS[args.only]()
        if rank == 0:
            line = {"metric": "ISMPC QP solves/s (batch, N=100 horizon)" if args.only not in a_legs else "ISMPC ticks/s (Formulation A)"}
            line.update(res); line.update(common); line.setdefault("scaling", "strong")
            if args.only in a_legs:
                line["steps"] = a_steps; line["warmup"] = 2
            with_cpu(line, args.only if args.only in a_legs else "B")
            if args.full_line:
                print(json.dumps(line), flush=True)
            else:
                emit(line)
        R.close()
        return

    head = LEGS["headline"]()
    others = []
    sustained = None
    if world == 1 and "multi_gpu" not in head:
        head["multi_gpu"] = leg_group_one(R, q, args.horizon, args.global_batch, K, W, M)
    if world == 1 and args.sustained_seconds > 0:
        sustained = leg_sustained(R, q, args.horizon, args.global_batch, args.sustained_seconds)
    if not args.no_other_configs:
        if world == 1:
            r1 = LEGS["config1_b1024"]()
            rs = LEGS["sweep_k64_b65536"]()
            r3 = LEGS["config3_walk_C150"]()
            r3f = LEGS["config3_walk_C150"]("f32") if have_f32 else None
        r4 = LEGS["config4_mc_C200"]()
        r4f = LEGS["config4_mc_C200"]("f32") if have_f32 else None
    if rank == 0:
        line = {"metric": "ISMPC QP solves/s (batch, N=100 horizon)"}
        line.update(head); line.update(common)
        line["scaling"] = "strong"
        cpu_b = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_b = CPU["B"]()
            line["cpu_baseline"] = cpu_b
        if not args.no_other_configs:
            if world == 1:
                r1.update({"name": "BASELINE configs[1]: 1 024 instances, N=100, fp64, one GPU", "n_gpus": 1, "steps": K, "warmup": W})
                if cpu_b is not None:
                    r1["cpu_baseline"] = dict(cpu_b, note="same per-tick workload as the headline (Formulation B, N=100): measured once")
                others.append(r1)
                rs.update({"name": "north_star parameter sweep: 64 parameter sets in one batch of 65 536, N=100, fp64, one GPU (per-instance set index)", "n_gpus": 1, "steps": K, "warmup": W})
                if cpu_b is not None:
                    rs["cpu_baseline"] = dict(cpu_b, note="per-tick CPU cost as the headline (the oracle rebuilds its tables per set; not timed here)")
                others.append(rs)
                r3.update({"name": "BASELINE configs[3]: walking gait, N=150, batch 16 384, one GPU (fp64 solve)", "steps": a_steps, "warmup": 2})
                with_cpu(r3, "config3_walk_C150"); others.append(r3)
                if r3f is not None:
                    r3f.update({"name": "BASELINE configs[3]: walking gait, N=150, batch 16 384, one GPU (fp32 solve, fp64 state update)", "steps": a_steps, "warmup": 2})
                    if "cpu_baseline" in r3:
                        r3f["cpu_baseline"] = r3["cpu_baseline"]
                    others.append(r3f)
            r4.update({"name": f"BASELINE configs[4] shape: Monte-Carlo trot/walk, N=200, 16 384 instances per GPU x {world} GPU(s) (fp64 solve)", "steps": a_steps, "warmup": 2})
            with_cpu(r4, "config4_mc_C200"); others.append(r4)
            if r4f is not None:
                r4f.update({"name": f"BASELINE configs[4] shape: Monte-Carlo trot/walk, N=200, 16 384 instances per GPU x {world} GPU(s) (fp32 solve, fp64 state update)", "steps": a_steps, "warmup": 2})
                if "cpu_baseline" in r4:
                    r4f["cpu_baseline"] = r4["cpu_baseline"]
                others.append(r4f)
            line["other_configs"] = others
        if sustained is not None:
            line["sustained"] = sustained
        if args.full_line:
            print(json.dumps(line), flush=True)
        else:
            emit(line)
    R.close()


if __name__ == "__main__":
    main()
