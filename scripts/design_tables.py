#!/usr/bin/env python3
"""Regenerates the measurement tables of DESIGN.md section 5 from the files committed under profiles/r04/ (bench lines printed by
scripts/profile_r04.sh, rocprofv3 kernel stats, PMC summaries, kernel_resources.md) -- the numbers in the document cannot go stale.
usage: python scripts/design_tables.py            (rewrites the block between the GENERATED markers of DESIGN.md)"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r04")


def line(path):
    try:
        return json.load(open(path))                      # a whole-file object (bench_detail.json)
    except Exception:
        pass
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])       # the last JSON line of a captured stdout
    except Exception:
        return None


def stats_avg(key, kern):
    path = os.path.join(P, f"{key}_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    best = None
    for row in csv.DictReader(open(path)):
        if kern in row["Name"]:
            v = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
            if best is None or v[1] > best[1]:
                best = v
    return best


def fmt(x, nd=2):
    return "—" if x is None else (f"{x:.{nd}e}" if abs(x) >= 1e5 or (abs(x) < 1e-2 and x != 0) else f"{x:.{nd}f}")


LEGS = [("headline_b65536", "B, 65 536 instances (configs[2], the bench line)", "ismpc_tick_quad_one<"),
        ("shard_b32768", "B, 32 768 instances (configs[2] per-GPU shard at N = 2)", "ismpc_tick_quad_one<"),
        ("shard_b16384", "B, 16 384 instances (configs[2] per-GPU shard at N = 4)", "ismpc_tick_quad_inline<"),
        ("shard_b8192", "B, 8 192 instances (configs[2] per-GPU shard at N = 8)", "ismpc_tick_quad_inline<"),
        ("config1_b1024", "B, 1 024 instances (configs[1])", "ismpc_tick_quad_inline<"),
        ("sweep_k64_b65536", "B, 65 536 instances, 64 parameter sets (sweep)", "ismpc_tick_quad<"),
        ("config3_walk_C150", "A, walk C=150, 16 384, fp64 solve (configs[3])", "ismpc_a_tick_wave<double"),
        ("config3_walk_C150_f32", "A, walk C=150, 16 384, fp32 solve (configs[3])", "ismpc_a_tick_wave<float"),
        ("config4_mc_C200", "A, Monte-Carlo C=200, 16 384, fp64 solve (configs[4] shard)", "ismpc_a_tick_wave<double"),
        ("config4_mc_C200_f32", "A, Monte-Carlo C=200, 16 384, fp32 solve (configs[4] shard)", "ismpc_a_tick_wave<float")]

out = []
out.append("| leg | `value` ticks/s | step ms | dominant kernel | rocprofv3 avg µs | live isolated / train µs | executed TFLOP/s | `frac` | VALU busy | FP share of VALU | HBM traffic / algorithmic | credit TFLOP/s (SURVEY 8d) |")
out.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
for key, name, kern in LEGS:
    d = line(os.path.join(P, f"{key}_bench_line_unprofiled.json"))
    if d is None:
        continue
    r = d["roofline"]; ex = r.get("executed") or {}
    st = stats_avg(key, kern)
    iso = r["kernel_ms"] * 1e3; tr = r.get("kernel_ms_train")
    traffic = r.get("traffic"); alg = r["algorithmic_credit"]["bytes_per_launch"]
    out.append(f"| {name} | {d['value']:.3g} | {d['ms_per_step']:.4f} | `{r['kernel']}` | {fmt(st[0] if st else None, 1)} | {iso:.1f} / {fmt(tr * 1e3 if tr else None, 1)} | "
               f"{fmt(r.get('achieved'))} | {fmt(r.get('frac'), 3)} | {fmt(ex.get('valu_busy_frac'), 2)} | {fmt(ex.get('fp_share_of_valu_instructions'), 2)} | "
               f"{fmt(traffic / 1e6 if traffic else None, 1)} MB / {alg / 1e6:.1f} MB | {r['algorithmic_credit']['tflops']:.1f} |")
h = line(os.path.join(P, "bench_full_line.json")) or line(os.path.join(P, "headline_b65536_bench_line_unprofiled.json"))
extra = []
if h and "steps" in h and h.get("regions"):
    r = h["roofline"]
    extra.append(f"The default `python bench.py` of the same session ({h['steps']} steps per region, {h['regions']} regions: the table's legs are the short profiled runs, "
                 f"40 steps, 5 ms regions, and the chip clocks higher in the long one): **{h['value']:.3g} ticks/s**, {h['ms_per_step'] * 1e3:.1f} µs per step, "
                 f"`{r['kernel']}` {r['kernel_ms'] * 1e3:.1f} µs isolated / {r['kernel_ms_train'] * 1e3:.1f} µs in a train, `frac` {r.get('frac') or 0:.3f}.")
if h:
    extra.append(f"Host entry point `ismpc_solve_batch`, 65 536 records in and out: page-locked caller buffers (zero copy) **{h.get('value_incl_pcie', 0):.3g} ticks/s** "
                 f"({h.get('ms_per_step_incl_pcie', 0):.3f} ms), pageable buffers (staged) {h.get('value_incl_pcie_pageable', 0):.3g} ticks/s ({h.get('ms_per_step_incl_pcie_pageable', 0):.3f} ms); "
                 f"batch of one {h.get('latency_batch1_us', 0):.1f} µs median / {h.get('latency_batch1_us_p99', 0):.1f} µs p99.")
    cb = h.get("cpu_baseline")
    if cb:
        extra.append(f"CPU beside it ({cb['cpu_model']}): reference qpOASES {cb['value']:.0f} ticks/s on one core ({cb['ms_per_unit']:.2f} ms/tick), "
                     f"{cb['all_cores']['value']:.3g} on {cb['all_cores']['cores']} cores.")
if h:
    cb = h.get("cpu_baseline") or {}
    if cb.get("own"):
        extra.append(f"The build's own CPU restatement with its dense Goldfarb–Idnani QP (SURVEY §8d (i)): {cb['own']['value']:.0f} ticks/s on one core.")
    su = h.get("sustained")
    if su:
        extra.append(f"Sustained closed loop (`ismpc_rollout_device`, {su['ticks_per_call']} ticks × {su['batch']} instances per call, {su['calls']} calls back to back, "
                     f"{su['gpu_seconds']:.1f} s of GPU time): **{su['value']:.3g} ticks/s**, {su['us_per_tick']:.1f} µs per tick of the whole batch.")
    mg = h.get("multi_gpu")
    if mg and mg.get("path") == "abi":
        extra.append(f"C-ABI group of one device (`ismpc_group_step_device`, in-place `ncclAllGather` on the side stream; RCCL {mg.get('rccl_version')}, `ncclCommCount` = "
                     f"{mg.get('rccl_world')}): {mg.get('group_step_ms', 0) * 1e3:.1f} µs per step of 65 536 instances ({mg.get('group_value', 0):.3g} ticks/s); no scaling curve measured.")
    try:
        ks = list(csv.DictReader(open(os.path.join(P, "driver_cmd_kernel_stats.csv"))))
        dl = line(os.path.join(P, "driver_cmd_bench_line.json"))
        top = [r for r in ks if h["roofline"]["kernel"] in r["Name"]]
        if top and dl:
            extra.append(f"The exact driver command (`python3 bench.py --gpus 1 --steps 20 --warmup 5`) under rocprofv3 (`profiles/r04/driver_cmd_kernel_stats.csv`): "
                         f"`{h['roofline']['kernel']}` {float(top[0]['AverageNs']) / 1e3:.1f} µs average over {top[0]['Calls']} launches of the whole process (isolated launches, "
                         f"the host-path calls and the group leg included; a launch alone is longer than its share of a train, see above), against `ms_per_step` "
                         f"{dl['ms_per_step'] * 1e3:.1f} µs, `roofline.kernel_ms_train` {dl['roofline']['kernel_ms_train'] * 1e3:.1f} µs and `kernel_ms` {dl['roofline']['kernel_ms'] * 1e3:.1f} µs of the "
                         f"un-profiled run of the same command (`driver_cmd_bench_line.json`: **{dl['value']:.3g} ticks/s**, `frac` {dl['roofline']['frac']:.3f}, a line of "
                         f"{len(json.dumps(dl, separators=(',', ':')))} bytes).")
    except (OSError, KeyError, ValueError):
        pass
sw = line(os.path.join(P, "sweep_k64_b65536_bench_line_unprofiled.json"))
if sw and "sweep" in sw:
    s = sw["sweep"]
    g = json.load(open(os.path.join(P, "pmc_sweep_gemm.json"))) if os.path.exists(os.path.join(P, "pmc_sweep_gemm.json")) else None
    st = stats_avg("sweep_k64_b65536", "sweep_gemm<1>")
    t = f"Sweep table build, {s['n_sets']} sets: {s['build_ms']:.2f} ms in all, {s['newton_iterations']} Newton–Schulz iterations = {s['mfma_gemm_launches']} batched MFMA products"
    if st:
        fl = 2.0 * 128 ** 3 * s["n_sets"]
        t += f" of {st[0]:.1f} µs each ({fl / (st[0] * 1e-6) / 1e12:.1f} TFLOP/s, {fl / (st[0] * 1e-6) / 1e12 / 78.6:.2f} of the FP64 matrix peak"
        if g:
            t += f"; `SQ_INSTS_VALU_MFMA_MOPS_F64` {g['counters_mean_per_launch'].get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0):.0f} per launch, MFMA busy {g['derived'].get('mfma_busy_frac', 0):.2f} of the kernel's cycles"
        t += ")"
    extra.append(t + ".")
k5 = line(os.path.join(P, "sweep_build_k512.json"))
if k5:
    st = stats_avg("sweep_build_k512", "sweep_gemm<1>")
    g = json.load(open(os.path.join(P, "pmc_sweep_gemm_k512.json"))) if os.path.exists(os.path.join(P, "pmc_sweep_gemm_k512.json")) else None
    t = f"The same build for {k5['n_sets']} sets (`scripts/sweep_build_probe.py`, a grid that fills the chip): {k5['build_ms']:.2f} ms in all"
    if st:
        fl = 2.0 * 128 ** 3 * k5["n_sets"]
        t += f"; `sweep_gemm` {st[0]:.1f} µs per launch = **{fl / (st[0] * 1e-6) / 1e12:.1f} TFLOP/s, {fl / (st[0] * 1e-6) / 1e12 / 78.6:.2f} of the FP64 matrix peak**"
    if g:
        t += f", MFMA busy {g['derived'].get('mfma_busy_frac', 0):.2f} of the kernel's cycles (`SQ_VALU_MFMA_BUSY_CYCLES`)"
    extra.append(t + f"; worst table error of three sampled sets against the long-double host build {k5['worst_rel_table_error_of_3_sets']:.1e}.")
vp = os.path.join(P, "valu_peak.json")
if os.path.exists(vp):
    try:
        v = json.loads("".join(l for l in open(vp) if not l.startswith("/opt") and "amdgpu.ids" not in l))
        extra.append(f"Measured VALU rates of the box (`scripts/micro/valu_peak.hip`): v_fma_f64 {v['v_fma_f64']['tflops']:.1f}, unpacked v_fma_f32 {v['v_fma_f32']['tflops']:.1f}, "
                     f"v_pk_fma_f32 {v['v_pk_fma_f32']['tflops']:.1f} TFLOP/s (spec: 78.6 / 157.3 / 157.3).")
    except Exception:
        pass
block = "\n".join(out) + "\n\n" + "\n\n".join(extra) + "\n"
path = os.path.join(ROOT, "DESIGN.md")
txt = open(path).read()
b, e = "<!-- BEGIN GENERATED:r04 (scripts/design_tables.py) -->", "<!-- END GENERATED:r04 -->"
if b in txt and e in txt:
    txt = txt[:txt.index(b) + len(b)] + "\n" + block + txt[txt.index(e):]
    open(path, "w").write(txt)
    print("DESIGN.md updated")
else:
    print(block)
