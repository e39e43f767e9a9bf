"""CPU-only: the host logic of bench.py that turns committed counter summaries into the contract's `roofline` object (no GPU,
no torch): executed flops over the kernel's own time against the VALU peak of the executed mix, never above 1 by construction of
its inputs; the SURVEY 8d figure kept apart as a credit; PMC summaries looked up by leg and batch, never scaled."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    monkeypatch.setenv("ISMPC_PROFILES_DIR", str(tmp_path))
    monkeypatch.setattr(b, "lib_sha256", lambda path=None: SHA)          # the library "loaded" in these tests ...
    monkeypatch.setattr(b, "lib_src_sha256", lambda path=None: SRC)      # ... and the sources it was built from
    return b, tmp_path


SHA, SRC = "5" * 64, "7" * 64


def _pmc(path, leg, batch, kernel, fl64, fl32, valu, active=None, busy=None, fetch=1000.0, write=500.0, sha=SHA, src=SRC):
    c = {"SQ_INSTS_VALU": valu}
    if active is not None:
        c["SQ_ACTIVE_INST_VALU"] = active; c["SQ_BUSY_CYCLES"] = busy
    d = {"flops_f64_per_launch": fl64, "flops_f32_per_launch": fl32, "fp_insts_f64_per_launch": fl64 / 100.0, "fp_insts_f32_per_launch": fl32 / 100.0,
         "fp_share_of_valu_insts": 0.4, "hbm_bytes_per_launch": 2.0 * fetch * 1024 + write * 1024}
    json.dump({"kernel": kernel, "batch": batch, "leg": leg, "launches_per_step": 1, "counters_mean_per_launch": c, "derived": d, "lib_sha256": sha, "src_sha256": src, "git_head": "abc1234"},
              open(path / f"pmc_{leg}.json", "w"))


def test_roofline_is_executed_work_over_the_blended_peak(bench):
    b, tmp = bench
    _pmc(tmp, "headline_b65536", 65536, "ismpc_tick_quad<", fl64=8.0e8, fl32=0.0, valu=2.2e7, active=2.3e7, busy=3.4e6)
    rf = b.roofline("headline_b65536", "ismpc_tick_quad<7, 16, false>", 0.05, 65536, "f64", alg_flops=4.06e9, alg_bytes=1.0e7)
    assert rf["bound"] == "valu" and rf["unit"] == "TFLOP/s"
    assert rf["achieved"] == pytest.approx(8.0e8 / 0.05e-3 / 1e12) and rf["peak"] == pytest.approx(78.6)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"]) and 0 < rf["frac"] <= 1
    assert rf["algorithmic_credit"]["tflops"] == pytest.approx(4.06e9 / 0.05e-3 / 1e12) and "frac" not in rf["algorithmic_credit"]   # a credit, not a fraction
    assert 0 < rf["executed"]["valu_busy_frac"] <= 1 and 0 < rf["executed"]["valu_issue_frac"] <= 1
    assert rf["traffic"] == pytest.approx(2.0 * 1000 * 1024 + 500 * 1024)
    # a kernel that executes both types is priced against the blend
    _pmc(tmp, "config3_walk_C150_f32", 16384, "ismpc_a_tick_wave<float", fl64=1.0e9, fl32=3.0e9, valu=1.6e8)
    rf = b.roofline("config3_walk_C150_f32", "ismpc_a_tick_wave<float, 3, 4, false>", 0.35, 16384, "f32", 1e10, 2e6)
    assert rf["peak"] == pytest.approx(4.0e9 / (1.0e9 / 78.6 + 3.0e9 / 157.3)) and 78.6 < rf["peak"] < 157.3
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])


def test_roofline_never_scales_another_batch_and_finds_the_shard_pass(bench):
    b, tmp = bench
    _pmc(tmp, "headline_b65536", 65536, "ismpc_tick_quad<", 8e8, 0.0, 2.2e7)
    rf = b.roofline("headline_b65536", "ismpc_tick_quad<7, 16, false>", 0.03, 32768, "f64", 1e9, 1e6)        # same leg name, other batch
    assert rf["achieved"] is None and rf["frac"] is None and rf["traffic"] is None and "no PMC summary" in rf["note"]
    rf = b.roofline("headline_b65536", "ismpc_tick_quad_inline<7, 16, 2>", 0.05, 65536, "f64", 1e9, 1e6)      # other kernel
    assert rf["achieved"] is None
    _pmc(tmp, "shard_b8192", 8192, "ismpc_tick_quad_inline<", 1e8, 0.0, 3e6)
    rf = b.roofline("headline_b8192", "ismpc_tick_quad_inline<7, 16, 2>", 0.012, 8192, "f64", 5e8, 1e6)       # the headline sharded over 8 ranks
    assert rf["achieved"] == pytest.approx(1e8 / 0.012e-3 / 1e12) and "shard_b8192" in rf["executed"]["source"]
    assert b.roofline("nothing_here", "k", 0.1, 1, "f64", 1.0, 1.0)["achieved"] is None


def test_kernel_names_and_credits_follow_the_launch_rules(bench, monkeypatch):
    b, _ = bench
    for k in ("ISMPC_PATH", "ISMPC_LPI", "ISMPC_ONE_LAUNCH", "ISMPC_Z_FALLBACK"):
        monkeypatch.delenv(k, raising=False)
    assert b.kernel_name_b(100, 65536, 256) == "ismpc_tick_quad_one<13, 8, 2, false>"     # > 8 wavefronts per CU: one launch at the tick's own residency; 8 lanes per instance beyond 8 192
    assert b.kernel_name_b(100, 16384, 256) == "ismpc_tick_quad_inline<13, 8, 2>"         # eight instances per wavefront: 16 384 are resident at once
    assert b.kernel_name_b(100, 65536, 256, sweep=True) == "ismpc_tick_quad_one<13, 8, 2, true>"          # sweep handles: 8 lanes beyond 8 192 too (round 4)
    assert b.kernel_name_b(100, 65536, 256, sweep=True, deferring=True) == "ismpc_tick_quad<13, 8, true>"   # ... two launches while instances are deferred
    assert b.kernel_name_b(100, 8192, 256, sweep=True, deferring=True) == "ismpc_tick_quad<7, 16, true>"
    assert b.kernel_name_b(100, 65536, 256, deferring=True) == "ismpc_tick_quad<13, 8, false>"
    assert b.kernel_name_b(100, 8192, 256, deferring=True) == "ismpc_tick_quad_inline<7, 16, 2>"
    monkeypatch.setenv("ISMPC_ONE_LAUNCH", "0")
    assert b.kernel_name_b(100, 65536, 256) == "ismpc_tick_quad<13, 8, false>"       # A/B: the two-launch form
    assert b.kernel_name_b(100, 8192, 256) == "ismpc_tick_quad<7, 16, false>"
    monkeypatch.setenv("ISMPC_ONE_LAUNCH", "1")
    assert b.kernel_name_b(100, 65536, 256) == "ismpc_tick_quad<13, 8, false>"       # A/B: one launch only for resident batches (rounds 2-3)
    monkeypatch.delenv("ISMPC_ONE_LAUNCH")
    assert b.kernel_name_b(100, 8192, 256) == "ismpc_tick_quad_inline<7, 16, 2>"     # every wavefront resident: one launch
    assert b.kernel_name_b(100, 1024, 256) == "ismpc_tick_quad_inline<4, 32, 2>"     # <= 2 048 instances: 32 lanes per instance
    assert b.kernel_name_b(200, 1024, 256) == "ismpc_tick_affine<4>"                 # N > 128: one instance per wavefront
    assert b.flops_b(100) == 6 * 100 * 100 + 20 * 100                                # SURVEY 8d, shared factor
    assert b.flops_a(150, 4, 0.0) == 2 * 6 * 154
    assert b.a_kernel_name(150, 4, False, "f32") == "ismpc_a_tick_wave<float, 3, 4, false>"
    assert b.a_kernel_name(200, 6, True, "f64") == "ismpc_a_tick_wave<double, 4, 6, true>"


def test_a_counter_summary_of_another_build_is_not_used(bench):
    """pmc_*.json carries the sha256 of the library it was collected from (scripts/pmc_summary.py); bench.py derives achieved / frac /
    traffic from it only when that is the library loaded now."""
    b, tmp = bench
    _pmc(tmp, "headline_b65536", 65536, "ismpc_tick_quad<", 8e8, 0.0, 2.2e7, sha="0" * 64, src="1" * 64)
    rf = b.roofline("headline_b65536", "ismpc_tick_quad<13, 8, false>", 0.05, 65536, "f64", 4e9, 1e7)
    assert rf["achieved"] is None and rf["frac"] is None and rf["traffic"] is None and rf["executed"] is None and "STALE" in rf["note"]
    # the same SOURCES built in another directory (the binary embeds its source paths: another lib_sha256) are the same code: accepted
    _pmc(tmp, "headline_b65536", 65536, "ismpc_tick_quad<", 8e8, 0.0, 2.2e7, sha="0" * 64)
    rf2 = b.roofline("headline_b65536", "ismpc_tick_quad<13, 8, false>", 0.05, 65536, "f64", 4e9, 1e7)
    assert rf2["frac"] is not None and rf2["pmc_matches_lib"] is True and b.compact_roofline(rf2)["pmc_matches_lib"] is True
    assert rf["algorithmic_credit"]["tflops"] > 0 and rf["pmc_lib_sha256"] == "0" * 64 and rf["lib_sha256"] == SHA
    assert b.compact_roofline(rf)["pmc_matches_lib"] is False and b.compact_roofline(rf)["frac"] is None
    _pmc(tmp, "headline_b65536", 65536, "ismpc_tick_quad<", 8e8, 0.0, 2.2e7)              # same counters, collected from the loaded library
    rf = b.roofline("headline_b65536", "ismpc_tick_quad<13, 8, false>", 0.05, 65536, "f64", 4e9, 1e7)
    assert rf["frac"] == pytest.approx(8e8 / 0.05e-3 / 1e12 / 78.6) and b.compact_roofline(rf)["pmc_matches_lib"] is True
    # the real hash function: a file's sha256, cached per path, None for a missing library
    import hashlib, importlib.util
    spec = importlib.util.spec_from_file_location("bench_real", os.path.join(ROOT, "bench.py"))
    real = importlib.util.module_from_spec(spec); spec.loader.exec_module(real)
    f = tmp / "lib.so"; f.write_bytes(b"\x7fELF" + bytes(1000))
    assert real.lib_sha256(str(f)) == hashlib.sha256(f.read_bytes()).hexdigest() and real.lib_sha256(str(tmp / "missing.so")) is None


def test_the_line_stays_under_the_limit_and_keeps_the_contract(bench, tmp_path):
    """Round 3's line was 27 KB and the driver (8 KB tail) could not parse it.  The line now carries numbers only; the full objects go to
    bench_detail.json.  Built here from a synthetic full result with every leg and prose notes of round-3 length."""
    b, tmp = bench
    note = "x" * 1500
    def leg(name, dtype="f64"):
        rf = {"bound": "valu", "achieved": 15.2, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.1934567, "traffic": 1.24e7, "kernel": "ismpc_a_tick_wave<double, 4, 6, true>",
              "kernel_ms": 0.0515, "kernel_ms_train": 0.0479, "algorithmic_credit": {"flops_per_launch": 4.06e9, "tflops": 78.9, "bytes_per_launch": 9.96e6, "note": note},
              "executed": {"valu_busy_frac": 0.7, "fp_share_of_valu_instructions": 0.45, "valu_issue_note": note}, "peak_note": note, "note": note,
              "lib_sha256": SHA, "pmc_lib_sha256": SHA}
        cb = {"value": 815.3, "unit": "ticks/s", "cores": 1, "kind": "reference", "sample": "s" * 400, "all_cores": {"value": 1.06e4, "cores": 16, "note": note},
              "own": {"value": 300.0, "cores": 1, "kind": "port", "sample": note}, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "nproc": 256}
        return {"name": name + " " + "n" * 150, "value": 1.31e9, "unit": "ticks/s (1 tick = ...)", "ms_per_step": 0.0499, "dtype": dtype, "qp_solves_per_s": 3.9e9,
                "regions": 23, "region_ms": {"median": 2.2, "min": 2.1, "max": 2.4}, "config": {"workload": "w" * 600, "horizon": 100, "global_batch": 65536,
                "batch_per_gpu": 65536, "active_box_fraction": 0.19}, "roofline": rf, "cpu_baseline": cb, "incl_pcie_note": note}
    full = leg("headline")
    full.update({"metric": "ISMPC QP solves/s (batch, N=100 horizon)", "n_gpus": 1, "steps": 20, "warmup": 5, "higher_is_better": True, "scaling": "strong",
                 "vs_baseline": None, "data": "synthetic", "value_incl_pcie": 3.1e8, "value_incl_pcie_pageable": 2.4e8, "latency_batch1_us": 17.2,
                 "multi_gpu": {"rccl_world": 1, "group_step_ms": 0.06, "note": note},
                 "sustained": {"value": 1.7e9, "unit": "ticks/s", "gpu_seconds": 8.2, "batch": 65536, "ticks_per_call": 2000, "calls": 105},
                 "other_configs": [leg("BASELINE configs[%d]" % k, "f32" if k % 2 else "f64") for k in range(8)]})
    text = b.compact_line(full, "bench_detail.json")
    assert len(text) <= 6000 and "\n" not in text
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "other_configs", "sustained", "multi_gpu", "detail"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["global_batch"] == 65536
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_ms_train", "algorithmic_credit", "pmc_matches_lib"):
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] == pytest.approx(0.1935, abs=1e-4) and d["roofline"]["algorithmic_credit"]["tflops"] == pytest.approx(78.9)
    for k in ("value", "unit", "cores", "kind", "sample", "all_cores", "own"):
        assert k in d["cpu_baseline"], k
    assert len(d["other_configs"]) == 8 and all(set(o) >= {"name", "value", "ms_per_step", "dtype", "roofline", "cpu_baseline"} for o in d["other_configs"])
    assert "x" * 40 not in text and '"note"' not in text                                           # no prose notes in the line
    assert d["multi_gpu"]["rccl_world"] == 1
    # a pathological number of legs still fits: the optional parts are shed, the contract keys stay
    full["other_configs"] = [leg("leg %d" % k) for k in range(40)]
    text = b.compact_line(full, "bench_detail.json")
    assert len(text) <= 6000 and json.loads(text)["roofline"]["frac"] is not None


def test_source_hash_is_by_content_not_by_path(tmp_path, monkeypatch):
    """build.py::source_sha256 -- what ties a counter summary to the code when the binary was built elsewhere -- depends on the content of the
    sources, the public headers and the flags, not on where the tree sits; the library's staleness check uses the same value."""
    import shutil
    from quadruped_gait_generation_ismpc_amd import build as B
    ref = B.source_sha256()
    assert len(ref) == 64 and B.source_sha256("-DX") != ref
    shutil.copytree(B.CSRC, tmp_path / "pkg" / "csrc"); shutil.copytree(os.path.join(B.ROOT, "include"), tmp_path / "include")
    monkeypatch.setattr(B, "CSRC", str(tmp_path / "pkg" / "csrc")); monkeypatch.setattr(B, "ROOT", str(tmp_path))
    assert B.source_sha256() == ref                                       # another directory, same content
    with open(tmp_path / "pkg" / "csrc" / "ismpc_a_wave.hpp", "a") as f:
        f.write("\n// touched\n")
    assert B.source_sha256() != ref
    lib = tmp_path / "lib.so"; lib.write_bytes(b"x")
    (tmp_path / "lib.so.flags").write_text(""); (tmp_path / "lib.so.src_sha256").write_text(ref)
    assert B._stale(str(lib), [], "") is True                             # built from other sources than the tree holds now
    (tmp_path / "lib.so.src_sha256").write_text(B.source_sha256())
    assert B._stale(str(lib), [], "") is False and B._stale(str(lib), [], "-DX") is True
