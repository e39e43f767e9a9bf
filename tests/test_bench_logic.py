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
    return b, tmp_path


def _pmc(path, leg, batch, kernel, fl64, fl32, valu, active=None, busy=None, fetch=1000.0, write=500.0):
    c = {"SQ_INSTS_VALU": valu}
    if active is not None:
        c["SQ_ACTIVE_INST_VALU"] = active; c["SQ_BUSY_CYCLES"] = busy
    d = {"flops_f64_per_launch": fl64, "flops_f32_per_launch": fl32, "fp_insts_f64_per_launch": fl64 / 100.0, "fp_insts_f32_per_launch": fl32 / 100.0,
         "fp_share_of_valu_insts": 0.4, "hbm_bytes_per_launch": 2.0 * fetch * 1024 + write * 1024}
    json.dump({"kernel": kernel, "batch": batch, "leg": leg, "launches_per_step": 1, "counters_mean_per_launch": c, "derived": d}, open(path / f"pmc_{leg}.json", "w"))


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
    assert b.kernel_name_b(100, 65536, 256, sweep=True) == "ismpc_tick_quad_one<7, 16, 2, true>"
    assert b.kernel_name_b(100, 65536, 256, sweep=True, deferring=True) == "ismpc_tick_quad<7, 16, true>"   # ... two while instances are deferred
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
