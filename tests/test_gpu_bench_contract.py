"""bench.py prints ONE JSON line of at most 6 000 bytes that carries the driver's contract (metric, value, numeric roofline and
cpu_baseline), writes every leg's full object to bench_detail.json, and its numbers are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_roofline_full(rf, ms_per_step, need_counters):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "executed", "algorithmic_credit"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma", "valu") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert rf["kernel_ms"] > 0.0
    if "kernel_ms_train" in rf:                                                           # Formulation B legs: isolated launch and launch train
        assert 0.0 < rf["kernel_ms_train"] <= ms_per_step * 1.05                          # the dominant kernel's launch interval fits in the step
        assert rf["kernel_ms"] >= 0.9 * rf["kernel_ms_train"]                             # one launch alone is not shorter than its share of a train
    else:
        assert rf["kernel_ms"] <= ms_per_step * 1.05
    ac = rf["algorithmic_credit"]
    assert abs(ac["tflops"] - ac["flops_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e12) <= 1e-9 * ac["tflops"]
    if rf["achieved"] is None:
        # allowed only when the committed counter summary is missing or was collected from another build of the library (said so in the note)
        assert not need_counters, rf.get("note")
        assert rf["frac"] is None and rf["traffic"] is None and ("STALE" in rf["note"] or "no PMC summary" in rf["note"])
        return
    assert rf["executed"] is not None and rf["pmc_lib_sha256"] == rf["lib_sha256"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9
    assert 0.0 < rf["frac"] <= 1.0
    assert 0.0 < rf["executed"]["valu_issue_frac"] <= 1.0                                  # executed work cannot exceed the pipe
    assert rf["frac"] <= rf["executed"]["valu_issue_frac"] * 2.0 + 1e-9                    # FP flops are a subset of the VALU issue (FMA = 2)


def _check_cpu_full(cb):
    for k in ("value", "unit", "cores", "kind", "sample", "all_cores", "cpu_model", "nproc", "own"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0
    assert cb["all_cores"]["cores"] >= 1 and cb["all_cores"]["value"] >= 0.5 * cb["value"]
    if cb["kind"] == "reference":
        assert cb["own"]["kind"] == "port" and cb["own"]["value"] > 0                     # SURVEY 8d (i) beside (ii)


def test_bench_line_contract(built_libs):
    detail_path = os.path.join(ROOT, "bench_detail.json")
    if os.path.exists(detail_path):
        os.remove(detail_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-budget", "1.5",
                        "--min-region-ms", "20", "--sustained-seconds", "3"], capture_output=True, text=True, timeout=1100, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) <= 6000, len(lines[0])                                            # the driver keeps an 8 KB tail: round 3's 27 KB line was never parsed
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "other_configs",
              "value_incl_pcie", "latency_batch1_us", "sustained", "multi_gpu", "detail"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["global_batch"] == 65536 and d["config"]["batch_per_gpu"] == 65536    # the configuration the metric is quoted on
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) <= 1e-4 * d["value"]         # value = instances / step time (both rounded in the line)
    assert d["regions"] >= 3 and d["region_ms_median"] * d["regions"] >= 15.0               # the timed work is not a 0.3 ms blip
    assert d["value_incl_pcie"] < d["value"] and 1.0 < d["latency_batch1_us"] < 1e4
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_ms_train", "algorithmic_credit", "pmc_matches_lib"):
        assert k in rf, k
    assert rf["kernel_ms_train"] <= d["ms_per_step"] * 1.05
    if rf["pmc_matches_lib"]:
        assert 0.0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 2e-3 and rf["traffic"] > 0
    else:
        assert rf["frac"] is None and rf["achieved"] is None
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and len(cb["sample"]) <= 200
    assert d["value"] > 1000 * cb["all_cores"]["value"]                                     # the GPU path is not a CPU path in disguise
    su = d["sustained"]
    assert su["gpu_seconds"] >= 3.0 and su["value"] > 0.5 * d["value"] and su["batch"] == 65536   # the closed loop keeps the state in registers: not slower than half a per-tick launch
    assert d["multi_gpu"]["rccl_world"] == 1                                                # what RCCL itself reports for the C-ABI group of this run
    names = [o["name"] for o in d["other_configs"]]
    assert any("configs[1]" in n for n in names) and any("configs[3]" in n for n in names) and any("configs[4]" in n for n in names)
    assert any("parameter sweep" in n for n in names)
    for o in d["other_configs"]:
        for k in ("value", "ms_per_step", "dtype", "roofline", "cpu_baseline"):
            assert k in o, (o["name"], k)
        assert o["value"] > 100 * o["cpu_baseline"]["value"]
    # ---- the detail file: the full objects, with the checks that need them
    assert d["detail"] == "bench_detail.json" and os.path.exists(detail_path)
    full = json.load(open(detail_path))
    assert abs(full["value"] - d["value"]) <= 1e-6 * full["value"] and len(full["other_configs"]) == len(d["other_configs"])
    assert 0.0 <= full["config"]["active_box_fraction"] <= 1.0
    fresh = bool(rf["pmc_matches_lib"])
    _check_roofline_full(full["roofline"], full["ms_per_step"], need_counters=fresh)
    _check_cpu_full(full["cpu_baseline"])
    assert full["sustained"]["final_status_bad_index"] == 0 and full["sustained"]["final_status_error"] == 0
    for o in full["other_configs"]:
        for k in ("value", "unit", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline"):
            assert k in o, (o["name"], k)
        _check_roofline_full(o["roofline"], o["ms_per_step"], need_counters=False)
        _check_cpu_full(o["cpu_baseline"])
        assert o["value"] > 100 * o["cpu_baseline"]["all_cores"]["value"]
