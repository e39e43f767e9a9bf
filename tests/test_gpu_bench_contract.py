"""bench.py prints ONE JSON line that carries the driver's contract (metric, value, roofline, cpu_baseline) and whose
numbers are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_roofline(rf, ms_per_step):
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
    # achieved / frac are measured (executed flops from the committed PMC pass of this leg): present and physical
    assert rf["achieved"] is not None and rf["executed"] is not None, rf.get("note")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9
    assert 0.0 < rf["frac"] <= 1.0
    assert 0.0 < rf["executed"]["valu_issue_frac"] <= 1.0                                  # executed work cannot exceed the pipe
    assert rf["frac"] <= rf["executed"]["valu_issue_frac"] * 2.0 + 1e-9                    # FP flops are a subset of the VALU issue (FMA = 2)


def _check_cpu(cb):
    for k in ("value", "unit", "cores", "kind", "sample", "all_cores", "cpu_model", "nproc"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0
    assert cb["all_cores"]["cores"] >= 1 and cb["all_cores"]["value"] >= 0.5 * cb["value"]


def test_bench_line_contract(built_libs):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-budget", "1.5",
                        "--min-region-ms", "20"], capture_output=True, text=True, timeout=1100, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "other_configs",
              "value_incl_pcie", "latency_batch1_us"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["global_batch"] == 65536 and d["config"]["batch_per_gpu"] == 65536    # the configuration the metric is quoted on
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]         # value = instances / step time
    assert d["regions"] >= 3 and d["region_ms"]["median"] * d["regions"] >= 15.0            # the timed work is not a 0.3 ms blip
    assert 0.0 <= d["config"]["active_box_fraction"] <= 1.0
    assert d["value_incl_pcie"] < d["value"] and 1.0 < d["latency_batch1_us"] < 1e4
    _check_roofline(d["roofline"], d["ms_per_step"])
    _check_cpu(d["cpu_baseline"])
    assert d["value"] > 1000 * d["cpu_baseline"]["all_cores"]["value"]                     # the GPU path is not a CPU path in disguise
    names = [o["name"] for o in d["other_configs"]]
    assert any("configs[1]" in n for n in names) and any("configs[3]" in n for n in names) and any("configs[4]" in n for n in names)
    assert any("parameter sweep" in n for n in names)
    for o in d["other_configs"]:
        for k in ("value", "unit", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline"):
            assert k in o, (o["name"], k)
        _check_roofline(o["roofline"], o["ms_per_step"])
        _check_cpu(o["cpu_baseline"])
        assert o["value"] > 100 * o["cpu_baseline"]["all_cores"]["value"]
