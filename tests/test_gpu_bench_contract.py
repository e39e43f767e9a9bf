"""bench.py prints ONE JSON line that carries the driver's contract (metric, value, roofline, cpu_baseline) and whose
numbers are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(built_libs):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-budget", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    B = d["config"]["batch_per_gpu"]
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]           # value = instances / step time
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9
    assert 0.0 < rf["kernel_ms"] <= d["ms_per_step"] * 1.05                               # the dominant kernel fits in the step
    assert abs(rf["achieved"] - rf["algorithmic_flops_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e12) <= 1e-9 * rf["achieved"]
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0
    assert d["value"] > 1000 * cb["value"]                                                # the GPU path is not a CPU path in disguise
