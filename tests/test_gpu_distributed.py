"""RCCL on the one leased MI355X: a ONE-rank "nccl" group (communicator set-up, the side-stream all-gather of
GatherPipeline and its event order, result bytes) and bench.py's own launcher (`--gpus 1 --spawn`) with the collective
forced on.  No scaling is measured here -- a group of one rank moves no bytes between GPUs; the N-rank path is the same
code (tests/test_distributed.py covers its buffer order over gloo with two ranks)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ISMPC_BENCH_CHILD")}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


def test_one_rank_rccl_group_runs_the_gather_pipeline(built_libs):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_one_rank.py"), str(_free_port())],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world"] == 1
    assert d["pipeline_bytes_ok"] and d["gather_records_ok"] and d["status_ok_fraction"] > 0.5


@pytest.mark.parametrize("collective", ["abi", "torch", "abi_fails"])
def test_bench_gpus_1_through_its_own_launcher_with_the_collective(built_libs, collective):
    """`bench.py --gpus 1 --spawn --force-collective`: the launcher path of `--gpus N` with a one-rank group.  abi (default): the native
    library's own communicator (ismpc_group_create_rank from a unique id handed over the gloo control plane, ismpc_group_step_device);
    torch: torch.distributed's nccl backend through GatherPipeline (round 3's path, kept for A/B)."""
    # abi_fails: the native group cannot be built (simulated) -- every rank must agree to give it up and the torch path carries the gather
    env = _env()
    fails = collective == "abi_fails"
    if fails:
        env["ISMPC_BENCH_FAIL_NATIVE_GROUP"] = "1"; collective = "torch"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--force-collective", "--collective", "abi" if fails else collective,
                        "--only", "shard_b8192", "--no-cpu-baseline", "--no-extras", "--full-line", "--steps", "10", "--warmup", "3", "--min-region-ms", "5"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["batch_per_gpu"] == 8192 and "RCCL all-gather" in d["config"]["collective"]
    mg = d["multi_gpu"]
    assert mg["path"] == collective and mg["rccl_world"] == 1                                   # what RCCL itself reports (abi) / torch's world (torch)
    assert mg["kernel_ms"] > 0 and mg["overlapped_step_ms"] > 0
    assert ("native_group_failed" in mg) == fails
    if collective == "torch":
        assert mg["collective_ms"] is not None and mg["collective_ms"] > 0
    else:
        assert "native library" in d["config"]["collective"] and mg["rccl_version"] > 20000 and mg["exposed_collective_ms"] >= 0.0


def test_bench_formulation_a_leg_through_the_native_group(built_libs):
    """configs[4]'s shape at --gpus N gathers its records through ismpc_a_group_step_device: the one-rank form of it, fp32."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--force-collective", "--only", "config4_mc_C200", "--dtype", "f32",
                        "--no-cpu-baseline", "--full-line", "--steps", "3", "--warmup", "1", "--min-region-ms", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dtype"] == "f32" and d["config"]["batch_per_gpu"] == 16384 and d["config"]["status_nonzero"] == 0
    assert d["multi_gpu"]["path"] == "abi" and d["multi_gpu"]["rccl_world"] == 1 and d["multi_gpu"]["overlapped_step_ms"] >= d["multi_gpu"]["kernel_ms"] * 0.9
