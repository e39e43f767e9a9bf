"""CPU-only, world_size 2 over gloo: the sharding of the batch axis and the single all-gather of output
records that bench.py --gpus N uses (no GPU compute here: records are stand-ins derived from the inputs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_out(tin):
    """Deterministic 80-byte stand-in for the kernel's output record of each input."""
    from quadruped_gait_generation_ismpc_amd import TICK_OUT
    o = np.zeros(len(tin), dtype=TICK_OUT)
    o["com_pos"] = tin["com_pos"] + 0.01 * tin["com_vel"]
    o["com_vel"] = tin["com_vel"]
    o["u0"][:, 0] = tin["simulation_time"]
    o["status"] = tin["mpc_iter"]; o["iters"] = tin["footstep_counter"]
    return o


def _worker(rank, world, port, global_batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quadruped_gait_generation_ismpc_amd import workload, TICK_OUT
        from quadruped_gait_generation_ismpc_amd.distributed import shard_range, gather_records
        first, count = shard_range(global_batch, rank, world)
        tin = workload.make_batch(100, count, first_instance=first)           # no communication to build a shard
        loc = _fake_out(tin)
        loc_t = torch.from_numpy(loc.view(np.uint8).reshape(count, 80).copy())
        allt = gather_records(loc_t, world)
        got = allt.numpy().view(TICK_OUT).reshape(-1)
        full = _fake_out(workload.make_batch(100, global_batch))
        q.put((rank, first, count, got.tobytes() == full.tobytes(), len(got)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [64, 37])
def test_shards_and_single_allgather(global_batch, built_libs):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert [r[1] for r in res] == [0, res[0][2]] and sum(r[2] for r in res) == global_batch
    assert all(r[3] for r in res) and all(r[4] == global_batch for r in res)


def test_shard_ranges_partition_the_batch():
    from quadruped_gait_generation_ismpc_amd.distributed import shard_range
    for gb in (1, 7, 8, 65536, 65537):
        for world in (1, 2, 4, 8):
            pos = 0
            for r in range(world):
                f, c = shard_range(gb, r, world)
                assert f == pos and c >= 0
                pos += c
            assert pos == gb
    assert shard_range(65536, 3, 8) == (3 * 8192, 8192)


def _pipe_worker(rank, world, port, steps, count, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quadruped_gait_generation_ismpc_amd.distributed import GatherPipeline
        pipe = GatherPipeline(world, count, 80, device="cpu")
        local = [torch.zeros((count, 80), dtype=torch.uint8) for _ in range(2)]

        def records(k, r):                      # what the "kernel" of step k writes on rank r
            return ((torch.arange(count * 80, dtype=torch.int64).reshape(count, 80) * 7 + 31 * k + 101 * r) % 251).to(torch.uint8)

        ok = True
        for k in range(steps):
            b = pipe.before_launch(k)
            assert b == (k & 1)
            if k >= 2:
                # the gather of step k-2 -- the last reader of local[b] -- is complete BEFORE step k overwrites local[b]
                exp = torch.cat([records(k - 2, r) for r in range(world)])
                ok = ok and bool(torch.equal(pipe.result(b), exp)) and pipe.issued[b] == k - 2
            local[b].copy_(records(k, rank))    # stand-in for the kernel launch of step k
            pipe.after_launch(k, local[b])
            # at most two gathers are ever outstanding, one per buffer; the other buffer's gather (step k-1) is untouched
            assert pipe.pending[b] is not None
        pipe.drain()
        for k in (steps - 1, steps - 2):
            exp = torch.cat([records(k, r) for r in range(world)])
            ok = ok and bool(torch.equal(pipe.result(k & 1), exp))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_double_buffered_gather_pipeline_order(built_libs):
    """bench.py --gpus N: step k's all-gather overlaps step k+1's kernel; buffer k & 1 is only reused after the gather of
    step k-2 has completed, and every gathered block is the block of exactly one step (gloo, world 2, CPU tensors)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, 7, 33, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _run_bench(*argv):
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ISMPC_BENCH_CHILD")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, timeout=120, env=env, cwd=root)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, lines, r.stderr


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus N` as typed: the parent starts N rank processes with the variables torch.distributed.run sets,
    forwards rank 0's line and exits 0 (no GPU is touched by the self-test ranks)."""
    rc, lines, err = _run_bench("--gpus", "4", "--launcher-selftest", "-1")
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["world"] == 4 and lines[0]["rank"] == 0 and lines[0]["master"] == "127.0.0.1" and lines[0]["port"] > 0
    rc, lines, err = _run_bench("--gpus", "1", "--spawn", "--launcher-selftest", "-1")
    assert rc == 0 and lines[0]["world"] == 1, err


def test_bench_launcher_fails_when_a_rank_fails():
    """One rank exits non-zero: the others (which would wait in a collective for ever) are stopped and the launcher fails."""
    import time
    t0 = time.time()
    rc, lines, err = _run_bench("--gpus", "3", "--launcher-selftest", "2")
    assert rc == 7 and "rank 2 exited" in err
    assert time.time() - t0 < 25                       # the surviving ranks (sleeping 30 s) were stopped, not waited for
