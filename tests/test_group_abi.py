"""CPU-only: the multi-GPU layer of the C ABI (include/ismpc_group.h) -- every declared symbol is exported, the shard arithmetic
(what decides which GPU owns which instance and where its records sit in the gathered buffer) for 1, 2, 3, 8 ranks incl. ragged
and tiny batches, and no group without a GPU (no CPU stand-in for the collective)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_header_symbols_are_exported(built_libs):
    from quadruped_gait_generation_ismpc_amd import group as G
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ismpc_group.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(ismpc_[a-z_0-9]+)\s*\(", txt)))
    assert declared == sorted(G.EXPORTS_GROUP)
    lib = G._l()
    for name in declared:
        assert getattr(lib, name) is not None
    assert G.UNIQUE_ID_BYTES == 128


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_range_covers_the_batch_in_rank_order(built_libs, world):
    from quadruped_gait_generation_ismpc_amd import group as G
    from quadruped_gait_generation_ismpc_amd.distributed import shard_range as py_shard
    for batch in (0, 1, 2, 5, 7, 8, 9, 1000, 1023, 65536, 65537, 131072 + 5):
        nxt, sizes = 0, []
        for r in range(world):
            first, count = G.shard_range(batch, r, world)
            assert (first, count) == py_shard(batch, r, world)           # the Python layer (torch.distributed tests) cuts the same way
            assert first == nxt and count >= 0                           # contiguous, in rank order: the gathered buffer IS the batch
            nxt += count; sizes.append(count)
        assert nxt == batch and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
        if batch % world == 0:
            assert len(set(sizes)) == 1                                  # equal shards: the in-place ncclAllGather form (sendbuff = recvbuff + rank * count)
            assert all(G.shard_range(batch, r, world)[0] == r * (batch // world) for r in range(world))


def test_shard_range_and_group_arguments_are_validated(built_libs):
    from quadruped_gait_generation_ismpc_amd import group as G
    for bad in ((-1, 0, 1), (8, -1, 2), (8, 2, 2), (8, 0, 0)):
        with pytest.raises(G.GroupError) as e:
            G.shard_range(*bad)
        assert e.value.code == -1
    lib = G._l()
    f = C.c_int()
    assert lib.ismpc_shard_range(8, 0, 2, None, C.byref(f)) == -1
    assert lib.ismpc_group_world(None) == -1 and lib.ismpc_a_group_world(None) == -1 and lib.ismpc_group_sync(None) == -1
    assert lib.ismpc_group_handle(None, 0) is None
    lib.ismpc_group_destroy(None); lib.ismpc_a_group_destroy(None)        # like free(NULL)


def test_no_group_without_a_gpu(built_libs):
    import torch
    import quadruped_gait_generation_ismpc_amd as q
    from quadruped_gait_generation_ismpc_amd import group as G, formulation_a as FA
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    p = q.default_params()
    with pytest.raises(G.GroupError) as e:
        G.Group(q.reference_plan(params=p), p, devices=[0])
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    g = FA.default_gait(FA.WALK, np.pi / 4, 0.1)
    with pytest.raises(G.GroupError) as e:
        G.GroupA(FA.default_params(FA.WALK), FA.plan(g)[1], devices=[0])
    assert e.value.code == -2
    with pytest.raises(G.GroupError) as e:
        G.Group(q.reference_plan(params=p), p, devices=[])
    assert e.value.code == -1
