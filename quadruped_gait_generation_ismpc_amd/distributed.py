"""Multi-GPU layout of a batch of gait instances.

Instances are independent, so the batch axis is cut into contiguous shards, one per rank (one process
per GPU), every read-only table is replicated by each rank's own ismpc_create, and the ONLY exchange
is one all-gather of the 80-byte output records (RCCL over xGMI on GPUs; gloo in the CPU tests).
There is no other collective on the data path.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world):
    """Contiguous shard [first, first+count) of rank `rank`; the first (global_batch % world) ranks get one more."""
    base, extra = divmod(int(global_batch), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_records(local_u8, world=None, out=None, group=None, counts=None):
    """All-gather per-rank record blocks [count_r, rec] (uint8) into [sum count_r, rec] in rank order.
    Equal shard sizes take the single fused all_gather_into_tensor; ragged ones pad to the largest.
    `counts` (per-rank shard sizes, e.g. from shard_range) skips the size exchange: with it the call
    is exactly ONE collective."""
    world = world if world is not None else dist.get_world_size(group)
    if world == 1:
        return local_u8
    n_local = int(local_u8.shape[0])
    if counts is None:
        cnt = torch.tensor([n_local], dtype=torch.int64, device=local_u8.device)
        all_cnt = torch.empty(world, dtype=torch.int64, device=local_u8.device)
        dist.all_gather_into_tensor(all_cnt, cnt, group=group)
        counts = [int(c) for c in all_cnt.cpu()]
    rec = local_u8.shape[1]
    if len(set(counts)) == 1:
        if out is None:
            out = torch.empty((world * n_local, rec), dtype=local_u8.dtype, device=local_u8.device)
        dist.all_gather_into_tensor(out, local_u8.contiguous(), group=group)
        return out
    m = max(counts)
    padded = torch.zeros((m, rec), dtype=local_u8.dtype, device=local_u8.device)
    padded[:n_local] = local_u8
    buf = torch.empty((world * m, rec), dtype=local_u8.dtype, device=local_u8.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * m: r * m + counts[r]] for r in range(world)], dim=0)
