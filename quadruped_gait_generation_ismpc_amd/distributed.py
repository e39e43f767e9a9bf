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


def gather_records(local_u8, world=None, out=None, group=None, counts=None, force=False):
    """All-gather per-rank record blocks [count_r, rec] (uint8) into [sum count_r, rec] in rank order.
    Equal shard sizes take the single fused all_gather_into_tensor; ragged ones pad to the largest.
    `counts` (per-rank shard sizes, e.g. from shard_range) skips the size exchange: with it the call
    is exactly ONE collective.  A world of one returns the input unless `force` (then the collective runs over the
    one-rank group: communicator and stream order are exercised, no bytes leave the GPU)."""
    world = world if world is not None else dist.get_world_size(group)
    if world == 1 and not force:
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


class GatherPipeline:
    """The path's one collective, overlapped with compute: step k's all-gather of the output records runs on a side stream
    while step k+1's kernel runs on the launch stream.  Two local record buffers and two gathered buffers alternate
    (buffer = k & 1); before the kernel of step k may overwrite local buffer k & 1, the gather of step k-2 -- the last
    reader of that buffer -- must have completed, and nothing else is ever waited for.

        b = pipe.before_launch(k)          # waits (stream-side) for gather k-2; returns the buffer index to write
        <launch the kernel of step k into local[b] on the current stream>
        pipe.after_launch(k, local[b])     # event on the current stream -> side stream -> async all_gather_into_tensor
        ...
        pipe.drain()                       # both outstanding gathers complete (stream-side); then synchronize

    On CPU tensors (gloo, the tests) the same calls run with host-side waits."""

    def __init__(self, world, count, rec, device, host_copies=False, group=None):
        self.world, self.count, self.rec, self.group, self.host_copies = int(world), int(count), int(rec), group, bool(host_copies)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.all = [torch.empty((self.world * self.count, self.rec), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.pending = [None, None]
        self.issued = [-1, -1]                      # step whose records each gathered buffer holds (or will hold once waited for)
        if self.cuda:
            self.side = torch.cuda.Stream(self.device)
            self.done = [torch.cuda.Event(), torch.cuda.Event()]

    def before_launch(self, k):
        b = k & 1
        w = self.pending[b]
        if w is not None:
            w.wait()                                # CUDA: the current stream waits for the collective; CPU: the host does
            self.pending[b] = None
        return b

    def after_launch(self, k, local):
        b = k & 1
        assert self.pending[b] is None, "before_launch(k) must precede after_launch(k)"
        if self.cuda:
            self.done[b].record()                   # the kernel of step k, on the launch stream
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.done[b])
                self.pending[b] = dist.all_gather_into_tensor(self.all[b], local, group=self.group, async_op=True)
        else:
            src = local.cpu() if self.host_copies else local
            self.pending[b] = dist.all_gather_into_tensor(self.all[b], src.contiguous(), group=self.group, async_op=True)
        self.issued[b] = k

    def drain(self):
        for b in (0, 1):
            if self.pending[b] is not None:
                self.pending[b].wait(); self.pending[b] = None

    def gather_blocking(self, local):
        """The collective alone on the current stream (for timing it beside the kernel)."""
        src = local.cpu() if (self.host_copies and not self.cuda) else local
        dist.all_gather_into_tensor(self.all[0], src.contiguous(), group=self.group)

    def result(self, b):
        return self.all[b]

    def result_numpy(self, b, dtype):
        a = self.all[b].detach().cpu().numpy()
        return a.view(dtype).reshape(-1)
