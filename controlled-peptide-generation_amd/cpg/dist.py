"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl") - gloo on CPU for tests.

The reference has no multi-device code; this is new.  Per training step the ranks exchange
  * ONE SUM all-reduce of the flat f32 gradient buffer (FusedAdamClip.reduce_fn), and
  * the tiny batch-global statistics that keep the loss equal to the single-device value (non-PAD target count,
    two [R] random-feature sums) - losses.set_distributed.
CLaSS sampling shards z across ranks with no data-path collective; accepted rows are all-gathered at the end.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("CPG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_device(local))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def local_device(local_rank):
    """GPU index of this rank.  One process per GPU; with fewer GPUs than ranks (single-GPU functional test of the
    data-parallel path with CPG_DIST_BACKEND=gloo) ranks wrap around."""
    n = torch.cuda.device_count()
    return local_rank % n if n else 0


def allreduce_sum(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def is_initialized():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def allreduce_sum_async(t):
    """Non-blocking SUM all-reduce; the returned Work's .wait() makes the current stream (RCCL) / the host (gloo) wait."""
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_max(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def broadcast_params(params, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for p in params:
            dist.broadcast(p.data, src)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def allgather_equal(t):
    """All-gather of equal-sized row blocks -> [world*b, ...] in rank order (global batch of the full-kernel MMD)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t.contiguous())
    return torch.cat(outs, 0)


def allgather_rows(t):
    """All-gather of a variable number of rows per rank (accepted CLaSS samples): counts first, then padded payload."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    mx = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    pad[:t.shape[0]] = t
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:int(c.item())] for o, c in zip(outs, counts)], 0)
