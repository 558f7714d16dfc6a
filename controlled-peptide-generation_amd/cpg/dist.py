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


class LibComm:
    """The library's own RCCL communicator (include/cpg_api.h: cpg_comm_*, cpg_allreduce_f32, cpg_allgatherv): collectives are
    plain launches on a HIP stream of the caller's choice - no process-group stream, no Work objects.  The 128-byte unique id
    travels over the torch.distributed group that is up anyway (any backend).  Opt-in: CPG_COMM=lib (default: torch.distributed)."""

    def __init__(self, rank, world):
        import ctypes
        from . import ops
        self.rank, self.world = rank, world
        idt = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            ops.call("cpg_comm_unique_id", buf)
            idt = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            idt = idt.to(dev)
            dist.broadcast(idt, 0)
            idt = idt.cpu()
        self._comm = ctypes.c_void_p()
        ops.call("cpg_comm_init", ctypes.c_char_p(bytes(idt.numpy().tobytes())), rank, world, ctypes.byref(self._comm))

    def allreduce_sum(self, t):
        """In-place SUM on the CURRENT stream (asynchronous to the host)."""
        from . import ops
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        ops.call("cpg_allreduce_f32", self._comm, ops._p(t), t.numel(), ops._stream())
        return t

    def allreduce_sum_async(self, t):
        self.allreduce_sum(t)
        return _StreamWork(torch.cuda.current_stream().record_event())

    def allgather_rows(self, t):
        """Variable number of rows per rank -> all rows in rank order (counts first, then one grouped set of broadcasts)."""
        import ctypes
        from . import ops
        t = t.contiguous()
        row_bytes = t[0:1].numel() * t.element_size() if t.dim() > 1 else t.element_size()
        mine = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        allc = torch.zeros(self.world, dtype=torch.int64, device=t.device)
        eight = (ctypes.c_size_t * self.world)(*([8] * self.world))
        ops.call("cpg_allgatherv", self._comm, ops._p(mine), eight, self.rank, self.world, ops._p(allc), ops._stream())
        rows = [int(x) for x in allc.cpu()]
        out = torch.empty((sum(rows),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        counts = (ctypes.c_size_t * self.world)(*[r * row_bytes for r in rows])
        if out.numel():
            ops.call("cpg_allgatherv", self._comm, ops._p(t) if t.numel() else None, counts, self.rank, self.world, ops._p(out), ops._stream())
        return out

    def close(self):
        from . import ops
        if self._comm:
            ops.call("cpg_comm_destroy", self._comm)
            self._comm = None


class _StreamWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


_lib_comm = None


def lib_comm():
    """The LibComm of this process when CPG_COMM=lib asked for it (created on first use, after init()), else None."""
    global _lib_comm
    if _lib_comm is None and os.environ.get("CPG_COMM") == "lib" and torch.cuda.is_available():
        world, rank, _ = env_world()
        if world == 1 or dist.is_initialized():
            _lib_comm = LibComm(rank, world)
    return _lib_comm


def allreduce_sum(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        c = lib_comm()
        if c is not None and t.is_cuda and t.dtype == torch.float32:
            return c.allreduce_sum(t)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def is_initialized():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def allreduce_sum_async(t):
    """Non-blocking SUM all-reduce; the returned Work's .wait() makes the current stream (RCCL) / the host (gloo) wait."""
    c = lib_comm()
    if c is not None and t.is_cuda and t.dtype == torch.float32:
        return c.allreduce_sum_async(t)
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
