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


class _CApi:
    """The library's RCCL entry points as LibComm drives them (include/cpg_api.h: cpg_comm_*, cpg_allreduce_f32, cpg_allgatherv).
    Kept behind this small object so that tests can run LibComm's host logic - id exchange, row counts, empty ranks - at
    world > 1 over a stand-in that speaks the same contract on CPU tensors (tests/test_dist_gloo.py)."""

    def unique_id(self):
        import ctypes
        from . import ops
        buf = ctypes.create_string_buffer(128)
        ops.call("cpg_comm_unique_id", buf)
        return bytes(buf.raw)

    def init(self, uid, rank, world):
        import ctypes
        from . import ops
        comm = ctypes.c_void_p()
        ops.call("cpg_comm_init", ctypes.c_char_p(uid), rank, world, ctypes.byref(comm))
        return comm

    def allreduce_f32(self, comm, t):
        from . import ops
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        ops.call("cpg_allreduce_f32", comm, ops._p(t), t.numel(), ops._stream())

    def allgatherv(self, comm, send, counts, rank, world, out):
        """counts: bytes contributed by every rank - IDENTICAL on every rank (the C entry point's contract); send may be None
        when this rank's count is 0."""
        import ctypes
        from . import ops
        arr = (ctypes.c_size_t * world)(*[int(c) for c in counts])
        ops.call("cpg_allgatherv", comm, ops._p(send) if send is not None else None, arr, rank, world, ops._p(out), ops._stream())

    def destroy(self, comm):
        from . import ops
        ops.call("cpg_comm_destroy", comm)

    def count(self, comm):
        from . import ops
        return int(ops.query("cpg_comm_count", comm))

    def record(self):
        return torch.cuda.current_stream().record_event()


class LibComm:
    """The library's own RCCL communicator (include/cpg_api.h: cpg_comm_*, cpg_allreduce_f32, cpg_allgatherv): collectives are
    plain launches on a HIP stream of the caller's choice - no process-group stream, no Work objects.  The 128-byte unique id
    travels over the torch.distributed group that is up anyway (any backend).  Opt-in: CPG_COMM=lib (default: torch.distributed)."""

    def __init__(self, rank, world, api=None):
        self.rank, self.world = rank, world
        self.api = api if api is not None else _CApi()
        idt = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            idt = torch.frombuffer(bytearray(self.api.unique_id()), dtype=torch.uint8).clone()
        if world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            idt = idt.to(dev)
            dist.broadcast(idt, 0)
            idt = idt.cpu()
        self._comm = self.api.init(bytes(idt.numpy().tobytes()), rank, world)

    def ranks_seen(self):
        """Ranks the communicator itself reports (ncclCommCount); -1 when the entry point is unavailable."""
        return self.api.count(self._comm) if hasattr(self.api, "count") else -1

    def allreduce_sum(self, t):
        """In-place SUM on the CURRENT stream (asynchronous to the host)."""
        self.api.allreduce_f32(self._comm, t)
        return t

    def allreduce_sum_async(self, t):
        self.allreduce_sum(t)
        return _StreamWork(self.api.record())

    def allgather_rows(self, t):
        """Variable number of rows per rank -> all rows in rank order (counts first, then one grouped set of broadcasts).
        A rank may contribute ZERO rows (no accepted CLaSS proposal in its shard): the row size comes from the trailing shape,
        never from the local row count, so every rank hands cpg_allgatherv the same byte table (round-3 advisor finding)."""
        import math
        t = t.contiguous()
        row_bytes = math.prod(t.shape[1:]) * t.element_size()
        mine = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        allc = torch.zeros(self.world, dtype=torch.int64, device=t.device)
        self.api.allgatherv(self._comm, mine, [8] * self.world, self.rank, self.world, allc)
        rows = [int(x) for x in allc.cpu()]
        out = torch.empty((sum(rows),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if out.numel():
            self.api.allgatherv(self._comm, t if t.numel() else None, [r * row_bytes for r in rows], self.rank, self.world, out)
        return out

    def close(self):
        if self._comm:
            self.api.destroy(self._comm)
            self._comm = None


class _StreamWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        if self.ev is not None:
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
