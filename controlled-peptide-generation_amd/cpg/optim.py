"""Fused global-norm clip + Adam over flat f32 buffers (train_vae.py:15,39-42).

Built from the same iterable the reference hands to torch.optim.Adam - `model.vae_params()` - INCLUDING its duplicate
entry for word_emb.weight.  Parameters are re-pointed into one flat buffer (duplicates first) and get
persistent `.grad` views into one flat gradient buffer, so
  * the data-parallel gradient exchange is ONE all-reduce of the flat buffer per step (RCCL over xGMI),
  * the norm is one two-stage reduction (+ one more over the duplicated segment: it is counted twice),
  * Adam is one launch for all singly-listed parameters and two sequential launches for a doubly-listed one
    (its gradient scaled by coef^2, its step counter advancing by two) - the reference's observable behaviour.

Gradient buckets (data parallel): `buckets=[(tag, params), ...]` lays each group out contiguously at the END of the flat
buffers, in the order given.  `backward(loss)` runs the backward pass inside cpg.ops.backward_scope; when the boundary
with a bucket's tag fires (cpg.ops.GradBoundaryFn: every gradient of that bucket has been enqueued) the bucket's slice is
all-reduced asynchronously - behind the side stream that carries the decoder's deferred dW_hh product - while the encoder
BPTT still runs on the main stream; `step()` reduces what is left (embedding + encoder recurrence) and waits for all of it.
"""
import torch

from . import ops
from .ops import _p, _stream, call


class BucketReducer:
    """When and how the slices of ONE flat gradient buffer are SUM-all-reduced across data-parallel ranks.  Pure host logic (no
    kernels: FusedAdamClip owns those), so it is the same object on the GPU - where the collectives are RCCL launches issued
    from a side stream - and in the CPU tests, where gloo carries them at world 2 and 8 (tests/test_dist_gloo.py).

    bucket_range: tag -> [start, end) of the flat buffer, laid out at its END in the order the backward pass completes them;
    [0, tail_end) is everything else (final only when the backward pass ends).  `on_boundary(tag)` - called from inside the
    backward pass when the bucket's gradients are final (cpg.ops.GradBoundaryFn) - starts its asynchronous all-reduce;
    `finish()` reduces what was not started and waits for everything in flight."""

    def __init__(self, flat_g, bucket_range, tail_end, reduce_fn=None, async_reduce_fn=None, world=1):
        self.flat_g, self.bucket_range, self.tail_end = flat_g, dict(bucket_range), int(tail_end)
        self.reduce_fn, self.async_reduce_fn, self.world = reduce_fn, async_reduce_fn, int(world)
        self.inflight, self.reduced = [], set()

    @property
    def overlapped(self):
        return self.async_reduce_fn is not None and self.world > 1 and bool(self.bucket_range)

    def reset(self):
        self.inflight, self.reduced = [], set()

    def on_boundary(self, tag):
        rng = self.bucket_range.get(tag)
        if rng is None or tag in self.reduced:
            return
        self.reduced.add(tag)
        g = self.flat_g[rng[0]:rng[1]]
        if not self.flat_g.is_cuda:                      # CPU (tests over gloo): no streams, the Work object is the handle
            self.inflight.append((self.async_reduce_fn(g), None))
            return
        main = torch.cuda.current_stream()
        streams = ops.side_streams(self.flat_g.device)
        side = streams[2]                                # the deferred dW_hh accumulation of this bucket is queued there
        side.wait_stream(main)
        for s in streams:                                # ... and EVERY other side stream that carries deferred accumulations into
            if s is not side:                            # .grad (GruBiSeqFn's reverse dW_hh on side[1]): a bucket must not leave
                side.wait_stream(s)                      # before them, whichever stream its parameters' products ran on
        with torch.cuda.stream(side):
            work = self.async_reduce_fn(g)
        self.inflight.append((work, side))

    def finish(self):
        """SUM over ranks of whatever has not been reduced yet, then wait for the bucket reductions started in backward()."""
        if self.reduce_fn is None and self.async_reduce_fn is None:
            return
        sync = self.reduce_fn if self.reduce_fn is not None else (lambda t: self.async_reduce_fn(t).wait())
        if not self.reduced:
            sync(self.flat_g)
        else:
            if self.tail_end > 0:
                sync(self.flat_g[:self.tail_end])
            for tag, (a, b) in self.bucket_range.items():
                if tag not in self.reduced:
                    sync(self.flat_g[a:b])
        for work, side in self.inflight:
            if side is None:
                work.wait()
                continue
            with torch.cuda.stream(side):
                work.wait()                                   # the issuing stream waits for the collective ...
            torch.cuda.current_stream().wait_stream(side)     # ... and the optimiser's stream for the issuing stream
        self.reset()


def bucket_layout(sizes, order_tags, pad):
    """Offsets of parameters laid out in `order_tags` order (one tag or None per parameter, bucketed ones last and grouped):
    -> (offsets, {tag: (start, end)}, tail_end, total).  Host arithmetic shared by FusedAdamClip and the CPU tests."""
    offs, off, rng = [], 0, {}
    for k, tag in zip(sizes, order_tags):
        offs.append(off)
        if tag is not None:
            a, _ = rng.get(tag, (off, off))
            rng[tag] = (a, off + pad(k))
        off += pad(k)
    tail_end = min([r[0] for r in rng.values()], default=off)
    return offs, rng, tail_end, off


class FusedAdamClip:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=None, reduce_fn=None, world=1, buckets=None,
                 async_reduce_fn=None):
        """reduce_fn(tensor): in-place SUM all-reduce (blocking on the current stream).  async_reduce_fn(tensor) -> handle with
        .wait() (torch.distributed Work): enables the bucket overlap; without it buckets only shape the layout."""
        plist = list(params)
        if not plist:
            raise ValueError("no parameters")
        uniq, mult = [], {}
        for p in plist:
            if id(p) not in mult:
                uniq.append(p)
                mult[id(p)] = 0
            mult[id(p)] += 1
        dups = [p for p in uniq if mult[id(p)] > 1]
        singles = [p for p in uniq if mult[id(p)] == 1]
        in_bucket, bucketed = {}, []
        for tag, bp in (buckets or []):
            grp = [p for p in bp if id(p) in mult and mult[id(p)] == 1 and id(p) not in in_bucket]
            for p in grp:
                in_bucket[id(p)] = tag
            bucketed.append((tag, grp))
        self.order = dups + [p for p in singles if id(p) not in in_bucket] + [p for _, grp in bucketed for p in grp]
        self.mult = [mult[id(p)] for p in self.order]
        dev = self.order[0].device
        if dev.type != "cuda":
            raise ops.CpgError("FusedAdamClip needs parameters on the GPU; there is no CPU fallback")
        # every parameter starts on a 64-byte boundary of the flat buffers (the kernels take 16-byte vector loads only from
        # aligned bases); the padding holds zeros: zero gradient, zero update, no effect on the norm
        ALIGN = 16
        pad = lambda k: -(-k // ALIGN) * ALIGN
        n = sum(pad(p.numel()) for p in self.order)
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.segs = []
        off = 0
        for p in self.order:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p.data)
            p.grad = self.flat_g[off:off + k].view_as(p.data)
            self.segs.append((off, k))
            off += pad(k)
        self.n_dup = sum(k for (o, k), mm in zip(self.segs, self.mult) if mm > 1)
        # bucket tag -> [start, end) of the flat buffers (padded segment boundaries); everything before the first bucket is the tail
        _, self.bucket_range, self.tail_end, total = bucket_layout([p.numel() for p in self.order],
                                                                   [in_bucket.get(id(p)) for p in self.order], pad)
        assert total == n
        self.reducer = BucketReducer(self.flat_g, self.bucket_range, self.tail_end, reduce_fn, async_reduce_fn, world)
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.world = int(world)
        # completed optimiser iterations, ON THE DEVICE: the Adam step numbers are formed there, so that a step captured into a
        # hipGraph advances them at every replay (there is deliberately no host-side counter: it would stop under replay)
        self.iter_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.ws = torch.empty(ops.query("cpg_sumsq_workspace") // 4, device=dev, dtype=torch.float32)
        self.rng = None
        # the one-launch forms cover: at most two parameters listed more than once, all the same number (<= 4) of times
        import ctypes
        dups = [((off, pad(k)), mm) for (off, k), mm in zip(self.segs, self.mult) if mm > 1]
        self._ndup = len(dups)
        self._fused_ok = self._ndup <= 2 and all(mm <= 4 and mm == dups[0][1] for _, mm in dups)
        self._dup_off = (ctypes.c_ulonglong * 2)(*([d[0][0] for d in dups] + [0, 0])[:2])
        self._dup_len = (ctypes.c_ulonglong * 2)(*([d[0][1] for d in dups] + [0, 0])[:2])
        self._dup_mult = (ctypes.c_int * 2)(*([d[1] for d in dups] + [1, 1])[:2])

    @property
    def _reduced(self):          # tags whose all-reduce was started from a gradient boundary of the current backward pass
        return self.reducer.reduced

    def zero_grad(self):
        ops.join_deferred()
        self.flat_g.zero_()
        self.reducer.reset()

    def backward(self, loss):
        """loss.backward() in the fused form (cpg.ops.backward_scope): direct accumulation into the flat gradient buffer, the
        decoder's dW_hh on the side stream, bucket all-reduces started from the gradient boundaries."""
        cb = self.reducer.on_boundary if self.reducer.overlapped else None
        one = self._one if getattr(self, "_one", None) is not None and self._one.device == loss.device and loss.dim() == 0 else None
        if one is None and loss.dim() == 0 and loss.dtype == torch.float32:
            one = self._one = torch.ones((), device=loss.device, dtype=torch.float32)   # the root gradient, built once (not one fill per step)
        with ops.backward_scope(cb, root=loss):   # boundaries are counted on loss's own graph: same decision on every rank
            loss.backward(one)

    def _finish_reduce(self):
        self.reducer.finish()

    def grad_norm(self):
        """Pre-clip total norm as clip_grad_norm_ would return it (duplicates counted by multiplicity); device scalar."""
        return torch.sqrt(self.sumsq[0]) / self.world

    def step(self):
        ops.join_deferred()
        for p, (off, k) in zip(self.order, self.segs):  # autograd may have replaced .grad if it was None'd by the user
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * 4:
                if p.grad is not None:
                    self.flat_g[off:off + k].copy_(p.grad.reshape(-1))
                p.grad = self.flat_g[off:off + k].view_as(p.data)
        self._finish_reduce()            # SUM over ranks; the 1/world factor is folded into the update (gscale)
        gscale = 1.0 / self.world
        n = self.flat_g.numel()
        b1, b2 = self.betas
        if self._fused_ok:
            # two launches for the whole iteration: partial sums of the weighted squares, then ONE Adam launch over the flat buffer that
            # reduces the partials itself and walks the doubly-listed embedding through its two consecutive steps (csrc/optim.hip)
            part = None
            if self.max_norm is not None:
                call("cpg_sumsq_segs", _p(self.flat_g), n, self._ndup, self._dup_off, self._dup_len, self._dup_mult, _p(self.ws), _stream())
                part = self.ws
            call("cpg_adam_step_segs", _p(self.flat_p), _p(self.flat_g), _p(self.m), _p(self.v), n, float(self.lr), float(b1), float(b2),
                 float(self.eps), _p(part), _p(self.sumsq) if part is not None else None,
                 float(self.max_norm if self.max_norm is not None else 0.0), float(gscale), _p(self.iter_dev), self._ndup, self._dup_off,
                 self._dup_len, self._dup_mult, _stream())
            self._advance_counters()
            return
        sumsq = None
        if self.max_norm is not None:
            call("cpg_sumsq", _p(self.flat_g), n, 1.0, 0, _p(self.sumsq), _p(self.ws), _stream())
            for (off, k), mm in zip(self.segs, self.mult):
                if mm > 1:
                    call("cpg_sumsq", _p(self.flat_g[off:]), k, float(mm - 1), 1, _p(self.sumsq), _p(self.ws), _stream())
            sumsq = self.sumsq

        def adam(off, k, step_add, step_mult, coef_pow):
            # step number = step_mult * iter_dev + step_add, formed on the device
            call("cpg_adam_step", _p(self.flat_p[off:]), _p(self.flat_g[off:]), _p(self.m[off:]), _p(self.v[off:]), k,
                 float(self.lr), float(b1), float(b2), float(self.eps), int(step_add), _p(sumsq),
                 float(self.max_norm if self.max_norm is not None else 0.0), int(coef_pow), float(gscale),
                 _p(self.iter_dev), int(step_mult), _stream())

        i = 0
        while i < len(self.order) and self.mult[i] > 1:     # a parameter listed mult times: mult consecutive Adam steps
            off, k = self.segs[i]
            for j in range(self.mult[i]):
                adam(off, k, j + 1, self.mult[i], self.mult[i])
            i += 1
        if i < len(self.order):
            off = self.segs[i][0]
            adam(off, n - off, 1, 1, 1)
        self._advance_counters()

    def _advance_counters(self):
        """iter_dev += 1 - together with the model's Philox base when a DeviceRng was attached (`attach_rng`): one launch for both
        device-side counters of a step (cpg_step_counters_add)."""
        rng = self.rng
        if rng is not None and rng.base is not None and rng.offset:
            call("cpg_step_counters_add", _p(rng.base), int(rng.offset), _p(self.iter_dev), 1, _stream())
            rng.offset = 0
        else:
            call("cpg_step_counters_add", None, 0, _p(self.iter_dev), 1, _stream())

    def attach_rng(self, rng):
        """The model's DeviceRng: step() then also closes the step's draws (DeviceRng.end_step's work) in its counter launch."""
        self.rng = rng
