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
        self.bucket_range, pos = {}, {id(p): i for i, p in enumerate(self.order)}
        for tag, grp in bucketed:
            if grp:
                i0, i1 = pos[id(grp[0])], pos[id(grp[-1])]
                self.bucket_range[tag] = (self.segs[i0][0], self.segs[i1][0] + pad(self.segs[i1][1]))
        self.tail_end = min([r[0] for r in self.bucket_range.values()], default=n)
        self.async_reduce_fn = async_reduce_fn
        self._inflight, self._reduced = [], set()
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.reduce_fn, self.world = reduce_fn, int(world)
        self.iter_dev = torch.zeros(1, device=dev, dtype=torch.int32)   # completed optimiser iterations, ON THE DEVICE: the Adam
        self.iters = 0                                                   # step numbers are formed there (graph-replayable steps)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.ws = torch.empty(ops.query("cpg_sumsq_workspace") // 4, device=dev, dtype=torch.float32)

    def zero_grad(self):
        ops.join_deferred()
        self.flat_g.zero_()
        self._inflight, self._reduced = [], set()

    def backward(self, loss):
        """loss.backward() in the fused form (cpg.ops.backward_scope): direct accumulation into the flat gradient buffer, the
        decoder's dW_hh on the side stream, bucket all-reduces started from the gradient boundaries."""
        cb = self._on_boundary if (self.async_reduce_fn is not None and self.world > 1 and self.bucket_range) else None
        with ops.backward_scope(cb):
            loss.backward()

    def _on_boundary(self, tag):
        rng = self.bucket_range.get(tag)
        if rng is None or tag in self._reduced:
            return
        self._reduced.add(tag)
        dev = self.flat_g.device
        main = torch.cuda.current_stream()
        side = ops.side_streams(dev)[2]      # the deferred dW_hh accumulation of this bucket is queued there
        side.wait_stream(main)
        with torch.cuda.stream(side):
            work = self.async_reduce_fn(self.flat_g[rng[0]:rng[1]])
        self._inflight.append((work, side))

    def _finish_reduce(self):
        """SUM over ranks of whatever has not been reduced yet, then wait for the bucket reductions started in backward()."""
        if self.reduce_fn is None and self.async_reduce_fn is None:
            return
        sync = self.reduce_fn if self.reduce_fn is not None else (lambda t: self.async_reduce_fn(t).wait())
        if not self._reduced:
            sync(self.flat_g)
        else:
            if self.tail_end > 0:
                sync(self.flat_g[:self.tail_end])
            for tag, (a, b) in self.bucket_range.items():
                if tag not in self._reduced:
                    sync(self.flat_g[a:b])
        cur = torch.cuda.current_stream()
        for work, side in self._inflight:
            with torch.cuda.stream(side):
                work.wait()                  # the issuing stream waits for the collective ...
            cur.wait_stream(side)            # ... and the optimiser's stream for the issuing stream
        self._inflight, self._reduced = [], set()

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
        sumsq = None
        if self.max_norm is not None:
            call("cpg_sumsq", _p(self.flat_g), n, 1.0, 0, _p(self.sumsq), _p(self.ws), _stream())
            for (off, k), mm in zip(self.segs, self.mult):
                if mm > 1:
                    call("cpg_sumsq", _p(self.flat_g[off:]), k, float(mm - 1), 1, _p(self.sumsq), _p(self.ws), _stream())
            sumsq = self.sumsq
        b1, b2 = self.betas

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
        call("cpg_counter_add_i32", _p(self.iter_dev), 1, _stream())
        self.iters += 1
