"""Fused global-norm clip + Adam over flat f32 buffers (train_vae.py:15,39-42).

Built from the same iterable the reference hands to torch.optim.Adam - `model.vae_params()` - INCLUDING its duplicate
entry for word_emb.weight.  Parameters are re-pointed into one flat buffer (duplicates first) and get
persistent `.grad` views into one flat gradient buffer, so
  * the data-parallel gradient exchange is ONE all-reduce of the flat buffer per step (RCCL over xGMI),
  * the norm is one two-stage reduction (+ one more over the duplicated segment: it is counted twice),
  * Adam is one launch for all singly-listed parameters and two sequential launches for a doubly-listed one
    (its gradient scaled by coef^2, its step counter advancing by two) - the reference's observable behaviour.
"""
import torch

from . import ops
from .ops import _p, _stream, call


class FusedAdamClip:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=None, reduce_fn=None, world=1):
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
        self.order = dups + singles
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
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.reduce_fn, self.world = reduce_fn, int(world)
        self.steps = [0] * len(self.order)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.ws = torch.empty(ops.query("cpg_sumsq_workspace") // 4, device=dev, dtype=torch.float32)
        ops.DEFER_WGRAD = True  # this optimiser joins the deferred weight-gradient stream before touching gradients

    def zero_grad(self):
        ops.join_deferred()
        self.flat_g.zero_()

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
        if self.reduce_fn is not None:
            self.reduce_fn(self.flat_g)  # SUM over ranks; the 1/world factor is folded into the update (gscale)
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

        def adam(off, k, step, coef_pow):
            call("cpg_adam_step", _p(self.flat_p[off:]), _p(self.flat_g[off:]), _p(self.m[off:]), _p(self.v[off:]), k,
                 float(self.lr), float(b1), float(b2), float(self.eps), int(step), _p(sumsq),
                 float(self.max_norm if self.max_norm is not None else 0.0), int(coef_pow), float(gscale), _stream())

        i = 0
        while i < len(self.order) and self.mult[i] > 1:
            off, k = self.segs[i]
            for _ in range(self.mult[i]):
                self.steps[i] += 1
                adam(off, k, self.steps[i], self.mult[i])
            i += 1
        if i < len(self.order):
            off = self.segs[i][0]
            for j in range(i, len(self.order)):
                self.steps[j] += 1
            adam(off, n - off, self.steps[i], 1)
