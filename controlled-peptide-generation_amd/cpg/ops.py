"""torch.autograd wrappers over the libcpg_hip.so kernels.

Every function here takes CUDA tensors and launches HIP kernels on torch's current stream; torch supplies device
memory, the stream and the autograd tape - nothing else.  There is no CPU path: tensors that are not on a GPU raise.
Reference call sites are cited per op (paths relative to the reference repo).
"""
import ctypes

import torch
from torch.autograd import Function

from ._lib import lib

PAD_IDX, UNK_IDX, START_IDX, EOS_IDX = 1, 0, 2, 3  # models/mutils.py:5-8


class CpgError(RuntimeError):
    pass


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise CpgError("cpg ops run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)
    return t.data_ptr()   # a plain int: the bindings carry argtypes (cpg/_lib.py), ctypes converts it to void* itself - a c_void_p object per
    #                       argument was 0.2 ms of the 1.5 ms of host time of a step at the reference's default sizes (tools/host_profile.py)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_id():
    """Raw handle (int) of the current stream of the current device: the key of stream-bound caches."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    """The current HIP stream of the current device as a launch argument.  torch.cuda.current_stream() costs ~9 us of python per call
    (device-index plumbing, an os.getenv inside is_available) - 45 calls per training step, 0.4 ms of its 2.5 ms of host time
    (tools/host_profile.py); the raw accessors are the same two C calls without the wrapping."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    L = lib()
    rc = getattr(L.dll, name)(*args)
    if rc != 0:
        raise CpgError(f"{name} failed (rc={rc}): {L.last_error()}")


def query(name, *args):
    return getattr(lib().dll, name)(*args)


def set_compute_mode(mode):
    """'f32' (default; f32-grade recurrent products: the parity path) or 'bf16' (BASELINE.json configs[1]/[4]: the recurrent
    products - forward / backward step products and dW_hh - round their operands to bf16 and issue one bf16 MFMA per block,
    f32 accumulation, f32 storage and master weights).  Process-wide (cpg_set_compute_mode)."""
    call("cpg_set_compute_mode", {'f32': 0, 'bf16': 1}[mode])


def get_compute_mode():
    return 'bf16' if query("cpg_get_compute_mode") == 1 else 'f32'


def set_option(name, value=None):
    """Launch-policy option of the library (KNOBS.md): cpg_set_option.  value None / '' = back to the built-in policy."""
    call("cpg_set_option", name.encode(), None if value is None else str(value).encode())


def get_option(name):
    buf = ctypes.create_string_buffer(32)
    rc = query("cpg_get_option", name.encode(), buf, 32)
    if rc < 0:
        raise CpgError("unknown option " + name)
    return buf.value.decode() if rc == 1 else None


class options:
    """`with options(gru_bwd_tile='64x32', tn_split=4): ...` - set for the block, restored afterwards (tests, A/B timing)."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.prev = {k: get_option(k) for k in self.kv}
        for k, v in self.kv.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False


_ws_cache = {}
PROFILE = None  # set to a list by bench.py to collect (family, start_event, end_event, launches, dims) records


class _prof:
    """HIP events on the CURRENT stream around a launch chain, recorded only while bench.py has PROFILE set.
    family: fwd_step | fwd_persist | bwd_step | wgrad_hh; dims: dict(T=, B=, H=, ndir=)."""

    def __init__(self, family, launches, **dims):
        self.family, self.launches, self.dims = family, launches, dims

    def __enter__(self):
        self.ev = None
        if PROFILE is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
            PROFILE.append((self.family, self.ev[0], self.ev[1], self.launches, self.dims))
        return False


GRAPH_CAPTURED = 0       # live hipGraphs (train_vae.GraphedTrainStep: graph_captured() / graph_released()) holding raw pointers into the caches below
_retired = []            # buffers superseded while a captured graph is alive: kept, a replay still writes to them


def graph_captured():
    """A capture is about to start: from here on superseded scratch buffers are parked in _retired (buffers replaced DURING the
    capture are referenced by its nodes as well)."""
    global GRAPH_CAPTURED
    GRAPH_CAPTURED += 1


def graph_released():
    """A captured graph was dropped (or its capture failed): once no graph is alive the parked buffers go back to the allocator
    (round-4 advisor finding: the flag was never reset, pinning every superseded buffer for the life of the process)."""
    global GRAPH_CAPTURED
    GRAPH_CAPTURED = max(0, GRAPH_CAPTURED - 1)
    if not GRAPH_CAPTURED:
        _retired.clear()


def _retire(old):
    """A cached scratch buffer is being replaced by a larger one.  Once a training step has been captured into a hipGraph the
    graph's kernel nodes carry the OLD buffer's address (the side-stream workspaces included: side_streams() is shared by eager
    and captured code), so dropping the last reference would hand that memory back to the caching allocator while every replay
    still writes to it.  Superseded buffers are therefore parked here while any captured graph is alive (round-3 advisor finding)."""
    if GRAPH_CAPTURED and old is not None:
        _retired.append(old)


def workspace(nbytes, device, tag=0):
    """Stream-keyed scratch buffer (grown on demand, reused across calls: launches on one stream are ordered)."""
    key = (device.index, _stream_id(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        _retire(buf)
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_side = {}
import os as _os
# run independent launch chains (encoder directions, deferred weight gradients) on side streams - never when several ranks share this GPU
# (CPG_SHARED_DEVICE, the single-GPU fallback of `bench.py --gpus N`): two processes' streams time-slice the device launch by launch, a
# 22 ms step became 150-780 ms with the side streams on
OVERLAP = _os.environ.get('CPG_NO_OVERLAP', '') == '' and not _os.environ.get('CPG_SHARED_DEVICE')


def side_streams(device, n=3):
    """Per-device pool of side HIP streams (created once)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _side:
        _side[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _side[key]


def _alloc_guard(*tensors):
    """Tensors allocated while a side stream is current but consumed on the main stream: tell the caching allocator."""
    cur = torch.cuda.current_stream()
    main = torch.cuda.default_stream()
    if cur != main:
        for t in tensors:
            if t is not None:
                t.record_stream(main)


class fork:
    """`with fork(device) as f: f.run(i, fn)` runs fn on side stream i after everything already queued on the current
    stream; leaving the block makes the current stream wait for all branches."""

    def __init__(self, device):
        self.device = device
        self.done = []

    def __enter__(self):
        self.main = torch.cuda.current_stream()
        self.ev = self.main.record_event()
        return self

    def run(self, i, fn):
        if not OVERLAP:
            return fn()
        s = side_streams(self.device)[i]
        s.wait_event(self.ev)
        with torch.cuda.stream(s):
            out = fn()
        self.done.append(s.record_event())
        return out

    def __exit__(self, *exc):
        for e in self.done:
            self.main.wait_event(e)
        return False


def _rowmajor(t):
    """(tensor, ld) for a 2-D tensor whose last dim is contiguous (row stride may exceed the width)."""
    assert t.dim() == 2
    if t.stride(1) != 1 or t.stride(0) < t.size(1):
        t = t.contiguous()
    return t, t.stride(0)


# ----------------------------------------------------------------------------------------------- dense
DIRECT_GRAD = False   # only inside backward_scope (FusedAdamClip.backward): see _grad_buf
DEFER_WGRAD = False   # only inside backward_scope: the decoder's dW_hh product runs on a side stream (GruSeqFn)
DEFER_SPLIT = _os.environ.get('CPG_DEFER_SPLIT', '')   # split-K of the DEFERRED dW_hh launch only: '' = 4/5 of the plan's, '0' = the plan's own, n = n


def _deferred_split(Mr, N, Kd, pairs=0):
    """Option override for the deferred (side-stream) dW_hh launch.  The plan of the TN product fills the chip with ONE round of
    150 KB-LDS workgroups (24 tiles x split-K 10 = 240 at config B): right for a launch that has the GPU to itself, wrong for one that
    is meant to run UNDER the main stream's small launches - those then crawl on the 16 CUs left over (profiles/r04: a 5-us
    gradient add takes 370 us).  Four fifths of the split leave 64 CUs to the main stream: 6.71 against 6.79-6.84 ms per step
    in situ (split 8; 6 / 5 / 4 / 3: 6.76 / 6.73 / 6.81 / 6.97 - longer workgroups run on into the encoder's BPTT)."""
    if DEFER_SPLIT == '0':
        return {}
    if DEFER_SPLIT:
        return {"tn_split": int(DEFER_SPLIT)}
    s = int(query("cpg_gemm_tn_split", int(Mr), int(N), int(Kd), int(pairs)))
    return {"tn_split": max(1, (4 * s) // 5)} if s >= 5 else {}
def _deferred_split_ap(R, M, N):
    """The same for the all-T planes product (cpg_gru_wgrad_hh_ap: 128 x 128 tiles, two 68 KB workgroups per CU, split over the rows so
    that one round covers the chip): four fifths of its split for the launch that runs under the main stream's launches."""
    if DEFER_SPLIT == '0':
        return {}
    if DEFER_SPLIT:
        return {"tn_split": int(DEFER_SPLIT)}
    s = int(query("cpg_pair_tn_split", int(M), int(N), int(R)))
    return {"tn_split": max(1, (4 * s) // 5)} if s >= 5 else {}


DEFER_DEC_WGRAD = _os.environ.get('CPG_DEFER_DEC_WGRAD', '1') != '0'   # the decoder's dW_hh product on the side stream (GruSeqFn(defer=True))
DEFER_ENC_WGRAD = _os.environ.get('CPG_DEFER_ENC_WGRAD', '1') != '0'   # the encoder's reverse-direction dW_hh beside the forward one (GruBiSeqFn)
DEFER_SMALL_WGRAD = _os.environ.get('CPG_DEFER_ROWC_WGRAD', '1') != '0'   # ... and so does the [z;c] block of its W_ih gradient (LinearColsFn)
BOUNDARY_CB = None    # only inside backward_scope: callable(tag) fired by GradBoundaryFn.backward (gradient buckets, cpg.optim)


def _count_boundaries(root):
    """tag -> number of GradBoundaryFn nodes reachable from `root`'s autograd graph: the boundaries THIS backward pass will reach."""
    roots = root if isinstance(root, (tuple, list)) else (root,)
    counts, seen, stack = {}, {}, [t.grad_fn for t in roots if t is not None and t.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn is None or id(fn) in seen:
            continue
        seen[id(fn)] = fn     # keeps the Python wrapper of the node alive: ids of collected wrappers are reused
        if type(fn).__name__ == 'GradBoundaryFnBackward':
            counts[fn.tag] = counts.get(fn.tag, 0) + 1
        stack.extend(nf for nf, _ in fn.next_functions)
    return counts


class backward_scope:
    """`with backward_scope(boundary_cb, root=loss): loss.backward()` - the fused-optimiser form of the backward pass: weight-gradient
    kernels accumulate straight into the parameters' existing .grad buffers (the Functions then return None for those
    inputs), the decoder's dW_hh product runs on a side stream, gradient-bucket boundaries fire `boundary_cb(tag)`.  Leaving
    the scope joins the side stream and restores plain autograd semantics (every Function returns its gradients), so
    torch.autograd.grad, parameter hooks and any other optimiser see standard behaviour outside it.
    root: the tensor(s) being differentiated - REQUIRED with a boundary_cb: the boundaries of each tag are counted ON THAT GRAPH, so
    whether and when a bucket's all-reduce starts is a function of the graph alone, identical on every rank that differentiates the
    same model (round-4 advisor finding: a process-global count of forward passes made it depend on host history - a grad-enabled
    forward that was never differentiated on ONE rank changed that rank's collective sequence)."""

    def __init__(self, boundary_cb=None, root=None):
        if boundary_cb is not None and root is None:
            raise ValueError("backward_scope: a boundary callback needs `root`, the tensor(s) being differentiated")
        self.cb, self.root = boundary_cb, root

    def __enter__(self):
        global DIRECT_GRAD, DEFER_WGRAD, BOUNDARY_CB, _boundary_left
        self.prev = (DIRECT_GRAD, DEFER_WGRAD, BOUNDARY_CB, _boundary_left)
        DIRECT_GRAD, DEFER_WGRAD, BOUNDARY_CB = True, True, self.cb
        _boundary_left = _count_boundaries(self.root) if self.cb is not None else {}
        return self

    def __exit__(self, *exc):
        global DIRECT_GRAD, DEFER_WGRAD, BOUNDARY_CB, _boundary_left
        DIRECT_GRAD, DEFER_WGRAD, BOUNDARY_CB, _boundary_left = self.prev
        join_deferred()
        return False


_boundary_left = {}   # inside a backward_scope with a callback: tag -> boundaries of the graph being differentiated not yet reached


class GradBoundaryFn(Function):
    """Identity on its tensor inputs.  Its backward runs once the gradients of ALL of them are complete - i.e. after every
    backward node downstream of the boundary has been enqueued - and then fires BOUNDARY_CB(tag): the optimiser starts the
    all-reduce of the gradient bucket that became final there while the rest of the backward pass still runs.
    A module may run more than once in the graph being differentiated (the decoder of a multi-decode loss): the callback fires
    only when the LAST boundary of the tag IN THAT GRAPH has been reached (backward_scope counts them on the graph) - the bucket's
    gradients are not final before that (round-3 advisor finding).  The forward pass keeps no state anywhere but on its own node."""

    @staticmethod
    def forward(ctx, tag, *ts):
        ctx.tag = tag
        # an output nobody differentiates (the embedding view when its consumers add their gradient straight into the parameter's
        # buffer) stays an undefined gradient: materialised zeros would travel on as a fill, an index_fill and an add per step
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for t in ts)

    @staticmethod
    def backward(ctx, *gs):
        if BOUNDARY_CB is not None:
            left = _boundary_left.get(ctx.tag, 1) - 1
            _boundary_left[ctx.tag] = left
            if left == 0:
                BOUNDARY_CB(ctx.tag)
        return (None,) + gs


def grad_boundary(tag, *ts):
    """Mark a gradient-bucket boundary on tensors that require grad (no-op on the others and outside training)."""
    if not torch.is_grad_enabled():
        return ts if len(ts) > 1 else ts[0]
    idx = [i for i, t in enumerate(ts) if t is not None and t.requires_grad]
    if idx:
        outs = GradBoundaryFn.apply(tag, *[ts[i] for i in idx])
        ts = list(ts)
        for i, o in zip(idx, outs):
            ts[i] = o
    return tuple(ts) if len(ts) > 1 else ts[0]


def _grad_buf(p):
    """The existing gradient buffer of a LEAF parameter, or None.  Inside backward_scope only (FusedAdamClip.backward): autograd
    would add a returned gradient into .grad with one elementwise kernel per parameter (AccumulateGrad: ~30 `add` launches per
    training step here); the weight-gradient kernels accumulate straight into it instead (their `accumulate` flag) and the
    Function returns None for that input.  Never for a parameter with hooks, never outside the scope."""
    if not DIRECT_GRAD or p is None or not p.is_leaf or not p.requires_grad:
        return None
    if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != p.shape:
        return None
    return g


def weight_exp(w):
    """Exponent record of a weight matrix (cpg_weight_exp: int32 [32] of partial maxima of |w|, one small launch): what the kernels that
    split weights into f16 pairs on the fly - the per-step GRU forward, cpg_linear_fwd_pairs - take the weights' power of two from, so
    that ANY finite f32 weight is covered (rounds 4-5 multiplied weights by a fixed 2^8: |w| >= 256 overflowed the f16 high half).
    A decode computes it once and hands it to every step."""
    w, ldw = _rowmajor(w)
    out = torch.empty(int(query("cpg_weight_exp_bytes")) // 4, device=w.device, dtype=torch.int32)
    call("cpg_weight_exp", _p(w), w.shape[0], w.shape[1], ldw, _p(out), _stream())
    return out


def linear_raw(x, w, b, out=None, accumulate=False, states=False):
    """states: x holds recurrent states (magnitudes O(1)) - large products then run on f16 pairs (cpg_linear_fwd_pairs; the weights'
    range is covered by their exponent record)."""
    x, ldx = _rowmajor(x)
    w, ldw = _rowmajor(w)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    if states:
        call("cpg_linear_fwd_pairs", _p(x), ldx, _p(w), ldw, _p(b), _p(out), out.stride(0), M, N, K, int(accumulate), _p(weight_exp(w)),
             _stream())
    else:
        call("cpg_linear_fwd", _p(x), ldx, _p(w), ldw, _p(b), _p(out), out.stride(0), M, N, K, int(accumulate), _stream())
    return out


class _GemmProb(ctypes.Structure):
    """CpgGemmProb of include/cpg_api.h, field for field (cpg_gemm_group_prob_bytes() pins the size at first use)."""
    _fields_ = [("A", ctypes.c_void_p * 2), ("B", ctypes.c_void_p * 2), ("lda", ctypes.c_int * 2), ("ldb", ctypes.c_int * 2),
                ("K", ctypes.c_int * 2), ("M", ctypes.c_int), ("N", ctypes.c_int), ("C", ctypes.c_void_p), ("C2", ctypes.c_void_p),
                ("ldc", ctypes.c_int), ("ldc2", ctypes.c_int), ("n_split", ctypes.c_int), ("accumulate", ctypes.c_int),
                ("bias", ctypes.c_void_p), ("wx_a", ctypes.c_void_p), ("wx_b", ctypes.c_void_p), ("pairs", ctypes.c_int), ("pad_", ctypes.c_int)]


NT, NN, TN = 0, 1, 2
_gemm_prob_checked = False


def _dp(t):
    return None if t is None else _p(t)


def gemm_prob(segs, C, M, N, bias=None, accumulate=False, C2=None, n_split=0, pairs=False, wx_a=None, wx_b=None):
    """One problem of a grouped launch.  segs: one or two (A, lda, B, ldb, K) tuples chained into the same accumulators; C (and C2 for
    columns >= n_split): 2-D row-major destinations (row stride = .stride(0))."""
    pr = _GemmProb()
    assert 1 <= len(segs) <= 2
    for i, (A, lda, B, ldb, K) in enumerate(segs):
        pr.A[i], pr.B[i], pr.lda[i], pr.ldb[i], pr.K[i] = _dp(A), _dp(B), int(lda), int(ldb), int(K)
    pr.M, pr.N = int(M), int(N)
    pr.C, pr.ldc = _dp(C), int(C.stride(0))
    if C2 is not None:
        pr.C2, pr.ldc2, pr.n_split = _dp(C2), int(C2.stride(0)), int(n_split)
    pr.accumulate = int(bool(accumulate))
    pr.bias = _dp(bias)
    pr.pairs, pr.wx_a, pr.wx_b = int(bool(pairs)), _dp(wx_a), _dp(wx_b)
    return pr


def gemm_group(form, probs):
    """Up to six independent products of one form (NT / NN / TN) in ONE launch (cpg_gemm_group, csrc/gemm.hip)."""
    global _gemm_prob_checked
    if not _gemm_prob_checked:
        assert ctypes.sizeof(_GemmProb) == int(query("cpg_gemm_group_prob_bytes")), "CpgGemmProb layout drifted from the library's"
        _gemm_prob_checked = True
    arr = (_GemmProb * len(probs))(*probs)
    call("cpg_gemm_group", int(form), len(probs), ctypes.cast(arr, ctypes.c_void_p), _stream())


def _ld(t):
    """(tensor, row stride) of a 2-D operand whose rows are contiguous (a column block of a wider matrix keeps its parent's stride)."""
    assert t.dim() == 2
    if t.stride(1) != 1:
        t = t.contiguous()
    return t, t.stride(0)


def colsum_into(x, out, accumulate=False, tag=9):
    """out[n] (+)= sum over rows of x [M, N] (cpg_colsum_f32)."""
    x, ld = _ld(x)
    M, N = x.shape
    nb = int(query("cpg_colsum_workspace_bytes", M, N))
    ws = workspace(nb, x.device, tag=tag)
    call("cpg_colsum_f32", _p(x), ld, M, N, _p(out), int(bool(accumulate)), _p(ws), nb, _stream())
    return out


def colsum_multi(xs, outs, accumulate=False):
    """outs[i][n] (+)= column sums of xs[i] [M, N] for up to four matrices of one shape, ONE single-stage launch (cpg_colsum_multi)."""
    xs = [_ld(x) for x in xs]
    M, N = xs[0][0].shape
    k = len(xs)
    px = (ctypes.c_void_p * 4)(*[_dp(x) for x, _ in xs], *([None] * (4 - k)))
    ld = (ctypes.c_int * 4)(*[int(l) for _, l in xs], *([0] * (4 - k)))
    po = (ctypes.c_void_p * 4)(*[_dp(o) for o in outs], *([None] * (4 - k)))
    call("cpg_colsum_multi", k, px, ld, M, N, po, int(bool(accumulate)), _stream())


class TokenTablesFn(Function):
    """tab_i = emb W_i[:, c0:c1]^T + b_i for n layers / directions in ONE launch (the W_ih half of nn.GRU over a V-row vocabulary:
    models/encoder.py:25-30, models/decoder.py:70-77; csrc/gemm.hip: cpg_gemm_group).  Arguments: emb [V,E]; meta = (c0, c1, leaf,
    pad_row): the columns of every W_i that the embedding multiplies, and - optionally - the LEAF embedding parameter behind `emb` with
    its padding row (emb = ZeroRowGradFn(leaf): inside backward_scope the embedding gradient is then accumulated straight into
    leaf.grad with that row skipped, instead of travelling through autograd's add / index_fill / AccumulateGrad launches); then
    (W_1, b_1, ..., W_n, b_n).  Backward: the n weight-gradient blocks as one grouped launch, the embedding gradient and the bias
    gradients by cpg_token_tables_bwd (one launch)."""

    @staticmethod
    def forward(ctx, emb, meta, *wb):
        n = len(wb) // 2
        ws, bs = wb[0::2], wb[1::2]
        c0, c1, leaf, pad_row = meta
        ctx.set_materialize_grads(False)
        embc, lde = _ld(emb)
        V, E = embc.shape
        assert c1 - c0 == E
        G = ws[0].shape[0]
        out = torch.empty(n, V, G, device=emb.device, dtype=torch.float32)
        probs = []
        for i in range(n):
            w = ws[i]
            assert w.stride(1) == 1 and w.shape[0] == G
            probs.append(gemm_prob([(embc, lde, w[:, c0:c1], w.stride(0), E)], out[i], V, G, bias=bs[i]))
        gemm_group(NT, probs)
        ctx.save_for_backward(embc, *ws)
        ctx.cols, ctx.n, ctx.leaves, ctx.emb_leaf = (c0, c1), n, (ws, bs), (leaf, pad_row)
        return tuple(out[i] for i in range(n))

    @staticmethod
    def backward(ctx, *dtabs):
        embc, *ws = ctx.saved_tensors
        c0, c1 = ctx.cols
        n = ctx.n
        V, E = embc.shape
        G = ws[0].shape[0]
        dev = embc.device
        lw, lb = ctx.leaves
        live = [i for i in range(n) if dtabs[i] is not None]
        outs = [None, None] + [None] * (2 * n)
        if not live:
            return tuple(outs)
        dts = {i: dtabs[i].contiguous() for i in live}
        gws = [_grad_buf(lw[i]) for i in range(n)]
        gbs = [_grad_buf(lb[i]) if lb[i] is not None else None for i in range(n)]
        direct = all(gws[i] is not None and (lb[i] is None or gbs[i] is not None) for i in live)
        dws, dbs = {}, {}
        for i in live:
            if not direct:
                dws[i] = torch.zeros_like(ws[i])
                dbs[i] = torch.empty(G, device=dev, dtype=torch.float32) if lb[i] is not None else None
        leaf, pad_row = ctx.emb_leaf
        gemb = _grad_buf(leaf) if leaf is not None else None
        demb, acc_emb, skip = None, 0, -1
        if gemb is not None:
            demb, acc_emb, skip = gemb, 1, (int(pad_row) if pad_row is not None else -1)
        elif ctx.needs_input_grad[0]:
            demb = torch.empty(V, E, device=dev, dtype=torch.float32)
        nl = len(live)
        if V > 32 or E > 256 or nl > 4:
            # shapes the two-launch kernel does not cover (large vocabularies / embeddings): one product pair per table
            for i in live:
                d = gws[i] if direct else dws[i]
                nbw = query("cpg_linear_bwd_weight_workspace", V, G, E)
                wsp = workspace(nbw, dev)
                call("cpg_linear_bwd_weight", _p(dts[i]), G, _p(embc), int(embc.stride(0)), _p(d[:, c0:c1]), int(d.stride(0)),
                     _p(gbs[i] if direct else dbs[i]), V, G, E, int(direct), _p(wsp), wsp.numel(), _stream())
            dx = None
            if demb is not None:
                dx = torch.zeros(V, E, device=dev, dtype=torch.float32)
                for i in live:
                    call("cpg_linear_bwd_input", _p(dts[i]), G, _p(ws[i][:, c0:c1]), int(ws[i].stride(0)), _p(dx), E, V, G, E, 1, _stream())
                if gemb is not None:
                    if skip >= 0:
                        dx[skip].zero_()
                    gemb.add_(dx)
            outs[0] = None if gemb is not None else dx
            if not direct:
                for i in live:
                    outs[2 + 2 * i], outs[3 + 2 * i] = dws[i], dbs[i]
            return tuple(outs)
        arr = lambda xs: (ctypes.c_void_p * 4)(*xs, *([None] * (4 - nl)))
        iarr = lambda xs: (ctypes.c_int * 4)(*xs, *([0] * (4 - nl)))
        dst = [gws[i] if direct else dws[i] for i in live]
        nb = int(query("cpg_token_tables_bwd_workspace", nl, V, G, E))
        wsp = workspace(nb, dev, tag=12)
        call("cpg_token_tables_bwd", nl, V, G, E, arr([_dp(dts[i]) for i in live]), arr([_dp(ws[i][:, c0:c1]) for i in live]),
             iarr([int(ws[i].stride(0)) for i in live]), _p(embc), int(embc.stride(0)), arr([_dp(d[:, c0:c1]) for d in dst]),
             iarr([int(d.stride(0)) for d in dst]), int(direct), arr([_dp(gbs[i] if direct else dbs[i]) for i in live]), int(direct), _p(demb),
             int(demb.stride(0)) if demb is not None else E, acc_emb, skip, _p(wsp), nb, _stream())
        outs[0] = None if gemb is not None else demb
        if not direct:
            for i in live:
                outs[2 + 2 * i], outs[3 + 2 * i] = dws[i], dbs[i]
        return tuple(outs)


def weight_exp2(w1, w2):
    """ONE exponent record for two matrices (their joint largest magnitude: cpg_weight_exp2)."""
    w1, l1 = _rowmajor(w1)
    w2, l2 = _rowmajor(w2)
    out = torch.empty(int(query("cpg_weight_exp_bytes")) // 4, device=w1.device, dtype=torch.int32)
    call("cpg_weight_exp2", _p(w1), w1.shape[0], w1.shape[1], l1, _p(w2), w2.shape[0], w2.shape[1], l2, _p(out), _stream())
    return out


def transpose_pad(src, Rpad=None, out=None):
    """out[c][r] = src[r][c], zeros for rows R <= r < Rpad (cpg_transpose_pad): a k-rows operand as K-contiguous, slab-padded rows."""
    src, lds = _ld(src)
    R, C = src.shape
    Rpad = R if Rpad is None else int(Rpad)
    if out is None:
        out = torch.empty(C, Rpad, device=src.device, dtype=torch.float32)
    call("cpg_transpose_pad", _p(src), lds, R, C, _p(out), int(out.stride(0)), Rpad, _stream())
    return out


class EncoderHeadsFn(Function):
    """(mu, logvar) = (h Wmu^T + bmu, h Wlv^T + blv) with h = [hf | hr] - the encoder's two final states, never concatenated - in ONE
    launch (models/encoder.py:35-36,46-51; cpg_gemm_group: two problems of two chained segments each).  Backward: dh = dmu Wmu + dlv Wlv
    as one chained product whose columns go straight to (dhf, dhr), the two weight gradients as one grouped launch, the two bias
    gradients as column sums.  Where the shapes allow (batch and widths multiples of 32, aligned rows) all three products run the
    direct-to-LDS loop on f16 pairs (csrc/gemm.hip: gemm_group_dl_kernel<., ., 4>) - the k-rows operands of the two backward products
    are transposed into K-contiguous, slab-padded rows first (bandwidth passes of a few MB)."""

    @staticmethod
    def forward(ctx, hf, hr, wmu, bmu, wlv, blv):
        ctx.set_materialize_grads(False)
        hfc, ldf = _ld(hf)
        hrc, ldr = (None, 0) if hr is None else _ld(hr)
        B, H = hfc.shape
        Z = wmu.shape[0]
        Hr = hrc.shape[1] if hrc is not None else 0
        assert wmu.shape[1] == H + Hr and wmu.stride(1) == 1 and wlv.stride(1) == 1
        out = torch.empty(2, B, Z, device=hf.device, dtype=torch.float32)
        # f16-pair form: states (|h| <= 1) against weights scaled by ONE power of two for both heads (their joint largest magnitude)
        fast = (B % 32 == 0 and H % 32 == 0 and Hr % 32 == 0 and B >= 64 and Z >= 32
                and _os.environ.get("CPG_HEADS_EXACT", "") == "")
        wx = weight_exp2(wmu, wlv) if fast else None
        probs = []
        for k, (w, b) in enumerate(((wmu, bmu), (wlv, blv))):
            segs = [(hfc, ldf, w[:, :H], w.stride(0), H)]
            if hrc is not None:
                segs.append((hrc, ldr, w[:, H:], w.stride(0), Hr))
            probs.append(gemm_prob(segs, out[k], B, Z, bias=b, pairs=fast, wx_b=wx))
        gemm_group(NT, probs)
        ctx.save_for_backward(hfc, hrc, wmu, wlv, wx)
        ctx.leaves = (wmu, bmu, wlv, blv)
        ctx.fast = fast
        return out[0], out[1]

    @staticmethod
    def backward(ctx, dmu, dlv):
        hf, hr, wmu, wlv, wx = ctx.saved_tensors
        B, H = hf.shape
        Hr = hr.shape[1] if hr is not None else 0
        Z = wmu.shape[0]
        dev = hf.device
        if dmu is None and dlv is None:
            return None, None, None, None, None, None
        pair = _take_padded_pair(dmu, dlv) if ctx.fast else None
        if dmu is None:
            dmu = torch.zeros(B, Z, device=dev)
        if dlv is None:
            dlv = torch.zeros(B, Z, device=dev)
        lw = ctx.leaves
        g = [_grad_buf(p) for p in lw]
        direct = all(x is not None for x in g)
        dwmu = g[0] if direct else torch.empty_like(wmu)
        dwlv = g[2] if direct else torch.empty_like(wlv)
        dbmu = g[1] if direct else torch.empty(Z, device=dev, dtype=torch.float32)
        dblv = g[3] if direct else torch.empty(Z, device=dev, dtype=torch.float32)
        need_dh = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dhf = torch.empty(B, H, device=dev, dtype=torch.float32) if need_dh else None
        dhr = torch.empty(B, Hr, device=dev, dtype=torch.float32) if (need_dh and hr is not None) else None
        if ctx.fast:
            Zp = -(-Z // 32) * 32
            if pair is None:      # gradients that did not come from LatentFn.backward: pad them here
                pair = torch.zeros(B, 2, Zp, device=dev, dtype=torch.float32)
                pair[:, 0, :Z].copy_(dmu)
                pair[:, 1, :Z].copy_(dlv)
            d2 = pair.view(B, 2 * Zp)
            wxd = weight_exp(d2)                      # ONE power of two for the gradient pair (its largest magnitude)
            if need_dh:
                # dh = [dmu | dlv] [Wmu ; Wlv]: NT over the transposed, row-padded weights
                wT = torch.empty(2, H + Hr, Zp, device=dev, dtype=torch.float32)
                transpose_pad(wmu, Zp, out=wT[0])
                transpose_pad(wlv, Zp, out=wT[1])
                gemm_group(NT, [gemm_prob([(d2, 2 * Zp, wT[0], Zp, Zp), (d2[:, Zp:], 2 * Zp, wT[1], Zp, Zp)], dhf, B, H + Hr, C2=dhr, n_split=H,
                                          pairs=True, wx_a=wxd, wx_b=wx)])
            # dW_s = d_s^T h: NT over the transposed gradient pair [2 Zp, B] and the transposed states [H + Hr, B]
            dT = transpose_pad(d2)
            hT = torch.empty(H + Hr, B, device=dev, dtype=torch.float32)
            transpose_pad(hf, out=hT[:H])
            if hr is not None:
                transpose_pad(hr, out=hT[H:])
            gemm_group(NT, [gemm_prob([(dT[s * Zp:], B, hT, B, B)], dw, Z, H + Hr, accumulate=direct, pairs=True, wx_a=wxd)
                            for s, dw in ((0, dwmu), (1, dwlv))])
            colsum_multi([d2[:, :Z], d2[:, Zp:Zp + Z]], [dbmu, dblv], accumulate=direct)
        else:
            dmu, ldm = _ld(dmu)
            dlv, ldl = _ld(dlv)
            if need_dh:
                gemm_group(NN, [gemm_prob([(dmu, ldm, wmu, wmu.stride(0), Z), (dlv, ldl, wlv, wlv.stride(0), Z)], dhf, B, H + Hr,
                                          C2=dhr, n_split=H)])
            probs = []
            for dy, ldy, dw in ((dmu, ldm, dwmu), (dlv, ldl, dwlv)):
                probs.append(gemm_prob([(dy, ldy, hf, hf.stride(0), B)], dw[:, :H], Z, H, accumulate=direct))
                if hr is not None:
                    probs.append(gemm_prob([(dy, ldy, hr, hr.stride(0), B)], dw[:, H:], Z, Hr, accumulate=direct))
            gemm_group(TN, probs)
            colsum_multi([dmu, dlv], [dbmu, dblv], accumulate=direct)
        if direct:
            return dhf, dhr, None, None, None, None
        return dhf, dhr, dwmu, dbmu, dwlv, dblv


class LinearFn(Function):
    """y = x W^T + b (nn.Linear: models/encoder.py:35-36,50-51; also the token-table and [z;c] projections that
    replace the W_ih half of nn.GRU at models/decoder.py:70-77)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        ctx.leaves = (w, b)
        return linear_raw(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy, lddy = _rowmajor(dy)
        xx, ldx = _rowmajor(x)
        ww, ldw = _rowmajor(w)
        M, K = xx.shape
        N = ww.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            call("cpg_linear_bwd_input", _p(dy), lddy, _p(ww), ldw, _p(dx), K, M, N, K, 0, _stream())
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            gw, gb = _grad_buf(ctx.leaves[0]), _grad_buf(ctx.leaves[1])
            direct = gw is not None and (not ctx.has_b or gb is not None)
            dw = gw if direct else torch.empty(N, K, device=dy.device, dtype=torch.float32)
            db = (gb if direct else torch.empty(N, device=dy.device, dtype=torch.float32)) if ctx.has_b else None
            nb = query("cpg_linear_bwd_weight_workspace", M, N, K)
            ws = workspace(nb, dy.device)
            call("cpg_linear_bwd_weight", _p(dy), lddy, _p(xx), ldx, _p(dw), K, _p(db), M, N, K, int(direct), _p(ws), ws.numel(),
                 _stream())
            if direct:
                dw = db = None
        return dx, dw, db


class LinearColsFn(Function):
    """y = x W[:, c0:c1]^T + b on a column block of a LEAF weight: the two halves of nn.GRU's W_ih at models/decoder.py:70-77 (token
    table from the embedding columns, row constant from the [z;c] columns).  Slicing the parameter in autograd instead costs, per
    slice and step, a parameter-sized fill and a copy (SliceBackward) plus the add that joins the two; here the weight-gradient
    product writes its block of the parameter's gradient in place."""

    @staticmethod
    def forward(ctx, x, w, b, c0, c1):
        ctx.save_for_backward(x, w)
        ctx.cols = (int(c0), int(c1))
        ctx.has_b = b is not None
        ctx.leaves = (w, b)
        return linear_raw(x, w[:, c0:c1], b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        c0, c1 = ctx.cols
        dy, lddy = _rowmajor(dy)
        xx, ldx = _rowmajor(x)
        ww, ldw = _rowmajor(w)
        M, K = xx.shape
        N = ww.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            call("cpg_linear_bwd_input", _p(dy), lddy, _p(ww[:, c0:c1]), ldw, _p(dx), K, M, N, K, 0, _stream())
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            gw, gb = _grad_buf(ctx.leaves[0]), _grad_buf(ctx.leaves[1])
            direct = gw is not None and (not ctx.has_b or gb is not None)
            dw = gw if direct else torch.zeros_like(ww)
            db = (gb if direct else torch.empty(N, device=dy.device, dtype=torch.float32)) if ctx.has_b else None
            nb = query("cpg_linear_bwd_weight_workspace", M, N, K)
            if direct and DEFER_WGRAD and OVERLAP and DEFER_SMALL_WGRAD and M >= 1024:
                # nothing downstream of the backward pass reads this product (it lands in the parameter's gradient buffer): queue it on
                # the side stream BEHIND the decoder's deferred dW_hh instead of in front of the launches that lead to the encoder's
                # BPTT.  On the main stream it ran beside that 240-workgroup product and crawled on the CUs left over (417 us for
                # 3.2 GFLOP at config B, profiles/r03), holding up everything queued behind it.
                side = side_streams(dy.device)[2]
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ws = workspace(nb, dy.device)
                    call("cpg_linear_bwd_weight", _p(dy), lddy, _p(xx), ldx, _p(dw[:, c0:c1]), dw.stride(0), _p(db), M, N, K, 1,
                         _p(ws), ws.numel(), _stream())
                    _pending_events.append(side.record_event())
                torch.autograd.Variable._execution_engine.queue_callback(join_deferred)
                for t in (dy, xx):
                    t.record_stream(side)
                return dx, None, None, None, None
            ws = workspace(nb, dy.device)
            call("cpg_linear_bwd_weight", _p(dy), lddy, _p(xx), ldx, _p(dw[:, c0:c1]), dw.stride(0), _p(db), M, N, K, int(direct),
                 _p(ws), ws.numel(), _stream())
            if direct:
                dw = db = None
        return dx, dw, db, None, None


class Linear2Fn(Function):
    """y = x1 W[:, :K1]^T + x2 W[:, K1:]^T + b  (input projection of an upper biGRU layer: its input is the
    concatenation of the lower layer's two directions, which are two separate state slabs here)."""

    @staticmethod
    def forward(ctx, x1, x2, w, b):
        ctx.save_for_backward(x1, x2, w)
        K1 = x1.shape[1]
        y = linear_raw(x1, w[:, :K1], b, states=True)      # x1, x2: state slabs of the lower layer
        linear_raw(x2, w[:, K1:], None, out=y, accumulate=True, states=True)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, w = ctx.saved_tensors
        K1, K2 = x1.shape[1], x2.shape[1]
        dy, lddy = _rowmajor(dy)
        M, N = dy.shape
        dw = torch.empty_like(w)
        db = torch.empty(N, device=dy.device, dtype=torch.float32)
        dxs = []
        for x, k0, kk in ((x1, 0, K1), (x2, K1, K2)):
            xx, ldx = _rowmajor(x)
            wv = w[:, k0:k0 + kk]
            dx = torch.empty(M, kk, device=dy.device, dtype=torch.float32)
            call("cpg_linear_bwd_input", _p(dy), lddy, _p(wv), w.stride(0), _p(dx), kk, M, N, kk, 0, _stream())
            nb = query("cpg_linear_bwd_weight_workspace", M, N, kk)
            ws = workspace(nb, dy.device)
            call("cpg_linear_bwd_weight", _p(dy), lddy, _p(xx), ldx, _p(dw[:, k0:]), dw.stride(0),
                 _p(db) if k0 == 0 else None, M, N, kk, 0, _p(ws), ws.numel(), _stream())
            dxs.append(dx)
        return dxs[0], dxs[1], dw, db


def planes_ok(R, K, N):
    """The plane-image forms of an nn.Linear-shaped product (csrc/planes.hip) cover R rows x contraction K x N outputs."""
    return _os.environ.get("CPG_NO_LINEAR_PLANES", "") == "" and bool(query("cpg_planes_ok", int(R), int(K), int(N)))


def pair_rows(x1, x2=None):
    """f16-pair image (uint8 tensor) of x = [x1 | x2]: rows [R, C1 (+ C2)] f32 -> [R][2 C] f16 (cpg_pair_rows); no gradient."""
    x1c, ld1 = _rowmajor(x1)
    R, C1 = x1c.shape
    x2c, ld2, C2 = (None, 0, 0) if x2 is None else (*_rowmajor(x2), x2.shape[1])
    img = torch.empty(int(query("cpg_pair_rows_bytes", R, C1 + C2)), device=x1.device, dtype=torch.uint8)
    call("cpg_pair_rows", _p(x1c), ld1, C1, _p(x2c), ld2, C2, R, _p(img), _stream())
    return img


class Linear2PlanesFn(Function):
    """Linear2Fn on f16-pair plane images (csrc/planes.hip; round 5): y = [x1 | x2] W^T + b over T B rows with every product
    conversion-free - ximg is the image of [x1 | x2] (ops.pair_rows: built ONCE per layer, shared by the layer's two directions and by the
    weight gradient), the incoming gradient is imaged once per direction and feeds both dX (one launch for both halves) and dW.
    gates: 3 (GRU) | 4 (LSTM) blocks of the output."""

    @staticmethod
    def forward(ctx, x1, x2, ximg, w, b, gates):
        R, K1, K2 = x1.shape[0], x1.shape[1], (x2.shape[1] if x2 is not None else 0)   # x2 None: one source (a decoder's upper layer)
        N, K = w.shape
        assert K == K1 + K2 and N % gates == 0
        wc = w.contiguous()
        y = torch.empty(R, N, device=w.device, dtype=torch.float32)
        sc = workspace(int(query("cpg_weight_image_bytes", N, K)), w.device, tag=7)
        call("cpg_linear_fwd_planes", _p(ximg), R, K, _p(wc), K, _p(b), _p(y), N, N, 0, _p(sc), sc.numel(), _stream())
        ctx.save_for_backward(ximg, wc)
        ctx.dims = (R, K1, K2, N, int(gates))
        return y

    @staticmethod
    def backward(ctx, dy):
        ximg, w = ctx.saved_tensors
        R, K1, K2, N, G = ctx.dims
        K, H = K1 + K2, N // G
        dev = dy.device
        dy, lddy = _rowmajor(dy)
        off = (ctypes.c_int * 4)(*[q * H for q in range(G)], *([0] * (4 - G)))
        gp = torch.empty(int(query("cpg_grad_planes_bytes", R, H, G)), device=dev, dtype=torch.uint8)
        call("cpg_grad_planes", _p(dy), lddy, R, H, G, off, _p(gp), _stream())
        dx = torch.empty(R, K, device=dev, dtype=torch.float32)
        sc = workspace(int(query("cpg_weight_image_bytes", K, N)), dev, tag=7)
        call("cpg_linear_bwd_input_planes", _p(gp), R, H, G, _p(w), K, _p(dx), K, K, 0, _p(sc), sc.numel(), _stream())
        dw = torch.empty_like(w)
        ws = workspace(int(query("cpg_linear_bwd_weight_planes_workspace", R, H, G, K)), dev)
        call("cpg_linear_bwd_weight_planes", _p(gp), R, H, G, _p(ximg), K, _p(dw), K, 0, _p(ws), ws.numel(), _stream())
        db = torch.empty(N, device=dev, dtype=torch.float32)
        nb = int(query("cpg_colsum_workspace_bytes", R, N))
        ws2 = workspace(nb, dev, tag=8)
        call("cpg_colsum_f32", _p(dy), lddy, R, N, _p(db), 0, _p(ws2), nb, _stream())
        return dx[:, :K1], (dx[:, K1:] if K2 else None), None, dw, db, None


class MaskedLinear2Fn(Function):
    """y = (x1 .* keep1*scale) W[:, :K1]^T (+ (x2 .* keep2*scale) W[:, K1:]^T) + b: nn.Dropout in front of the input projection of an
    upper encoder layer - nn.GRU(dropout=p_dropout) drops the (concatenated) output of every layer but the last in train mode
    (models/encoder.py:25-30).  x1 / x2 are the lower layer's two state slabs (x2, keep2 None for a unidirectional encoder); the
    keep masks are uint8 0/1 with their x's shape - injected by the caller or drawn from the model's streams."""

    @staticmethod
    def forward(ctx, x1, keep1, x2, keep2, scale, w, b):
        xs = [(x1.contiguous(), keep1.contiguous())] + ([(x2.contiguous(), keep2.contiguous())] if x2 is not None else [])
        M, N = xs[0][0].shape[0], w.shape[0]
        y = torch.empty(M, N, device=w.device, dtype=torch.float32)
        k0 = 0
        for i, (x, keep) in enumerate(xs):
            assert keep.dtype == torch.uint8 and keep.shape == x.shape
            kk = x.shape[1]
            call("cpg_linear_masked_fwd", _p(x), kk, _p(keep), float(scale), _p(w[:, k0:]), w.stride(0), _p(b) if i == 0 else None,
                 _p(y), N, M, N, kk, int(i > 0), _stream())
            k0 += kk
        ctx.save_for_backward(w, *[t for pair in xs for t in pair])
        ctx.scale, ctx.two, ctx.has_b = float(scale), x2 is not None, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        w, *rest = ctx.saved_tensors
        xs = [(rest[0], rest[1])] + ([(rest[2], rest[3])] if ctx.two else [])
        dy, lddy = _rowmajor(dy)
        M, N = dy.shape
        dw = torch.empty_like(w)
        db = torch.empty(N, device=dy.device, dtype=torch.float32) if ctx.has_b else None
        dxs, k0 = [], 0
        for i, (x, keep) in enumerate(xs):
            kk = x.shape[1]
            dx = torch.empty(M, kk, device=dy.device, dtype=torch.float32)
            call("cpg_linear_masked_bwd_input", _p(dy), lddy, _p(w[:, k0:]), w.stride(0), _p(keep), ctx.scale, _p(dx), kk, M, N, kk,
                 _stream())
            nb = query("cpg_linear_bwd_weight_workspace", M, N, kk)
            ws = workspace(nb, dy.device)
            call("cpg_linear_masked_bwd_weight", _p(dy), lddy, _p(x), kk, _p(keep), ctx.scale, _p(dw[:, k0:]), dw.stride(0),
                 _p(db) if i == 0 else None, M, N, kk, 0, _p(ws), ws.numel(), _stream())
            dxs.append(dx)
            k0 += kk
        return dxs[0], None, (dxs[1] if ctx.two else None), None, None, dw, db


class SkipAddFn(Function):
    """y[t*B + b] = x[t*B + b] Wx^T + sz[b]: the decoder's skip connection rnn_out := skip_weight_x(rnn_out) + skip_weight_z([z;c])
    (models/decoder.py:48-51,80-81,103-105; both bias-free) with sz = skip_weight_z([z;c]) formed once per row by the caller - the
    second term is constant over time.  x [T*B,H] time-major step outputs, sz [B,H].  Backward: dx = dy Wx, dWx = dy^T x,
    dsz = the sum of dy over time (column sums of dy viewed as [T, B*H])."""

    @staticmethod
    def forward(ctx, x, w, sz, T):
        x, sz = x.contiguous(), sz.contiguous()
        B, H = sz.shape
        assert x.shape == (T * B, H)
        y = sz.unsqueeze(0).expand(T, B, H).contiguous().view(T * B, H)    # data movement; the product accumulates onto it
        linear_raw(x, w, None, out=y, accumulate=True)
        ctx.save_for_backward(x, w)
        ctx.dims, ctx.leaf = (T, B, H), w
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        T, B, H = ctx.dims
        dy = dy.contiguous()
        dev = dy.device
        dx = torch.empty(T * B, H, device=dev, dtype=torch.float32)
        call("cpg_linear_bwd_input", _p(dy), H, _p(w), w.stride(0), _p(dx), H, T * B, H, H, 0, _stream())
        gw = _grad_buf(ctx.leaf)
        dw = gw if gw is not None else torch.empty(H, H, device=dev, dtype=torch.float32)
        nb = query("cpg_linear_bwd_weight_workspace", T * B, H, H)
        ws = workspace(nb, dev)
        call("cpg_linear_bwd_weight", _p(dy), H, _p(x), H, _p(dw), H, None, T * B, H, H, int(gw is not None), _p(ws), ws.numel(),
             _stream())
        dsz = torch.empty(B, H, device=dev, dtype=torch.float32)
        nb = query("cpg_colsum_workspace_bytes", T, B * H)
        ws = workspace(nb, dev)
        call("cpg_colsum_f32", _p(dy), B * H, T, B * H, _p(dsz), 0, _p(ws), ws.numel(), _stream())
        return dx, (None if gw is not None else dw), dsz, None


def tag_emb(t, leaf, pad_row):
    """Remember, on the tensor object, the leaf embedding parameter (and its padding row) a graph-side view of it stands for."""
    t._cpg_emb_leaf = (leaf, pad_row)
    return t


def emb_leaf(t):
    return getattr(t, "_cpg_emb_leaf", (None, None))


class ZeroRowGradFn(Function):
    """Identity whose backward zeroes one row: nn.Embedding(padding_idx=PAD) gives <pad> no gradient (models/model.py:47)."""

    @staticmethod
    def forward(ctx, w, row):
        ctx.row = row
        ctx.set_materialize_grads(False)   # no consumer returned a gradient (they added it straight into .grad): nothing to do, not a zero fill
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        key = (g.device, ctx.row)
        idx = _ROW_IDX.get(key)
        if idx is None:   # built once per process: a host -> device copy inside a step would stall the enqueue
            idx = _ROW_IDX[key] = torch.tensor([ctx.row], device=g.device, dtype=torch.int64)
        return g.index_fill(0, idx, 0.0), None


_ROW_IDX = {}


def tokens_prepare(ids, wd_mask=None):
    """ids int64 [B,T] -> int32 [T,B] time-major, WordDropout applied (models/decoder.py:117-133)."""
    B, T = ids.shape
    ids = ids.contiguous()
    tok = torch.empty(T, B, device=ids.device, dtype=torch.int32)
    if wd_mask is not None:
        wd_mask = wd_mask.to(torch.uint8).contiguous()
    call("cpg_tokens_prepare", _p(ids), _p(wd_mask), B, T, UNK_IDX, _p(tok), _stream())
    return tok


class Transpose01Fn(Function):
    """[d0,d1,inner] -> [d1,d0,inner] contiguous (time-major logits back to the reference's [B,T,V])."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        d0, d1, inner = x.shape
        y = torch.empty(d1, d0, inner, device=x.device, dtype=torch.float32)
        call("cpg_transpose01_f32", _p(x), d0, d1, inner, _p(y), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        d1, d0, inner = g.shape
        y = torch.empty(d0, d1, inner, device=g.device, dtype=torch.float32)
        call("cpg_transpose01_f32", _p(g), d1, d0, inner, _p(y), _stream())
        return y


def transpose01_u8(x):
    x = x.contiguous()
    d0, d1, inner = x.shape
    y = torch.empty(d1, d0, inner, device=x.device, dtype=torch.uint8)
    call("cpg_transpose01_u8", _p(x), d0, d1, inner, _p(y), _stream())
    return y


# ----------------------------------------------------------------------------------------------- GRU
_persist_scratch = {}   # (kind, device, stream, B, H) -> [scratch tensor, T it was sized for, pinned error word, copy event]
import os as _os2
SHARED_DEVICE = bool(_os2.environ.get("CPG_SHARED_DEVICE"))   # ranks share this GPU: persistent kernels cannot own every CU


def persistent_rows(H):
    """Batch rows ONE launch of the whole-sequence persistent forward kernel (csrc/gru_persist.hip) covers at width H on this
    device; 0 when the width is not covered (or the path is off)."""
    return 0 if SHARED_DEVICE else int(query("cpg_gru_persistent_rows", int(H)))


def persistent_fits(B, H):
    """The persistent forward path covers this shape: in one launch, or - rows being independent recurrences - in consecutive
    launches over row ranges when the batch is wider than one launch holds."""
    return B > 0 and persistent_rows(H) >= 256 and B * H * 6 * 64 <= (3 << 30)


def lstm_persistent_fits(B, H):
    """Whole-sequence persistent LSTM forward kernel (csrc/lstm_persist.hip) covers this shape on this device."""
    return not SHARED_DEVICE and bool(query("cpg_lstm_persistent_fits", int(B), int(H)))


def _persist_entry(kind, T, B, H, dev):
    """Scratch of the persistent launches on the current stream: ONE buffer per (kind, device, stream, B, H), grown to the largest
    T seen; with it a pinned, host-mapped error word that a timed-out wave sets directly - looked at here, before the next launch
    on this scratch, without any stream operation."""
    key = (kind, dev.index, _stream_id(), B, H)
    ent = _persist_scratch.get(key)
    if ent is None or ent[1] < T:
        nb = query("cpg_gru_persistent_scratch_bytes" if kind == "gru" else "cpg_lstm_persistent_scratch_bytes", T, B, H)
        host = ent[2] if ent is not None else torch.zeros(1, dtype=torch.int32).pin_memory()
        _retire(ent[0] if ent is not None else None)
        ent = _persist_scratch[key] = [torch.zeros(nb, dtype=torch.uint8, device=dev), T, host]   # counters + error word + slots
    if ent[2][0] != 0:
        _persist_failed(kind)      # an EARLIER launch on this scratch timed out
    return ent


def _persist_failed(kind):
    raise CpgError("persistent %s kernel: an inter-workgroup wait timed out (its workgroups were not co-resident: another process "
                   "or kernel held CUs); outputs of that launch are NaN-poisoned.  cpg.ops.set_option('%s_persist', 0) selects the "
                   "per-step kernels" % (kind.upper(), kind))


def _pair_scratch(B, H, ndir, dev, lstm=False):
    """Scratch of the f16-pair backward step (cpg_gru_bwd_pair_bytes / cpg_lstm_bwd_pair_bytes): [ndir, bytes] uint8, or None where
    that step does not cover the shape / compute mode (the exact-f32 product then)."""
    nb = int(query("cpg_lstm_bwd_pair_bytes", int(B), int(H))) if lstm else int(query("cpg_gru_bwd_pair_bytes", int(B), int(H), int(ndir)))
    if nb == 0:
        return None
    t = torch.empty(ndir, (nb + 255) // 256 * 256, device=dev, dtype=torch.uint8)
    return t if ndir == 2 else t[0]


AP_VMAX = 31   # largest token table the plane-reading input-side reductions take (csrc/gru.hip: DM_VMAX)


def _ap_scratch(T, B, H, ndir, dev, lstm=False, V=1):
    """Scratch of the all-T planes form of the f16-pair BPTT chain (cpg_gru_ap_bytes / cpg_lstm_ap_bytes: kept gate-gradient planes of
    every step, their exponent tables, the state planes): [ndir, bytes] uint8, or None where the form does not cover the shape / mode.
    V: rows of the layer's token table - the form's input-side reductions (cpg_*_dgi_reduce_ap) take 1..AP_VMAX of them; a larger
    vocabulary keeps the round-4 form, whose reductions have one-hot GEMM / by-token fallbacks (round-5 advisor finding)."""
    if not 0 < int(V) <= AP_VMAX:
        return None
    nb = int(query("cpg_lstm_ap_bytes", int(T), int(B), int(H))) if lstm else int(query("cpg_gru_ap_bytes", int(T), int(B), int(H), int(ndir)))
    if nb == 0:
        return None
    t = torch.empty(ndir, (nb + 255) // 256 * 256, device=dev, dtype=torch.uint8)
    return t if ndir == 2 else t[0]


def gates_dtype(B, H, ragged=False):
    """Element type of a GRU sequence's saved gates [T,4,B,H]: bf16 in the bf16 compute mode on dense batches that the
    direct-to-LDS backward step covers (cpg_gru_gates_bf16: the BPTT epilogue is HBM-bound there), f32 otherwise."""
    return torch.bfloat16 if query("cpg_gru_gates_bf16", B, H, int(bool(ragged))) == 1 else torch.float32


def dg_dtype(B, H, V, ragged=False):
    """Element type of a GRU sequence's gate gradients dG [T,B,4H]: bf16 in the bf16 compute mode where every consumer of dG has a
    bf16 form (cpg_gru_dg_bf16: dense batch, H % 128 == 0, B % 128 == 0, token table of 0 < V <= 31 rows), f32 otherwise."""
    return torch.bfloat16 if query("cpg_gru_dg_bf16", int(B), int(H), int(bool(ragged)), int(V)) == 1 else torch.float32


def _check_gates(gates, B, H, ragged=False):
    if gates is not None and gates.dtype != gates_dtype(B, H, ragged):
        raise CpgError("saved gates are %s but the library now expects %s: compute mode / options changed between the forward "
                       "and the backward pass of a sequence" % (gates.dtype, gates_dtype(B, H, ragged)))


def gru_seq_fwd_persistent(T, B, H, reverse, w_hh, b_hh, tok, tab, rowc, dense, hs, gates):
    _check_gates(gates, B, H)
    ent = _persist_entry("gru", T, B, H, hs.device)
    rows = persistent_rows(H)
    for r0 in range(0, B, rows):     # one launch when the batch fits, else row ranges back to back on this stream
        call("cpg_gru_seq_fwd_persistent", T, B, H, int(reverse), _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), _p(dense),
             _p(hs), _p(gates), r0, min(B, r0 + rows), _p(ent[0]), ctypes.c_void_p(ent[2].data_ptr()), _stream())


def lstm_seq_fwd_persistent(T, B, H, reverse, w_hh, b_hh, tok, tab, rowc, dense, hs, cs, gates):
    ent = _persist_entry("lstm", T, B, H, hs.device)
    call("cpg_lstm_seq_fwd_persistent", T, B, H, int(reverse), _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), _p(dense),
         _p(hs), _p(cs), _p(gates), _p(ent[0]), ctypes.c_void_p(ent[2].data_ptr()), _stream())


def check_persistent():
    """Raise if any in-kernel wait of a persistent launch has timed out.  Synchronises the device first, so that launches still
    running are covered: call it where a host synchronisation is acceptable (logging iterations, the end of an inference entry
    point, bench, tests); the hot path itself only glances at the host-mapped error words between launches."""
    if _persist_scratch:
        torch.cuda.synchronize()
    for (kind, *_), ent in list(_persist_scratch.items()):
        if ent[2][0] != 0:
            _persist_failed(kind)


class GruSeqFn(Function):
    """One direction of one GRU layer over the whole sequence (torch.nn.GRU at models/encoder.py:25-30,42 and
    models/decoder.py:40-41,77).  Returns the state slab [(T+1),B,H] (layout in include/cpg_api.h)."""

    @staticmethod
    def forward(ctx, tok, tab, rowc, dense, h0, w_hh, b_hh, T, reverse, defer=False, step_rows=None, tail=False):
        """step_rows: optional device int32 [T] of live-row counts per step (length-sorted batch, see cpg_api.h).
        tail (forward direction): return the step outputs only - slots 1..T as one [T,B,H] tensor - so that their gradient
        arrives as it is consumed (time-aligned), not through autograd's slice backward (a slab-sized fill and a copy)."""
        dev = w_hh.device
        ctx.tail = bool(tail)
        assert not (tail and reverse)
        # deferred mode: the 80-GFLOP dW_hh product of this sequence runs on a side stream and is added straight into the
        # parameters' existing .grad buffers, overlapping with the rest of the backward pass (only with FusedAdamClip,
        # which joins before it reads the gradients)
        ctx.defer_req = (w_hh, b_hh) if defer else None
        H = w_hh.shape[1]
        B = tok.shape[1] if tok is not None else (rowc.shape[0] if rowc is not None else dense.shape[1])
        w_hh_c, b_hh_c = w_hh.contiguous(), b_hh.contiguous()
        tab_c = tab.contiguous() if tab is not None else None
        rowc_c = rowc.contiguous() if rowc is not None else None
        dense_c = dense.contiguous() if dense is not None else None
        # ragged batches leave the slots of dead (t,row) pairs untouched: they are read by the vocabulary projection and
        # by the dW_hh product (against zero gradients), so they must hold finite numbers
        hs = (torch.zeros if step_rows is not None else torch.empty)(T + 1, B, H, device=dev, dtype=torch.float32)
        slot0 = T if reverse else 0
        if h0 is None:
            hs[slot0].zero_()
        else:
            hs[slot0].copy_(h0)
        need_grad = any(t is not None and t.requires_grad for t in (tab, rowc, dense, h0, w_hh, b_hh))
        gates = torch.empty(T, 4, B, H, device=dev, dtype=gates_dtype(B, H, step_rows is not None)) if need_grad else None
        _alloc_guard(hs, gates)
        if step_rows is None and persistent_fits(B, H):
            with _prof("fwd_persist", 1, T=T, B=B, H=H, ndir=1):
                gru_seq_fwd_persistent(T, B, H, reverse, w_hh_c, b_hh_c, tok, tab_c, rowc_c, dense_c, hs, gates)
        else:
            with _prof("fwd_step", T, T=T, B=B, H=H, ndir=1):
                call("cpg_gru_seq_fwd", T, B, H, int(reverse), _p(w_hh_c), _p(b_hh_c), _p(tok), _p(tab_c), _p(rowc_c),
                     _p(dense_c), _p(hs), _p(gates), 0, B, _p(step_rows), _p(weight_exp(w_hh_c)), _stream())
        ctx.save_for_backward(tok, w_hh_c, hs, gates)
        ctx.step_rows = step_rows
        ctx.dims = (T, B, H, bool(reverse))
        ctx.V = tab.shape[0] if tab is not None else 0
        ctx.has = (tab is not None, rowc is not None, dense is not None, h0 is not None)
        return hs[1:] if tail else hs

    @staticmethod
    def backward(ctx, ghs):
        tok, w_hh, hs, gates = ctx.saved_tensors
        T, B, H, reverse = ctx.dims
        dev = ghs.device
        ghs = ghs.contiguous()
        BH = B * H
        flat = ghs.view(-1)
        # time-aligned gradients on the step outputs: slots 1..T (forward) / 0..T-1 (reverse)
        dhs_ext = flat if ctx.tail else (flat[BH:] if not reverse else flat[:T * BH])
        has_tab, has_rowc, has_dense, has_h0 = ctx.has
        step_rows = ctx.step_rows
        # ragged batch: gradient rows of dead (t,row) pairs are not written but are read by the reductions below
        # bf16 gradient storage (bf16 compute mode): decided HERE, together with the saved gates' type, and handed to every consumer
        dgt = dg_dtype(B, H, ctx.V if has_tab else 0, step_rows is not None) if (gates is not None and gates.dtype == torch.bfloat16) else torch.float32
        dgb = int(dgt == torch.bfloat16)
        # all-T planes form (cpg_gru_ap_bytes): sequences with a token table and no dense input term (their input-side gradients are
        # consumed by the plane-reading reductions only); the recurrent dG blocks then exist only as the kept f16-pair planes
        # (bf16 compute mode with bf16 gradient storage: the same entry points keep that mode's dG and add a bf16 copy of the states,
        # so that its dW_hh product runs the conversion-free loop on one bf16 plane per operand)
        ap = _ap_scratch(T, B, H, 1, dev, V=ctx.V) if (step_rows is None and has_tab and not has_dense and gates is not None
                                              and (dgb or gates.dtype == torch.float32)) else None
        scratch = torch.empty(2, B, H, device=dev, dtype=torch.float32)
        dh0 = torch.empty(B, H, device=dev, dtype=torch.float32) if has_h0 else None
        wT = torch.empty(H, 3 * H, device=dev, dtype=torch.float32)  # receives W_hh^T for the direct-to-LDS step kernel
        _check_gates(gates, B, H, step_rows is not None)
        if ap is not None:
            pair = None
            dG = torch.empty(T, B, 4 * H, device=dev, dtype=dgt) if dgb else torch.empty(T, B, H, device=dev, dtype=torch.float32)   # f32-grade: dn_pre only
            with _prof("bwd_step", T + (1 if has_h0 else 0), T=T, B=B, H=H, ndir=1, ap=1):
                call("cpg_gru_seq_bwd_ap", T, B, H, int(reverse), _p(w_hh), _p(hs), _p(gates), _p(dhs_ext), None, _p(dG),
                     _p(scratch), _p(dh0), _p(wT), _p(ap), _stream())
        else:
            dG = (torch.zeros if step_rows is not None else torch.empty)(T, B, 4 * H, device=dev, dtype=dgt)
            pair = _pair_scratch(B, H, 1, dev) if (step_rows is None and not dgb) else None   # f16-pair form of that step
            with _prof("bwd_step", T + (1 if has_h0 else 0), T=T, B=B, H=H, ndir=1):
                call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(w_hh), _p(hs), _p(gates), _p(dhs_ext), None, _p(dG),
                     _p(scratch), _p(dh0), 0, B, _p(step_rows), _p(wT), _p(pair), dgb, _stream())
        if has_h0 and not ctx.tail:
            dh0 = dh0 + (ghs[T] if reverse else ghs[0])
        nb = query("cpg_gru_wgrad_workspace", T, B, H, max(ctx.V, 1))
        ws = workspace(nb, dev)
        dtab = torch.empty(ctx.V, 3 * H, device=dev, dtype=torch.float32) if has_tab else None
        drowc = torch.empty(B, 3 * H, device=dev, dtype=torch.float32) if has_rowc else None
        # with a token table, its gradient and the column sums of dG (= the b_hh gradient) come out of one pass over dG
        dsum = torch.empty(4 * H, device=dev, dtype=torch.float32) if has_tab else None
        if ap is not None and not dgb:
            call("cpg_gru_dgi_reduce_ap", T, B, H, _p(ap), _p(dG), _p(tok), ctx.V, _p(dtab), _p(dsum), _p(drowc), 0, _p(ws), ws.numel(),
                 _stream())
        elif has_tab or has_rowc:
            call("cpg_gru_dgi_reduce", T, B, H, _p(dG), _p(tok), ctx.V, _p(dtab), _p(dsum), _p(drowc), 0, _p(ws), ws.numel(), dgb,
                 _stream())
        dl = ctx.defer_req
        defer = dl if (dl is not None and DEFER_WGRAD and DEFER_DEC_WGRAD and OVERLAP and _grad_buf(dl[0]) is not None and _grad_buf(dl[1]) is not None) else None
        if defer is not None:
            db_hh = dsum[:3 * H] if has_tab else None
            side = side_streams(dev)[2]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ws2 = workspace(nb, dev)
                # DEFER_SPLIT: split-K of the deferred launch only = its workgroup count (24 tiles x S at config B).  The default
                # plan's 240 workgroups hold 150 KB of LDS and every register of 240 CUs: whatever the main stream launches meanwhile
                # crawls on the 16 CUs left (profiles/r04: a 5-us gradient add takes 370 us there).  Fewer, longer workgroups leave
                # whole CUs to the main stream's small launches.
                if ap is not None:
                    with _prof("wgrad_hh", 1, T=T, B=B, H=H, ndir=1, ap=1, side=1), options(**_deferred_split_ap(T * B, 3 * H, H)):
                        call("cpg_gru_wgrad_hh_ap", T, B, H, _p(ap), _p(dG) if dgb else None, _p(defer[0].grad), 1, _p(ws2), ws2.numel(), _stream())
                else:
                    with _prof("wgrad_hh", 1, T=T, B=B, H=H, ndir=1, side=1), options(**_deferred_split(T * B, 3 * H, H, pair is not None)):
                        call("cpg_gru_wgrad_hh", T, B, H, int(reverse), _p(dG), _p(hs), _p(defer[0].grad),
                             None if has_tab else _p(defer[1].grad), 1, _p(ws2), ws2.numel(), _p(pair), dgb, _stream())
                if has_tab:
                    defer[1].grad.add_(db_hh)
                _pending_events.append(side.record_event())
            # the accumulation into .grad runs on the side stream: make the stream that called backward() wait for it when
            # the backward pass ends, so ANY reader of .grad after loss.backward() (clip_grad_norm_, another optimiser, a
            # test) sees the finished gradient - not only FusedAdamClip, which joins explicitly
            torch.autograd.Variable._execution_engine.queue_callback(join_deferred)
            for t in (dG, hs, db_hh, pair, ap):
                if t is not None:
                    t.record_stream(side)
            dw_hh = db_hh = None
        else:
            dw_hh = torch.empty(3 * H, H, device=dev, dtype=torch.float32)
            db_hh = dsum[:3 * H] if has_tab else torch.empty(3 * H, device=dev, dtype=torch.float32)
            if ap is not None:
                with _prof("wgrad_hh", 1, T=T, B=B, H=H, ndir=1, ap=1):
                    call("cpg_gru_wgrad_hh_ap", T, B, H, _p(ap), _p(dG) if dgb else None, _p(dw_hh), 0, _p(ws), ws.numel(), _stream())
            else:
                with _prof("wgrad_hh", 1, T=T, B=B, H=H, ndir=1):
                    call("cpg_gru_wgrad_hh", T, B, H, int(reverse), _p(dG), _p(hs), _p(dw_hh), None if has_tab else _p(db_hh), 0, _p(ws),
                         ws.numel(), _p(pair), dgb, _stream())
        ddense = None
        if has_dense:
            # input-side gate gradients are columns {0..2H, 3H..4H} of dG (layout only; upper encoder layers)
            ddense = torch.cat([dG[:, :, :2 * H], dG[:, :, 3 * H:]], 2).float()
        return None, dtab, drowc, ddense, dh0, dw_hh, db_hh, None, None, None, None, None


class GruBiSeqFn(Function):
    """Both directions of one biGRU layer with ONE launch per time step for the pair (cpg_gru_biseq_fwd/_bwd): twice the
    work per launch amortises the fixed per-launch phases.  Returns (slab_fwd, slab_rev), layouts as GruSeqFn."""

    @staticmethod
    def forward(ctx, tok, tab_f, tab_r, dense_f, dense_r, w_hh_f, b_hh_f, w_hh_r, b_hh_r, T, finals=False):
        """finals: return only the two final states (forward slot T, reverse slot 0) - what a top encoder layer is read for.
        Their gradients then enter the backward recurrence directly (dh_last), instead of through slab-sized zero tensors
        that autograd's select/slice backward would allocate, fill and copy into (and every backward launch would read)."""
        dev = w_hh_f.device
        ctx.finals = bool(finals)
        H = w_hh_f.shape[1]
        B = tok.shape[1] if tok is not None else dense_f.shape[1]
        cont = lambda t: t.contiguous() if t is not None else None
        wf, bf, wr, br = cont(w_hh_f), cont(b_hh_f), cont(w_hh_r), cont(b_hh_r)
        tf, tr, df, dr = cont(tab_f), cont(tab_r), cont(dense_f), cont(dense_r)
        # one allocation [reverse | forward]: the two zero initial states (slot T of the reverse slab, slot 0 of the forward one) are
        # adjacent - one fill launch
        hs2 = torch.empty(2 * (T + 1), B, H, device=dev, dtype=torch.float32)
        hs_r, hs_f = hs2[:T + 1], hs2[T + 1:]
        hs2[T:T + 2].zero_()
        need_grad = any(t is not None and t.requires_grad for t in (tab_f, tab_r, dense_f, dense_r, w_hh_f, b_hh_f, w_hh_r, b_hh_r))
        g_f = torch.empty(T, 4, B, H, device=dev, dtype=gates_dtype(B, H)) if need_grad else None
        g_r = torch.empty(T, 4, B, H, device=dev, dtype=gates_dtype(B, H)) if need_grad else None
        if persistent_fits(B, H):
            # two persistent launches back to back on ONE stream (each needs all of its workgroups co-resident)
            with _prof("fwd_persist", 2, T=T, B=B, H=H, ndir=1):
                gru_seq_fwd_persistent(T, B, H, False, wf, bf, tok, tf, None, df, hs_f, g_f)
                gru_seq_fwd_persistent(T, B, H, True, wr, br, tok, tr, None, dr, hs_r, g_r)
        else:
            with _prof("fwd_step", T, T=T, B=B, H=H, ndir=2):
                call("cpg_gru_biseq_fwd", T, B, H, _p(wf), _p(bf), _p(wr), _p(br), _p(tok), _p(tf), _p(tr), _p(df), _p(dr),
                     _p(hs_f), _p(hs_r), _p(g_f), _p(g_r), _p(weight_exp(wf)), _p(weight_exp(wr)), _stream())
        ctx.save_for_backward(tok, wf, wr, hs_f, hs_r, g_f, g_r)
        ctx.leaves = (w_hh_f, w_hh_r)
        ctx.dims = (T, B, H)
        ctx.V = tab_f.shape[0] if tab_f is not None else 0
        ctx.has_tab, ctx.has_dense = tab_f is not None, dense_f is not None
        if finals:
            return hs_f[T], hs_r[0]
        return hs_f, hs_r

    @staticmethod
    def backward(ctx, g_hs_f, g_hs_r):
        tok, wf, wr, hs_f, hs_r, gt_f, gt_r = ctx.saved_tensors
        T, B, H = ctx.dims
        dev = hs_f.device
        BH = B * H
        last_f = last_r = None
        if ctx.finals:
            ext_f = ext_r = None
            last_f = g_hs_f.contiguous() if g_hs_f is not None else None
            last_r = g_hs_r.contiguous() if g_hs_r is not None else None
        else:
            ext_f = g_hs_f.contiguous().view(-1)[BH:] if g_hs_f is not None else None      # slots 1..T
            ext_r = g_hs_r.contiguous().view(-1)[:T * BH] if g_hs_r is not None else None  # slots 0..T-1
        dgt = dg_dtype(B, H, ctx.V if ctx.has_tab else 0) if (gt_f is not None and gt_f.dtype == torch.bfloat16) else torch.float32
        dgb = int(dgt == torch.bfloat16)   # bf16 gradient storage (bf16 compute mode; see GruSeqFn.backward)
        # all-T planes form (see GruSeqFn.backward): token-table layers only
        ap = _ap_scratch(T, B, H, 2, dev, V=ctx.V) if (ctx.has_tab and not ctx.has_dense and (dgb or gt_f.dtype == torch.float32)) else None
        sc = torch.empty(2, 2, B, H, device=dev, dtype=torch.float32)
        wT = torch.empty(2, H, 3 * H, device=dev, dtype=torch.float32)
        _check_gates(gt_f, B, H)
        if ap is not None:
            pair = None
            dG_f = torch.empty(T, B, 4 * H, device=dev, dtype=dgt) if dgb else torch.empty(T, B, H, device=dev, dtype=torch.float32)   # f32-grade: dn_pre only
            dG_r = torch.empty_like(dG_f)
            with _prof("bwd_step", T, T=T, B=B, H=H, ndir=2, ap=1):
                call("cpg_gru_biseq_bwd_ap", T, B, H, _p(wf), _p(wr), _p(hs_f), _p(hs_r), _p(gt_f), _p(gt_r), _p(ext_f), _p(ext_r),
                     _p(last_f), _p(last_r), _p(dG_f), _p(dG_r), _p(sc[0]), _p(sc[1]), _p(wT[0]), _p(wT[1]), _p(ap[0]), _p(ap[1]), _stream())
        else:
            dG_f = torch.empty(T, B, 4 * H, device=dev, dtype=dgt)
            dG_r = torch.empty(T, B, 4 * H, device=dev, dtype=dgt)
            pair = _pair_scratch(B, H, 2, dev) if not dgb else None
            with _prof("bwd_step", T, T=T, B=B, H=H, ndir=2):
                call("cpg_gru_biseq_bwd", T, B, H, _p(wf), _p(wr), _p(hs_f), _p(hs_r), _p(gt_f), _p(gt_r), _p(ext_f), _p(ext_r),
                     _p(last_f), _p(last_r), _p(dG_f), _p(dG_r), _p(sc[0]), _p(sc[1]), _p(wT[0]), _p(wT[1]),
                     _p(pair[0]) if pair is not None else None, _p(pair[1]) if pair is not None else None, dgb, _stream())
        nb = query("cpg_gru_wgrad_workspace", T, B, H, max(ctx.V, 1))
        ws = workspace(nb, dev)
        outs = []

        def wgrad(rev, dG, hs, dst, acc, wsx):
            if ap is not None:
                with _prof("wgrad_hh", 1, T=T, B=B, H=H, ndir=1, ap=1):
                    call("cpg_gru_wgrad_hh_ap", T, B, H, _p(ap[rev]), _p(dG) if dgb else None, _p(dst), acc, _p(wsx), wsx.numel(), _stream())
            else:
                with _prof("wgrad_hh", 1, T=T, B=B, H=H, ndir=1):
                    call("cpg_gru_wgrad_hh", T, B, H, rev, _p(dG), _p(hs), _p(dst), None, acc, _p(wsx), wsx.numel(),
                         _p(pair[rev]) if pair is not None else None, dgb, _stream())
        for rev, dG, hs in ((0, dG_f, hs_f), (1, dG_r, hs_r)):
            gw = _grad_buf(ctx.leaves[rev]) if ctx.has_tab else None
            dw = gw if gw is not None else torch.empty(3 * H, H, device=dev, dtype=torch.float32)
            dtab = None
            if ctx.has_tab:
                if rev == 1 and gw is not None and DEFER_WGRAD and OVERLAP and DEFER_ENC_WGRAD and ap is None:
                    # the reverse direction's dW_hh product beside the forward direction's, on a side stream: either launch alone
                    # is bound by the latency of its operand loads (two workgroups per CU), together they fill each other's stalls.
                    # (Not the all-T planes product: its loop is conversion-free and fills the CUs by itself - side by side the two
                    # launches only time-slice them, 5.10 against 5.05 ms per step in situ, round 5.)
                    side = side_streams(dev)[1]
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        ws2 = workspace(nb, dev)
                        wgrad(rev, dG, hs, gw, 1, ws2)
                        _pending_events.append(side.record_event())
                    torch.autograd.Variable._execution_engine.queue_callback(join_deferred)
                    for t in (dG, hs, pair, ap):
                        if t is not None:
                            t.record_stream(side)
                else:
                    wgrad(rev, dG, hs, dw, int(gw is not None), ws)
                if gw is not None:
                    dw = None
                dtab = torch.empty(ctx.V, 3 * H, device=dev, dtype=torch.float32)
                dsum = torch.empty(4 * H, device=dev, dtype=torch.float32)
                if ap is not None and not dgb:
                    call("cpg_gru_dgi_reduce_ap", T, B, H, _p(ap[rev]), _p(dG), _p(tok), ctx.V, _p(dtab), _p(dsum), None, 0, _p(ws), ws.numel(),
                         _stream())
                else:
                    call("cpg_gru_dgi_reduce", T, B, H, _p(dG), _p(tok), ctx.V, _p(dtab), _p(dsum), None, 0, _p(ws), ws.numel(), dgb,
                         _stream())
                db = dsum[:3 * H]
            else:
                db = torch.empty(3 * H, device=dev, dtype=torch.float32)
                call("cpg_gru_wgrad_hh", T, B, H, rev, _p(dG), _p(hs), _p(dw), _p(db), 0, _p(ws), ws.numel(),
                     _p(pair[rev]) if pair is not None else None, 0, _stream())
            ddense = torch.cat([dG[:, :, :2 * H], dG[:, :, 3 * H:]], 2).float() if ctx.has_dense else None
            outs.append((dtab, ddense, dw, db))
        (dtab_f, dd_f, dw_f, db_f), (dtab_r, dd_r, dw_r, db_r) = outs
        return None, dtab_f, dtab_r, dd_f, dd_r, dw_f, db_f, dw_r, db_r, None, None


_pending_events = []


def join_deferred():
    """Make the current stream wait for every deferred weight-gradient accumulation (called by the optimiser)."""
    if not _pending_events:
        return
    cur = torch.cuda.current_stream()
    while _pending_events:
        cur.wait_event(_pending_events.pop())


def gru_step(tok, tab, rowc, h_prev, h_out, w_hh, b_hh, wx=None):
    """One decode step's recurrent part (GRUDecoder.forward_sample, models/decoder.py:86-99); inference only.
    wx: weight_exp(w_hh) - a decode loop computes it once and passes it to every step (None: computed here, one more launch)."""
    B, H = h_prev.shape
    if wx is None:
        wx = weight_exp(w_hh)
    call("cpg_gru_step_fwd", B, H, _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), _p(h_prev), _p(h_out), _p(wx), _stream())
    return h_out


class LstmSeqFn(Function):
    """One direction of one LSTM layer (torch.nn.LSTM semantics; not a reference component - SURVEY F2).
    Returns the hidden-state slab [(T+1),B,H]; the cell slab stays internal."""

    @staticmethod
    def forward(ctx, tok, tab, rowc, dense, h0, c0, w_hh, b_hh, T, reverse):
        dev = w_hh.device
        H = w_hh.shape[1]
        B = tok.shape[1] if tok is not None else (rowc.shape[0] if rowc is not None else dense.shape[1])
        w_hh_c, b_hh_c = w_hh.contiguous(), b_hh.contiguous()
        tab_c = tab.contiguous() if tab is not None else None
        rowc_c = rowc.contiguous() if rowc is not None else None
        dense_c = dense.contiguous() if dense is not None else None
        hs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        cs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        slot0 = T if reverse else 0
        hs[slot0].zero_() if h0 is None else hs[slot0].copy_(h0)
        cs[slot0].zero_() if c0 is None else cs[slot0].copy_(c0)
        need_grad = any(t is not None and t.requires_grad for t in (tab, rowc, dense, h0, c0, w_hh, b_hh))
        gates = torch.empty(T, 4, B, H, device=dev, dtype=torch.float32) if need_grad else None
        if lstm_persistent_fits(B, H):
            with _prof("lstm_fwd_persist", 1, T=T, B=B, H=H, ndir=1):
                lstm_seq_fwd_persistent(T, B, H, reverse, w_hh_c, b_hh_c, tok, tab_c, rowc_c, dense_c, hs, cs, gates)
        else:
            with _prof("lstm_fwd_step", T, T=T, B=B, H=H, ndir=1):
                call("cpg_lstm_seq_fwd", T, B, H, int(reverse), _p(w_hh_c), _p(b_hh_c), _p(tok), _p(tab_c), _p(rowc_c), _p(dense_c),
                     _p(hs), _p(cs), _p(gates), _stream())
        ctx.save_for_backward(tok, w_hh_c, hs, cs, gates)
        ctx.dims = (T, B, H, bool(reverse))
        ctx.V = tab.shape[0] if tab is not None else 0
        ctx.has = (tab is not None, rowc is not None, dense is not None, h0 is not None, c0 is not None)
        return hs

    @staticmethod
    def backward(ctx, ghs):
        tok, w_hh, hs, cs, gates = ctx.saved_tensors
        T, B, H, reverse = ctx.dims
        dev = ghs.device
        ghs = ghs.contiguous()
        BH = B * H
        flat = ghs.view(-1)
        dhs_ext = flat[BH:] if not reverse else flat[:T * BH]
        has_tab, has_rowc, has_dense, has_h0, has_c0 = ctx.has
        # all-T planes form (token-table layers: their bias gradient comes from the input-side reduction): the images of dG the steps
        # hand to each other are kept, feed the conversion-free dW_hh product and the input-side reductions - no f32 dG at all
        ap = _ap_scratch(T, B, H, 1, dev, lstm=True, V=ctx.V) if (has_tab and not has_dense) else None
        dG = torch.empty(T, B, 4 * H, device=dev, dtype=torch.float32) if ap is None else None
        scratch = torch.empty(2, B, H, device=dev, dtype=torch.float32)
        need0 = has_h0 or has_c0
        dh0 = torch.empty(B, H, device=dev, dtype=torch.float32) if need0 else None
        dc0 = torch.empty(B, H, device=dev, dtype=torch.float32) if need0 else None
        with _prof("lstm_bwd_step", T + (1 if need0 else 0), T=T, B=B, H=H, ndir=1):
            wT = torch.empty(H, 4 * H, device=dev, dtype=torch.float32)   # W_hh^T for the direct-to-LDS step kernel
            pair = _pair_scratch(B, H, 1, dev, lstm=True) if ap is None else None
            if ap is not None:
                call("cpg_lstm_seq_bwd_ap", T, B, H, int(reverse), _p(w_hh), _p(cs), _p(gates), _p(dhs_ext), _p(dG), _p(scratch),
                     _p(dh0), _p(dc0), _p(wT), _p(ap), _stream())
            else:
                call("cpg_lstm_seq_bwd", T, B, H, int(reverse), _p(w_hh), _p(cs), _p(gates), _p(dhs_ext), _p(dG), _p(scratch),
                     _p(dh0), _p(dc0), _p(wT), _p(pair), _stream())
        if has_h0:
            dh0 = dh0 + (ghs[T] if reverse else ghs[0])
        nb = query("cpg_gru_wgrad_workspace", T, B, H, max(ctx.V, 1))
        ws = workspace(nb, dev)
        dw_hh = torch.empty(4 * H, H, device=dev, dtype=torch.float32)
        db_hh = torch.empty(4 * H, device=dev, dtype=torch.float32)
        with _prof("lstm_wgrad_hh", 1, T=T, B=B, H=H, ndir=1, ap=int(ap is not None)):
            if ap is not None:
                call("cpg_lstm_wgrad_hh_ap", T, B, H, int(reverse), _p(ap), _p(hs), _p(dw_hh), 0, _p(ws), ws.numel(), _stream())
            else:
                call("cpg_lstm_wgrad_hh", T, B, H, int(reverse), _p(dG), _p(hs), _p(dw_hh), None if has_tab else _p(db_hh), 0, _p(ws),
                     ws.numel(), _p(pair), _stream())
        dtab = torch.empty(ctx.V, 4 * H, device=dev, dtype=torch.float32) if has_tab else None
        drowc = torch.empty(B, 4 * H, device=dev, dtype=torch.float32) if has_rowc else None
        if ap is not None:
            call("cpg_lstm_dgi_reduce_ap", T, B, H, _p(ap), _p(tok), ctx.V, _p(dtab), _p(db_hh), _p(drowc), 0, _p(ws), ws.numel(), _stream())
        elif has_tab or has_rowc:
            call("cpg_lstm_dgi_reduce", T, B, H, _p(dG), _p(tok), ctx.V, _p(dtab), _p(db_hh) if has_tab else None, _p(drowc), 0,
                 _p(ws), ws.numel(), _stream())
        return (None, dtab, drowc, dG if has_dense else None, dh0 if has_h0 else None, dc0 if has_c0 else None, dw_hh, db_hh,
                None, None)


class LstmBiSeqFn(Function):
    """Both directions of one biLSTM layer (torch.nn.LSTM semantics; extension - SURVEY F2) with ONE backward launch per time step
    for the pair (cpg_lstm_biseq_bwd), as GruBiSeqFn.  Returns (slab_fwd, slab_rev) of hidden states, or with finals=True only
    the two final hidden states (forward slot T, reverse slot 0), whose gradients then enter the backward recurrence directly."""

    @staticmethod
    def forward(ctx, tok, tab_f, tab_r, dense_f, dense_r, w_hh_f, b_hh_f, w_hh_r, b_hh_r, T, finals=False):
        dev = w_hh_f.device
        ctx.finals = bool(finals)
        H = w_hh_f.shape[1]
        B = tok.shape[1] if tok is not None else dense_f.shape[1]
        cont = lambda t: t.contiguous() if t is not None else None
        wf, bf, wr, br = cont(w_hh_f), cont(b_hh_f), cont(w_hh_r), cont(b_hh_r)
        tf, tr, df, dr = cont(tab_f), cont(tab_r), cont(dense_f), cont(dense_r)
        # [reverse | forward] per slab: the zero initial slots (T of the reverse half, 0 of the forward half) are adjacent - one fill each
        hs2 = torch.empty(2 * (T + 1), B, H, device=dev, dtype=torch.float32)
        cs2 = torch.empty(2 * (T + 1), B, H, device=dev, dtype=torch.float32)
        hs_r, hs_f, cs_r, cs_f = hs2[:T + 1], hs2[T + 1:], cs2[:T + 1], cs2[T + 1:]
        hs2[T:T + 2].zero_(), cs2[T:T + 2].zero_()
        need_grad = any(t is not None and t.requires_grad for t in (tab_f, tab_r, dense_f, dense_r, w_hh_f, b_hh_f, w_hh_r, b_hh_r))
        g_f = torch.empty(T, 4, B, H, device=dev, dtype=torch.float32) if need_grad else None
        g_r = torch.empty(T, 4, B, H, device=dev, dtype=torch.float32) if need_grad else None
        if lstm_persistent_fits(B, H):
            with _prof("lstm_fwd_persist", 2, T=T, B=B, H=H, ndir=1):   # back to back on ONE stream: each needs every CU
                lstm_seq_fwd_persistent(T, B, H, False, wf, bf, tok, tf, None, df, hs_f, cs_f, g_f)
                lstm_seq_fwd_persistent(T, B, H, True, wr, br, tok, tr, None, dr, hs_r, cs_r, g_r)
        else:
            with _prof("lstm_fwd_step", 2 * T, T=T, B=B, H=H, ndir=1):
                for rev, w, b, tb, dn, hs, cs, gt in ((0, wf, bf, tf, df, hs_f, cs_f, g_f), (1, wr, br, tr, dr, hs_r, cs_r, g_r)):
                    call("cpg_lstm_seq_fwd", T, B, H, rev, _p(w), _p(b), _p(tok), _p(tb), None, _p(dn), _p(hs), _p(cs), _p(gt), _stream())
        ctx.save_for_backward(tok, wf, wr, hs_f, hs_r, cs_f, cs_r, g_f, g_r)
        ctx.leaves = (w_hh_f, w_hh_r)
        ctx.dims = (T, B, H)
        ctx.V = tab_f.shape[0] if tab_f is not None else 0
        ctx.has_tab, ctx.has_dense = tab_f is not None, dense_f is not None
        if finals:
            return hs_f[T], hs_r[0]
        return hs_f, hs_r

    @staticmethod
    def backward(ctx, g_hs_f, g_hs_r):
        tok, wf, wr, hs_f, hs_r, cs_f, cs_r, gt_f, gt_r = ctx.saved_tensors
        T, B, H = ctx.dims
        dev = hs_f.device
        BH = B * H
        ext_f = ext_r = last_f = last_r = None
        z = lambda g, like: g.contiguous() if g is not None else torch.zeros_like(like)
        if ctx.finals:
            last_f, last_r = z(g_hs_f, hs_f[0]), z(g_hs_r, hs_f[0])
        else:
            ext_f = z(g_hs_f, hs_f).view(-1)[BH:]        # slots 1..T
            ext_r = z(g_hs_r, hs_r).view(-1)[:T * BH]    # slots 0..T-1
        ap = _ap_scratch(T, B, H, 2, dev, lstm=True, V=ctx.V) if (ctx.has_tab and not ctx.has_dense) else None   # all-T planes form (see LstmSeqFn.backward)
        dG_f = torch.empty(T, B, 4 * H, device=dev, dtype=torch.float32) if ap is None else None
        dG_r = torch.empty(T, B, 4 * H, device=dev, dtype=torch.float32) if ap is None else None
        sc = torch.empty(2, 2, B, H, device=dev, dtype=torch.float32)
        wT = torch.empty(2, H, 4 * H, device=dev, dtype=torch.float32)
        pair = _pair_scratch(B, H, 2, dev, lstm=True) if ap is None else None
        with _prof("lstm_bwd_step", T, T=T, B=B, H=H, ndir=2):
            if ap is not None:
                call("cpg_lstm_biseq_bwd_ap", T, B, H, _p(wf), _p(wr), _p(cs_f), _p(cs_r), _p(gt_f), _p(gt_r), _p(ext_f), _p(ext_r),
                     _p(last_f), _p(last_r), _p(dG_f), _p(dG_r), _p(sc[0]), _p(sc[1]), _p(wT[0]), _p(wT[1]), _p(ap[0]), _p(ap[1]), _stream())
            else:
                call("cpg_lstm_biseq_bwd", T, B, H, _p(wf), _p(wr), _p(cs_f), _p(cs_r), _p(gt_f), _p(gt_r), _p(ext_f), _p(ext_r),
                     _p(last_f), _p(last_r), _p(dG_f), _p(dG_r), _p(sc[0]), _p(sc[1]), _p(wT[0]), _p(wT[1]),
                     _p(pair[0]) if pair is not None else None, _p(pair[1]) if pair is not None else None, _stream())
        nb = query("cpg_gru_wgrad_workspace", T, B, H, max(ctx.V, 1))
        ws = workspace(nb, dev)
        outs = []
        for rev, dG, hs in ((0, dG_f, hs_f), (1, dG_r, hs_r)):
            gw = _grad_buf(ctx.leaves[rev])
            dw = gw if gw is not None else torch.empty(4 * H, H, device=dev, dtype=torch.float32)
            db = torch.empty(4 * H, device=dev, dtype=torch.float32)
            with _prof("lstm_wgrad_hh", 1, T=T, B=B, H=H, ndir=1, ap=int(ap is not None)):
                if ap is not None:
                    call("cpg_lstm_wgrad_hh_ap", T, B, H, rev, _p(ap[rev]), _p(hs), _p(dw), int(gw is not None), _p(ws), ws.numel(), _stream())
                else:
                    call("cpg_lstm_wgrad_hh", T, B, H, rev, _p(dG), _p(hs), _p(dw), None if ctx.has_tab else _p(db), int(gw is not None),
                         _p(ws), ws.numel(), _p(pair[rev]) if pair is not None else None, _stream())
            dtab = None
            if ctx.has_tab:
                dtab = torch.empty(ctx.V, 4 * H, device=dev, dtype=torch.float32)
                if ap is not None:
                    call("cpg_lstm_dgi_reduce_ap", T, B, H, _p(ap[rev]), _p(tok), ctx.V, _p(dtab), _p(db), None, 0, _p(ws), ws.numel(), _stream())
                else:
                    call("cpg_lstm_dgi_reduce", T, B, H, _p(dG), _p(tok), ctx.V, _p(dtab), _p(db), None, 0, _p(ws), ws.numel(), _stream())
            outs.append((dtab, dG if ctx.has_dense else None, None if gw is not None else dw, db))
        (dtab_f, dd_f, dw_f, db_f), (dtab_r, dd_r, dw_r, db_r) = outs
        return None, dtab_f, dtab_r, dd_f, dd_r, dw_f, db_f, dw_r, db_r, None, None


def lstm_step(tok, tab, rowc, h_prev, c_prev, h_out, c_out, w_hh, b_hh):
    B, H = h_prev.shape
    call("cpg_lstm_step_fwd", B, H, _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), _p(h_prev), _p(c_prev), _p(h_out),
         _p(c_out), _stream())
    return h_out, c_out


# ----------------------------------------------------------------------------------------------- vocab projection
class VocabFcFn(Function):
    """nn.Dropout(p_out) + nn.Linear(h_dim, n_vocab) (models/decoder.py:43-45,83): logits = (hs .* keep/(1-p)) W^T + b."""

    @staticmethod
    def forward(ctx, hs, keep, scale, w, b):
        hs = hs.contiguous()
        R, H = hs.shape
        V = w.shape[0]
        w_c, b_c = w.contiguous(), b.contiguous()
        logits = torch.empty(R, V, device=hs.device, dtype=torch.float32)
        call("cpg_vocab_fc_fwd", _p(hs), _p(keep), float(scale), _p(w_c), _p(b_c), _p(logits), R, H, V, _stream())
        ctx.save_for_backward(hs, keep, w_c)
        ctx.scale = float(scale)
        ctx.leaves = (w, b)
        return logits

    @staticmethod
    def backward(ctx, dl):
        hs, keep, w = ctx.saved_tensors
        dl = dl.contiguous()
        R, H = hs.shape
        V = w.shape[0]
        dev = dl.device
        dhs = torch.empty(R, H, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gw, gb = _grad_buf(ctx.leaves[0]), _grad_buf(ctx.leaves[1])
        direct = gw is not None and gb is not None
        dw = gw if direct else torch.empty(V, H, device=dev, dtype=torch.float32)
        db = gb if direct else torch.empty(V, device=dev, dtype=torch.float32)
        nb = query("cpg_vocab_fc_bwd_workspace", R, H, V)
        ws = workspace(nb, dev)
        call("cpg_vocab_fc_bwd", _p(dl), _p(hs), _p(keep), ctx.scale, _p(w), _p(dhs), _p(dw), _p(db), R, H, V, int(direct), None, None,
             _p(ws), ws.numel(), _stream())
        if direct:
            dw = db = None
        return dhs, None, None, dw, db


class VocabReconFn(Function):
    """The trainer's form of the decoder's tail: nn.Dropout(p_out) + nn.Linear(h_dim, n_vocab) + losses.recon_dec (models/decoder.py:43-45,
    83, losses.py:18-31) as ONE node over time-major rows - the logits are never transposed to [B,T,V] and back, the cross-entropy pass
    leaves the unscaled logit gradient (cpg_recon_ce_tm_fwd), and the backward of the projection takes the upstream gradient and the
    target count as device scalars (cpg_vocab_fc_bwd's g / count): no separate d-logits launch.  hs [T B, H] time-major step outputs,
    ids int64 [B,T]; count_override (optional device scalar: the GLOBAL number of scored targets under data parallelism).
    Returns (loss, logits_tm [T B, V] - not differentiable: the loss is the only consumer)."""

    @staticmethod
    def forward(ctx, hs, keep, scale, w, b, ids, count_override):
        ctx.set_materialize_grads(False)
        hs = hs.contiguous()
        R, H = hs.shape
        V = w.shape[0]
        B, T = ids.shape
        assert R == B * T
        dev = hs.device
        w_c, b_c, ids = w.contiguous(), b.contiguous(), ids.contiguous()
        logits = torch.empty(R, V, device=dev, dtype=torch.float32)
        call("cpg_vocab_fc_fwd", _p(hs), _p(keep), float(scale), _p(w_c), _p(b_c), _p(logits), R, H, V, _stream())
        out = torch.empty(3, device=dev, dtype=torch.float32)
        dl = torch.empty(R, V, device=dev, dtype=torch.float32)
        ws = workspace(512 * 4, dev, tag=1)
        call("cpg_recon_ce_tm_fwd", _p(ids), _p(logits), B, T, V, PAD_IDX, _p(out), _p(dl), _p(ws), _stream())
        if count_override is None:
            count, loss = out[1:2], out[2]
        else:
            count = count_override.reshape(1).to(torch.float32).contiguous()
            loss = out[0] / count.clamp(min=1.0)[0]
        ctx.save_for_backward(hs, keep, w_c, dl, count)
        ctx.scale, ctx.leaves = float(scale), (w, b)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, g, _glogits):
        hs, keep, w, dl, count = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None, None, None
        R, H = hs.shape
        V = w.shape[0]
        dev = hs.device
        g = g.contiguous()
        dhs = torch.empty(R, H, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gw, gb = _grad_buf(ctx.leaves[0]), _grad_buf(ctx.leaves[1])
        direct = gw is not None and gb is not None
        dw = gw if direct else torch.empty(V, H, device=dev, dtype=torch.float32)
        db = gb if direct else torch.empty(V, device=dev, dtype=torch.float32)
        nb = query("cpg_vocab_fc_bwd_workspace", R, H, V)
        ws = workspace(nb, dev)
        call("cpg_vocab_fc_bwd", _p(dl), _p(hs), _p(keep), ctx.scale, _p(w), _p(dhs), _p(dw), _p(db), R, H, V, int(direct), _p(g), _p(count),
             _p(ws), ws.numel(), _stream())
        if direct:
            dw = db = None
        return dhs, None, None, dw, db, None, None


# ----------------------------------------------------------------------------------------------- latent / losses
class ReparamFn(Function):
    """z = mu + exp(logvar/2) * eps (RNN_VAE.sample_z, models/model.py:107-112)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        z = torch.empty_like(mu)
        call("cpg_reparam_fwd", _p(mu), _p(logvar), _p(eps), _p(z), mu.numel(), _stream())
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = dz.contiguous()
        dmu, dlv = torch.empty_like(dz), torch.empty_like(dz)
        call("cpg_reparam_bwd", _p(dz), _p(logvar), _p(eps), _p(dmu), _p(dlv), dz.numel(), _stream())
        return dmu, dlv, None


def latent_sums(mu, logvar):
    """Device tensor [5]: sums behind kl, kl_sharedmu, |logvar|, |mu|, logvar (losses.py:8-15, train_vae.py:33,44-45)."""
    mu, logvar = mu.contiguous(), logvar.contiguous()
    out = torch.empty(5, device=mu.device, dtype=torch.float32)
    ws = workspace(1280 * 4, mu.device, tag=1)
    call("cpg_latent_stats_fwd", _p(mu), _p(logvar), mu.numel(), _p(out), _p(ws), _stream())
    return out


class LatentTermFn(Function):
    """One of the three analytic latent penalties as a differentiable scalar: which = 0 kl_gaussianprior,
    1 kl_gaussian_sharedmu, 2 logvar L1 (mean over the batch of per-row sums)."""

    @staticmethod
    def forward(ctx, mu, logvar, which, b_global):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        sums = latent_sums(mu, logvar)
        ctx.save_for_backward(mu, logvar)
        ctx.which, ctx.bg = which, b_global
        return sums[which] / b_global

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        g = g.contiguous()
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        gs = [None, None, None]
        gs[ctx.which] = _p(g)
        call("cpg_latent_stats_bwd", _p(mu), _p(logvar), mu.numel(), ctx.bg, gs[0], gs[1], gs[2], _p(dmu), _p(dlv), 0,
             _stream())
        return dmu, dlv, None, None


class LatentTermsFn(Function):
    """The three analytic latent penalties of a training step - kl_gaussianprior, kl_gaussian_sharedmu, logvar L1 (losses.py:8-15,
    train_vae.py:33) - from ONE statistics pass, with one backward pass for whichever of them carry a gradient.  Values and
    gradients are those of three LatentTermFn calls (same kernels, same division)."""

    @staticmethod
    def forward(ctx, mu, logvar, b_global):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        scaled = latent_sums(mu, logvar)[:3] / b_global
        ctx.save_for_backward(mu, logvar)
        ctx.bg = b_global
        return scaled[0], scaled[1], scaled[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        mu, logvar = ctx.saved_tensors
        gs = [g.contiguous() if g is not None else None for g in (g0, g1, g2)]
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        call("cpg_latent_stats_bwd", _p(mu), _p(logvar), mu.numel(), ctx.bg, _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(dmu), _p(dlv), 0,
             _stream())
        return dmu, dlv, None


_PADDED_PAIRS = {}   # data_ptr -> [B, 2, Zp] buffer whose halves LatentFn.backward returned as (dmu, dlogvar): taken by EncoderHeadsFn.backward


def _take_padded_pair(dmu, dlv):
    """The zero-padded [B, 2, Zp] buffer behind (dmu, dlv) if they are exactly the two views LatentFn.backward handed out, else None."""
    if dmu is None or dlv is None or dmu.dim() != 2:
        return None
    buf = _PADDED_PAIRS.pop(dmu.data_ptr(), None)
    _PADDED_PAIRS.clear()
    if buf is None:
        return None
    Zp = buf.shape[2]
    ok = (dmu.shape == dlv.shape and dmu.stride() == (2 * Zp, 1) and dlv.stride() == (2 * Zp, 1) and dmu.shape[0] == buf.shape[0]
          and dlv.data_ptr() == dmu.data_ptr() + 4 * Zp)
    return buf if ok else None


class LatentFn(Function):
    """The latent block of a training step as ONE node (cpg_latent_fused_fwd / _bwd): z = mu + exp(logvar / 2) eps
    (RNN_VAE.sample_z, models/model.py:107-112), c ~ Cat(.5,.5) (sample_c_prior :121-126) or a given c, the decoder's initial state /
    constant input zc = [z ; c] (GRUDecoder.init_hidden, models/decoder.py:53-54) and the three analytic penalties kl_gaussianprior,
    kl_gaussian_sharedmu, |logvar|_1 (losses.py:8-15, train_vae.py:33).  eps / c None: drawn inside the kernel from the model's device
    streams (`rng` = DeviceRng; the numbers DeviceRng.normal / onehot2 would have produced at the same point of the stream).
    Returns (z, zc, c, kl, klmu, l1, sums5); backward = one launch for every gradient that reaches mu / logvar through them."""

    @staticmethod
    def forward(ctx, mu, logvar, eps, c, rng):
        ctx.set_materialize_grads(False)     # penalties that do not enter the loss, c, the logged sums: undefined gradients, not zero fills
        mu, logvar = mu.contiguous(), logvar.contiguous()
        B, Z = mu.shape
        dev = mu.device
        C = 2 if c is None else c.shape[1]
        seed = off_e = off_c = 0
        base = None
        if eps is None or c is None:
            assert rng is not None, "LatentFn: draws inside the kernel need the model's DeviceRng"
            base = rng.base_for(dev)
            if eps is None:
                seed, off_e = rng.next(B * Z)
            if c is None:
                seed, off_c = rng.next(B)
        need = mu.requires_grad or logvar.requires_grad
        eps_c = eps.contiguous() if eps is not None else None
        eps_keep = eps_c if eps_c is not None else (torch.empty(B, Z, device=dev, dtype=torch.float32) if need else None)
        z = torch.empty(B, Z, device=dev, dtype=torch.float32)
        zc = torch.empty(B, Z + C, device=dev, dtype=torch.float32)
        c_out = torch.empty(B, C, device=dev, dtype=torch.float32)
        out5 = torch.empty(5, device=dev, dtype=torch.float32)
        ws = workspace(int(query("cpg_latent_fused_workspace")), dev, tag=1)
        call("cpg_latent_fused_fwd", _p(mu), _p(logvar), _p(eps_c), _p(c.contiguous() if c is not None else None), B, Z, C, int(seed),
             int(off_e), int(off_c), _p(base), 0.5, _p(eps_keep) if eps_c is None else None, _p(z), _p(zc), _p(c_out), _p(out5), _p(ws),
             _stream())
        ctx.save_for_backward(mu, logvar, eps_keep)
        ctx.mark_non_differentiable(c_out, out5)
        return z, zc, c_out, out5[0], out5[1], out5[2], out5

    @staticmethod
    def backward(ctx, dz, dzc, _dc, g_kl, g_klmu, g_l1, _d5):
        mu, logvar, eps = ctx.saved_tensors
        B, Z = mu.shape
        if dz is None and dzc is None and g_kl is None and g_klmu is None and g_l1 is None:
            return None, None, None, None, None
        dz = dz.contiguous() if dz is not None else None
        ldzc = 0
        if dzc is not None:
            dzc, ldzc = _ld(dzc)
        gs = [g.contiguous() if g is not None else None for g in (g_kl, g_klmu, g_l1)]
        # dmu / dlogvar as the two halves of ONE row-padded buffer [B, 2, Zp] (Zp = Z rounded up to whole 32-deep slabs, zeros behind
        # column Z): what the encoder heads' backward products contract over, read as it lies (EncoderHeadsFn.backward)
        Zp = -(-Z // 32) * 32
        pair = torch.empty(B, 2, Zp, device=mu.device, dtype=torch.float32)
        dmu, dlv = pair[:, 0, :Z], pair[:, 1, :Z]
        call("cpg_latent_fused_bwd", _p(dz), _p(dzc), int(ldzc), _p(mu), _p(logvar), _p(eps), B, Z, _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(dmu),
             _p(dlv), 2 * Zp, Zp, _stream())
        _PADDED_PAIRS.clear()
        _PADDED_PAIRS[pair.data_ptr()] = pair
        return dmu, dlv, None, None, None


class WeightedSumFn(Function):
    """sum_i w_i * term_i over up to four device scalars, python-float weights (train_vae.py:35-37), one launch forward and one
    backward; products and sums rounded one by one, left to right, as the element-wise expression rounds them."""

    @staticmethod
    def forward(ctx, w, *terms):
        """w: python floats, or a device float32 tensor [4] read by the kernels (captured training steps: the host rewrites it
        between graph replays)."""
        assert 1 <= len(terms) <= 4
        ts = [t.contiguous() for t in terms] + [None] * (4 - len(terms))
        if isinstance(w, torch.Tensor):
            assert w.is_cuda and w.dtype == torch.float32 and w.numel() == 4 and w.is_contiguous()
            ws, wdev = [0.0] * 4, w
        else:
            assert len(w) == len(terms)
            ws, wdev = [float(x) for x in w] + [0.0] * (4 - len(terms)), None
        out = torch.empty(1, device=terms[0].device, dtype=torch.float32)
        call("cpg_weighted_sum4", _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), ws[0], ws[1], ws[2], ws[3], _p(wdev), _p(out), _stream())
        ctx.w, ctx.wdev, ctx.n = ws, wdev, len(terms)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        o = torch.empty(4, device=g.device, dtype=torch.float32)
        call("cpg_scale_fanout4", _p(g), ctx.w[0], ctx.w[1], ctx.w[2], ctx.w[3], _p(ctx.wdev), _p(o), _stream())
        return (None,) + tuple(o[i] for i in range(ctx.n))


class ReconCEFn(Function):
    """losses.recon_dec (losses.py:18-31).  Returns (sum_nll, count) as device scalars; loss = sum/count is formed by
    the caller so a data-parallel run can all-reduce both first."""

    @staticmethod
    def forward(ctx, logits, ids):
        logits, ids = logits.contiguous(), ids.contiguous()
        B, T, V = logits.shape
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        ws = workspace(512 * 4, logits.device, tag=1)
        call("cpg_recon_ce_fwd", _p(ids), _p(logits), B, T, V, PAD_IDX, _p(out), _p(ws), _stream())
        ctx.save_for_backward(logits, ids)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, g):  # pragma: no cover - the differentiable form is ReconLossFn
        raise CpgError("use ReconLossFn")


class ReconLossFn(Function):
    @staticmethod
    def forward(ctx, logits, ids, count_override):
        logits, ids = logits.contiguous(), ids.contiguous()
        B, T, V = logits.shape
        out = torch.empty(3, device=logits.device, dtype=torch.float32)
        ws = workspace(512 * 4, logits.device, tag=1)
        call("cpg_recon_ce_loss_fwd", _p(ids), _p(logits), B, T, V, PAD_IDX, _p(out), _p(ws), _stream())
        if count_override is None:   # the kernel's own sum / max(count, 1)
            ctx.save_for_backward(logits, ids, out[1:2])
            return out[2]
        count = count_override.reshape(1).to(torch.float32)
        ctx.save_for_backward(logits, ids, count.contiguous())
        return out[0] / count.clamp(min=1.0)[0]

    @staticmethod
    def backward(ctx, g):
        logits, ids, count = ctx.saved_tensors
        B, T, V = logits.shape
        g = g.contiguous()
        dl = torch.empty_like(logits)
        call("cpg_recon_ce_bwd", _p(ids), _p(logits), B, T, V, PAD_IDX, _p(g), _p(count), _p(dl), _stream())
        return dl, None, None


def rf_sums(z, rf_w, rf_b, sigma):
    """(raw [B,R] = z @ rf_w, sums [R] = sum_b phi(z_b))  - compute_gaussian_rf, losses.py:84-88."""
    z = z.contiguous()
    B, Z = z.shape
    R = rf_w.shape[1]
    raw = torch.empty(B, R, device=z.device, dtype=torch.float32)
    call("cpg_matmul_nn", _p(z), Z, _p(rf_w), R, _p(raw), R, B, R, Z, 0, _stream())
    sums = torch.empty(R, device=z.device, dtype=torch.float32)
    ws = workspace(128 * R * 4, z.device, tag=1)
    call("cpg_rf_feature_sums", _p(raw), _p(rf_b), B, R, float(sigma), _p(sums), _p(ws), ws.numel(), _stream())
    return raw, sums


class MmdRfFn(Function):
    """losses.mmd_rf (losses.py:59-93) with a fixed random-feature basis.  `reduce` (optional) all-reduces the two
    feature sums across data-parallel ranks; b_global is the global batch."""

    @staticmethod
    def forward(ctx, z, z_prior, rf_w, rf_b, sigma, b_global, reduce, world=1):
        rf_w, rf_b = rf_w.contiguous(), rf_b.contiguous()
        z, z_prior = z.contiguous(), z_prior.contiguous()
        B, Z = z.shape
        R = rf_w.shape[1]
        dev = z.device
        # both feature sums from ONE grouped launch whose epilogue applies the cosine and sums 64-row chunks (cpg_rf_features), then
        # one launch for the chunk sums + the loss (three launches in place of eight)
        raw1 = torch.empty(B, R, device=dev, dtype=torch.float32)
        nb = int(query("cpg_rf_features_workspace", 2, B, R))
        part = workspace(nb, dev, tag=11)
        call("cpg_rf_features", 2, _p(z), _p(z_prior), Z, B, Z, _p(rf_w), R, _p(rf_b), float(sigma), _p(raw1), _p(part), nb, _stream())
        chunks = (B + 63) // 64
        s12 = torch.empty(2, R, device=dev, dtype=torch.float32)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        diff = torch.empty(R, device=dev, dtype=torch.float32)
        if reduce is None:
            call("cpg_rf_sums_loss", _p(part), chunks, R, int(b_global), _p(s12[0]), _p(s12[1]), _p(loss), _p(diff), _stream())
        else:
            call("cpg_rf_sums_loss", _p(part), chunks, R, int(b_global), _p(s12[0]), _p(s12[1]), None, None, _stream())
            reduce(s12[0])
            reduce(s12[1])
            call("cpg_rf_loss", _p(s12[0]), _p(s12[1]), R, int(b_global), _p(loss), _p(diff), _stream())
        ctx.save_for_backward(raw1, rf_w, rf_b, diff)
        # every rank differentiates the GLOBAL loss wrt its local rows; the later gradient all-reduce averages over
        # ranks (SUM * 1/world), so the local contribution is pre-scaled by world: 1/B_global -> 1/B_local.
        ctx.sigma, ctx.bg = float(sigma), max(int(b_global) // max(int(world), 1), 1)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        raw1, rf_w, rf_b, diff = ctx.saved_tensors
        B, R = raw1.shape
        Z = rf_w.shape[0]
        g = g.contiguous()
        dpre = torch.empty_like(raw1)
        call("cpg_rf_bwd", _p(raw1), _p(rf_b), _p(diff), _p(g), B, R, ctx.sigma, ctx.bg, _p(dpre), _stream())
        dz = linear_raw(dpre, rf_w, None)  # dpre [B,R] @ rf_w^T ([Z,R] rows) -> [B,Z]
        return dz, None, None, None, None, None, None, None


class AllGatherRowsFn(Function):
    """Equal-shard row all-gather with the data-parallel backward: every rank evaluates the same global loss, rank r keeps the
    gradient rows of its own shard - pre-scaled by world, because the later gradient all-reduce is a SUM / world."""

    @staticmethod
    def forward(ctx, x, gather_fn, rank, world):
        ctx.rows, ctx.rank, ctx.world = x.shape[0], int(rank), int(world)
        return gather_fn(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        r0 = ctx.rank * ctx.rows
        return g[r0:r0 + ctx.rows] * float(ctx.world), None, None, None


MMD_KERNELS = {"gaussian": 0, "laplace": 1, "energy": 2}  # compute_mmd_kernel, losses.py:102-107


class MmdFullFn(Function):
    """losses.mmd_full_kernel (losses.py:47-56,96-108), F7 quirk included; kernel: index into MMD_KERNELS."""

    @staticmethod
    def forward(ctx, z, z_prior, sigma, kernel=0):
        z, z_prior = z.contiguous(), z_prior.contiguous()
        N, D = z.shape
        dev = z.device
        out = torch.empty(3, device=dev, dtype=torch.float32)
        need = z.requires_grad
        P = torch.empty(N, N, device=dev, dtype=torch.float32) if need else None
        Q = torch.empty(N, N, device=dev, dtype=torch.float32) if need else None
        nb = query("cpg_mmd_full_workspace", N, D)
        ws = workspace(nb, dev)
        call("cpg_mmd_full_fwd", _p(z), _p(z_prior), N, D, float(sigma), int(kernel), _p(out), _p(P), _p(Q), _p(ws),
             ws.numel(), _stream())
        ctx.save_for_backward(z, z_prior, P, Q)
        ctx.sigma, ctx.kernel = float(sigma), int(kernel)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        z, z_prior, P, Q = ctx.saved_tensors
        N, D = z.shape
        g = g.contiguous()
        dz = torch.empty_like(z)
        ws = workspace((N * D + N) * 4 + 256, z.device)
        call("cpg_mmd_full_bwd", _p(z), _p(z_prior), _p(P), _p(Q), _p(g), N, D, ctx.sigma, ctx.kernel, _p(dz), _p(ws),
             ws.numel(), _stream())
        return dz, None, None, None


# ----------------------------------------------------------------------------------------------- CNN classifier
class CnnPoolFn(Function):
    """ReLU + max-over-positions of the token-table form of the classifier's convolutions (models/classifier.py:49-53):
    pooled [B, nconv*F] from ids [B,T], tabs [sum_w * V, F] (emb @ W_l[:, dw, :]^T per layer and tap) and bias [nconv, F].
    Backward: the gradient of each pooled feature goes to the position its maximum came from - to the bias and to the table rows
    that position read (cpg_cnn_classifier_pool_bwd); the tables' own gradients then reach the embedding and the filters."""

    @staticmethod
    def forward(ctx, ids, tabs, bias, V, min_w):
        ids, tabs, bias = ids.contiguous(), tabs.contiguous(), bias.contiguous()
        B, T = ids.shape
        nconv, F_ = bias.shape
        pooled = torch.empty(B, nconv * F_, device=ids.device, dtype=torch.float32)
        need = tabs.requires_grad or bias.requires_grad
        argpos = torch.empty(B, nconv * F_, device=ids.device, dtype=torch.int16) if need else None
        call("cpg_cnn_classifier_pool", _p(ids), B, T, int(V), F_, int(min_w), nconv, _p(tabs), _p(bias), _p(pooled), _p(argpos),
             _stream())
        ctx.save_for_backward(ids, argpos)
        ctx.dims = (B, T, int(V), F_, int(min_w), nconv, tabs.shape[0])
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        ids, argpos = ctx.saved_tensors
        B, T, V, F_, min_w, nconv, rows = ctx.dims
        dpooled = dpooled.contiguous()
        dtabs = torch.empty(rows, F_, device=ids.device, dtype=torch.float32)
        dbias = torch.empty(nconv, F_, device=ids.device, dtype=torch.float32)
        call("cpg_cnn_classifier_pool_bwd", _p(ids), _p(argpos), _p(dpooled), B, T, V, F_, min_w, nconv, _p(dtabs), _p(dbias),
             _stream())
        return None, dtabs, dbias, None, None


# ----------------------------------------------------------------------------------------------- random streams
def rng_normal(shape, seed, offset, device, base=None):
    """base: optional device int64[1] added to `offset` on the device (DeviceRng: hipGraph-replayable training steps)."""
    out = torch.empty(shape, device=device, dtype=torch.float32)
    call("cpg_rng_normal", _p(out), out.numel(), int(seed), int(offset), _p(base), _stream())
    return out


def rng_uniform(shape, seed, offset, device, dtype=torch.float32, base=None):
    out = torch.empty(shape, device=device, dtype=dtype)
    name = "cpg_rng_uniform_f64" if dtype == torch.float64 else "cpg_rng_uniform"
    call(name, _p(out), out.numel(), int(seed), int(offset), _p(base), _stream())
    return out


def rng_bernoulli(shape, p_one, seed, offset, device, base=None):
    out = torch.empty(shape, device=device, dtype=torch.uint8)
    call("cpg_rng_bernoulli_u8", _p(out), out.numel(), float(p_one), int(seed), int(offset), _p(base), _stream())
    return out


def rng_onehot2(n, p_one, seed, offset, device, base=None):
    """One-hot float rows [n,2] of the draws rng_bernoulli((n,), p_one, ...) makes (cpg_rng_onehot2)."""
    out = torch.empty(n, 2, device=device, dtype=torch.float32)
    call("cpg_rng_onehot2", _p(out), int(n), float(p_one), int(seed), int(offset), _p(base), _stream())
    return out


class DeviceRng:
    """Counter-based device streams of one model (Philox4x32-10, cpg_rng_*): every draw takes the next counter range.  The
    counter is split in two: a host-side offset RELATIVE to the current training step and a device-side base that `end_step()`
    advances by the step's total - the launches of a step captured into a hipGraph carry the same relative offsets at every
    replay and still draw fresh numbers, because the base they add lives in device memory."""

    def __init__(self, seed, device=None):
        self.seed, self.offset, self.device = int(seed), 0, device
        self.base = None

    def base_for(self, device):
        if self.base is None:
            self.base = torch.zeros(1, dtype=torch.int64, device=device)
        return self.base

    def next(self, n):
        """(seed, offset) of a fresh n-element range, relative to the device base (pass base=self.base to ops.rng_*)."""
        off = self.offset
        self.offset += (int(n) + 3) // 4 + 1
        return self.seed, off

    def normal(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        seed, off = self.next(n)
        return rng_normal(shape, seed, off, device, self.base_for(device))

    def bernoulli(self, shape, p_one, device):
        n = 1
        for d in shape:
            n *= int(d)
        seed, off = self.next(n)
        return rng_bernoulli(shape, p_one, seed, off, device, self.base_for(device))

    def onehot2(self, n, p_one, device):
        """c ~ Cat([1 - p_one, p_one]) as one-hot float rows [n,2]: the draws of bernoulli((n,), p_one), one launch."""
        seed, off = self.next(int(n))
        return rng_onehot2(n, p_one, seed, off, device, self.base_for(device))

    def end_step(self):
        """Close the current step: the device base moves past everything drawn since the last call, offsets restart at 0."""
        if self.base is not None and self.offset:
            call("cpg_counter_add_u64", _p(self.base), int(self.offset), _stream())
        self.offset = 0
