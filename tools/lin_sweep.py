"""Tile sweep of the nn.Linear-shaped products of the headline step (heads, [z;c] row constant, random features, vocabulary
projection): us per launch of cpg_linear_fwd / _bwd_input / _bwd_weight under every value of options gemm_tile / tn_tile / tn_split,
cache-cold-ish (a 512 MB buffer is rewritten between timed batches).  Standalone products: a micro-benchmark is representative here
(unlike the recurrent kernels, which must be measured inside the step: tools/insitu.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
from cpg import ops  # noqa: E402
from cpg.ops import _p, _stream, call, query, workspace  # noqa: E402

dev = torch.device("cuda")
SHAPES = {  # name: (M, N, K, ldw)   y[M,N] = x[M,K] w[N,K]^T
    "heads": (2048, 510, 1024, 1024),
    "rowc": (2048, 1536, 512, 662),
    "rf": (2048, 500, 510, 510),
}
flush = torch.empty(128 << 20, device=dev)


def timeit(fn, iters=40):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        flush.add_(1.0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


def main():
    g = torch.Generator().manual_seed(0)
    for name, (M, N, K, ldw) in SHAPES.items():
        x = torch.randn(M, K, generator=g).to(dev)
        wfull = torch.randn(N, ldw, generator=g).to(dev)
        w = wfull[:, ldw - K:] if ldw != K else wfull
        b = torch.randn(N, generator=g).to(dev)
        y = torch.empty(M, N, device=dev)
        dy = torch.randn(M, N, generator=g).to(dev)
        dx = torch.empty(M, K, device=dev)
        dw = torch.zeros(N, ldw, device=dev)
        db = torch.empty(N, device=dev)
        tiles = [None, "128x64", "64x64", "64x32", "32x64", "32x32", "128x32", "32x128"]
        for kind in ("fwd", "bwd_input"):
            res = []
            for t in tiles:
                with ops.options(**({"gemm_tile": t} if t else {})):
                    if kind == "fwd":
                        fn = lambda: call("cpg_linear_fwd", _p(x), K, _p(w), ldw, _p(b), _p(y), N, M, N, K, 0, _stream())
                    else:
                        fn = lambda: call("cpg_linear_bwd_input", _p(dy), N, _p(w), ldw, _p(dx), K, M, N, K, 0, _stream())
                    res.append((timeit(fn), t or "policy"))
            print(f"{name:6s} {kind:10s} M={M} N={N} K={K}: " + "  ".join(f"{t}={u:.1f}" for u, t in res), flush=True)
        res = []
        for t in (None, "128x64", "64x64", "128x32", "32x128"):
            for sp in (None, 1, 2, 4, 8, 16):
                o = {}
                if t:
                    o["tn_tile"] = t
                if sp:
                    o["tn_split"] = sp
                with ops.options(**o):
                    nb = query("cpg_linear_bwd_weight_workspace", M, N, K)
                    ws = workspace(nb, dev)
                    fn = lambda: call("cpg_linear_bwd_weight", _p(dy), N, _p(x), K, _p(dw[:, ldw - K:]), ldw, _p(db), M, N, K, 1, _p(ws),
                                      ws.numel(), _stream())
                    try:
                        res.append((timeit(fn), f"{t or 'policy'}/s{sp or 'p'}"))
                    except ops.CpgError as e:
                        res.append((float('nan'), f"{t}/s{sp}:err"))
        res.sort()
        print(f"{name:6s} bwd_weight M={M} N={N} K={K}: " + "  ".join(f"{t}={u:.1f}" for u, t in res[:8]) +
              "  | policy=" + "".join(f"{u:.1f}" for u, t in res if t == "policy/sp"), flush=True)


if __name__ == "__main__":
    main()
