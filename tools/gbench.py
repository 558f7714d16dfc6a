#!/usr/bin/env python3
"""Micro-benchmark of the nn.Linear-shaped products (cpg_linear_fwd = NT, cpg_matmul_nn = NN) at the shapes of the training
step, sweeping the CPG_GEMM_TILE knob (read by the launcher at every call)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch  # noqa: E402
from cpg.ops import _p, _stream, call  # noqa: E402

SHAPES = [("nt", 2048, 1536, 512, "rowc: [z;c] W_ih^T"), ("nt", 2048, 510, 1024, "q_mu / q_logvar"),
          ("nt", 2048, 2048, 510, "MMD Gram"), ("nt", 2048, 500, 510, "rf features (as NT)"),
          ("nn", 2048, 512, 1536, "d rowc -> d[z;c]"), ("nn", 2048, 1024, 510, "d mu -> d h_enc"),
          ("nn", 2048, 500, 510, "z @ rf_w"), ("nn", 51200, 512, 24, "d logits -> d out")]
TILES = ["auto", "128x64", "64x64", "64x32", "32x64", "32x32"]


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda")
    for kind, M, N, K, what in SHAPES:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) if kind == "nt" else torch.randn(K, N, device=dev)
        y = torch.empty(M, N, device=dev)
        if kind == "nt":
            fn = lambda: call("cpg_linear_fwd", _p(x), K, _p(w), K, None, _p(y), N, M, N, K, 0, _stream())
        else:
            fn = lambda: call("cpg_matmul_nn", _p(x), K, _p(w), N, _p(y), N, M, N, K, 0, _stream())
        out = []
        for t in TILES:
            if t == "auto":
                os.environ.pop("CPG_GEMM_TILE", None)
            else:
                os.environ["CPG_GEMM_TILE"] = t
            us = timeit(fn)
            out.append(f"{t} {us:6.1f}")
        os.environ.pop("CPG_GEMM_TILE", None)
        print(f"{kind} M={M:5d} N={N:4d} K={K:4d} ({what}; {2.0 * M * N * K / 1e9:.1f} GFLOP): " + " | ".join(out))


if __name__ == "__main__":
    main()
