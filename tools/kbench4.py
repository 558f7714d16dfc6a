#!/usr/bin/env python3
"""Plain NT product timing at the recurrent-step shape and at longer K (is the K=512 product itself the limit?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch
from cpg.ops import _p, _stream, call
dev = torch.device("cuda")
def t(M, N, K, iters=20):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
    f = lambda: call("cpg_linear_fwd", _p(x), K, _p(w), K, None, _p(y), N, M, N, K, 0, _stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF")
for shape in [(2048, 1536, 512), (2048, 1536, 2048), (2048, 1536, 8192), (4096, 4096, 4096), (8192, 8192, 1024), (8192, 1536, 512)]:
    t(*shape)
