#!/usr/bin/env python3
"""Do independent row-group chains of the GRU forward sequence overlap when issued on separate streams?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch
from cpg.ops import _p, call

dev = torch.device("cuda")
B, H, T, V = 2048, 512, 25, 24
g = torch.Generator().manual_seed(0)
w_hh = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(dev); b_hh = torch.zeros(3 * H, device=dev)
tab = torch.randn(V, 3 * H, generator=g).to(dev) * 0.3; rowc = torch.randn(B, 3 * H, generator=g).to(dev) * 0.3
tok = torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev)
hs = torch.zeros(T + 1, B, H, device=dev); gates = torch.empty(T, 4, B, H, device=dev)
import ctypes
def chain(r0, r1, stream):
    call("cpg_gru_seq_fwd", T, B, H, 0, _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), None, _p(hs), _p(gates), r0, r1, None,
         ctypes.c_void_p(stream.cuda_stream))
main = torch.cuda.current_stream()
for G in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(G)]
    step = B // G
    def run():
        ev = main.record_event()
        for i, s in enumerate(streams):
            s.wait_event(ev); chain(i * step, (i + 1) * step, s)
        for s in streams: main.wait_stream(s)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f"G={G}: {e0.elapsed_time(e1) / 10 * 1e3 / T:.1f} us per time step (whole batch)")
