#!/usr/bin/env python3
"""Fused step kernel vs split (product + cell kernel) with G row groups on G streams: time per full-batch time step."""
import ctypes, os, sys
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
hs = torch.zeros(T + 1, B, H, device=dev); hs[0] = torch.randn(B, H, generator=g).to(dev)
hs2 = hs.clone(); gates = torch.empty(T, 4, B, H, device=dev); gh = torch.empty(B, 3 * H, device=dev)
main = torch.cuda.current_stream()

def chain(split, hsb, r0, r1, stream):
    st = ctypes.c_void_p(stream.cuda_stream)
    if split:
        call("cpg_gru_seq_fwd_split", T, B, H, 0, _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), None, _p(hsb), _p(gates), _p(gh), r0, r1, st)
    else:
        call("cpg_gru_seq_fwd", T, B, H, 0, _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), None, _p(hsb), _p(gates), r0, r1, None, st)

chain(False, hs, 0, B, main); chain(True, hs2, 0, B, main); torch.cuda.synchronize()
print("max |fused - split| over the sequence:", float((hs - hs2).abs().max()))
for split in (False, True):
    for G in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(G)]
        step = B // G
        def run():
            ev = main.record_event()
            for i, s in enumerate(streams):
                s.wait_event(ev); chain(split, hs2 if split else hs, i * step, (i + 1) * step, s)
            for s in streams: main.wait_stream(s)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print(f"{'split' if split else 'fused'} G={G}: {e0.elapsed_time(e1) / 10 * 1e3 / T:.1f} us per time step (whole batch)")
