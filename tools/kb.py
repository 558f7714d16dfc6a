#!/usr/bin/env python3
"""Micro-benchmark of the recurrent kernels at a given (B, H, T): persistent / per-step forward, backward step (single
direction and pair), dW_hh product.   CPG_LIB_PATH=build_variants/libcpg_<tag>.so python tools/kb.py [--only fwdp,wgrad]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch  # noqa: E402
from cpg import ops  # noqa: E402
from cpg.ops import _p, _stream, call, query  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=2048)
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--T", type=int, default=25)
    ap.add_argument("--V", type=int, default=24)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--only", default="fwdp,fwd,bwd,bwd2,wgrad")
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("CPG_LIB_PATH", "default")))
    ap.add_argument("--opt", action="append", default=[], help="name=value launch-policy option (cpg_set_option), repeatable")
    a = ap.parse_args()
    for kv in a.opt:
        k, v = kv.split("=")
        ops.set_option(k, v)
    if a.opt:
        a.tag += " " + ",".join(a.opt)
    dev = torch.device("cuda")
    ops.set_compute_mode(a.dtype)
    B, H, T, V = a.B, a.H, a.T, a.V
    g = torch.Generator(device="cpu").manual_seed(0)
    w_hh = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(dev)
    b_hh = torch.randn(3 * H, generator=g).to(dev) * 0.1
    tab = torch.randn(V, 3 * H, generator=g).to(dev) * 0.3
    rowc = torch.randn(B, 3 * H, generator=g).to(dev) * 0.3
    tok = torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev)
    hs = torch.zeros(T + 1, B, H, device=dev)
    hs[0] = torch.randn(B, H, generator=g).to(dev)
    gates = torch.empty(T, 4, B, H, device=dev, dtype=ops.gates_dtype(B, H))
    dhs = torch.randn(T, B, H, generator=g).to(dev) * 0.1
    dG, dG2 = torch.empty(T, B, 4 * H, device=dev), torch.empty(T, B, 4 * H, device=dev)
    scr, sc2 = torch.empty(2, B, H, device=dev), torch.empty(2, 2, B, H, device=dev)
    dh0 = torch.empty(B, H, device=dev)
    dw, db = torch.empty(3 * H, H, device=dev), torch.empty(3 * H, device=dev)
    wT, wT2 = torch.empty(H, 3 * H, device=dev), torch.empty(2, H, 3 * H, device=dev)
    ps1, ps2 = ops._pair_scratch(B, H, 1, dev), ops._pair_scratch(B, H, 2, dev)   # f16-pair backward step (None: not covered)
    big_ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    fns = {
        "fwdp": (lambda: ops.gru_seq_fwd_persistent(T, B, H, False, w_hh, b_hh, tok, tab, rowc, None, hs, gates), T),
        "fwd": (lambda: call("cpg_gru_seq_fwd", T, B, H, 0, _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), None, _p(hs), _p(gates), 0, B,
                             None, _p(ops.weight_exp(w_hh)), _stream()), T),
        "bwd": (lambda: call("cpg_gru_seq_bwd", T, B, H, 0, _p(w_hh), _p(hs), _p(gates), _p(dhs), None, _p(dG), _p(scr), _p(dh0), 0, B, None,
                             _p(wT), _p(ps1), 0, _stream()), T + 1),
        "bwd2": (lambda: call("cpg_gru_biseq_bwd", T, B, H, _p(w_hh), _p(w_hh), _p(hs), _p(hs), _p(gates), _p(gates), _p(dhs), _p(dhs), None,
                              None, _p(dG), _p(dG2), _p(sc2[0]), _p(sc2[1]), _p(wT2[0]), _p(wT2[1]), _p(ps2[0]) if ps2 is not None else None,
                              _p(ps2[1]) if ps2 is not None else None, 0, _stream()), T),
        "wgrad": (lambda: call("cpg_gru_wgrad_hh", T, B, H, 0, _p(dG), _p(hs), _p(dw), _p(db), 0, _p(big_ws), big_ws.numel(), None, 0, _stream()), 1),
    }
    fns["fwd"][0]()   # valid state slab / gates for the backward kernels
    fl = 2.0 * B * H * 3 * H
    for rnd in range(a.rounds):
        out = []
        for k in a.only.split(","):
            if k == "fwdp" and not ops.persistent_fits(B, H):
                continue
            fn, div = fns[k]
            us = timeit(fn, a.iters) / div
            tf = fl * (T if k == "wgrad" else (2 if k == "bwd2" else 1)) / us / 1e6
            out.append(f"{k} {us:8.2f} us ({tf:6.1f} TF)")
        print(f"[{rnd}] {a.tag:28s} " + "  ".join(out), flush=True)
    ops.check_persistent()


if __name__ == "__main__":
    main()
