#!/usr/bin/env python3
"""Micro-benchmark of the GRU step kernels and the weight-gradient product at a given (B, H, T) - for A/B tuning runs
on the GPU box (within-process interleaved variants selected through CPG_* environment knobs read by the library)."""
import argparse
import os
import sys
import time

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
    ap.add_argument("--no-gates", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--knob", action="append", default=[], help="NAME=v1,v2,... environment knob values to sweep")
    a = ap.parse_args()
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
    gates = torch.empty(T, 4, B, H, device=dev)
    dhs = torch.randn(T, B, H, generator=g).to(dev) * 0.1
    dG = torch.empty(T, B, 4 * H, device=dev)
    scr = torch.empty(2, B, H, device=dev)
    dh0 = torch.empty(B, H, device=dev)
    dw = torch.empty(3 * H, H, device=dev)
    db = torch.empty(3 * H, device=dev)
    nb = query("cpg_gru_wgrad_workspace", T, B, H, V)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def fwdp():
        ops.gru_seq_fwd_persistent(T, B, H, False, w_hh, b_hh, tok, tab, rowc, None, hs, None if a.no_gates else gates)

    def fwd():
        call("cpg_gru_seq_fwd", T, B, H, 0, _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), None, _p(hs), None if a.no_gates else _p(gates), 0, B, None, _stream())

    wT = torch.empty(H, 3 * H, device=dev)

    def bwd():
        call("cpg_gru_seq_bwd", T, B, H, 0, _p(w_hh), _p(hs), _p(gates), _p(dhs), None, _p(dG), _p(scr), _p(dh0), 0, B, None,
             _p(wT), _stream())

    def bwdp():
        ops.gru_seq_bwd_persistent(T, B, H, False, w_hh, hs, gates, dhs, None, dG, dh0)

    def bwdc():
        ops.gru_seq_bwd_chain(T, B, H, False, w_hh, hs, gates, dhs, None, dG, dh0, wT)

    dG2 = torch.empty(T, B, 4 * H, device=dev)
    wT2 = torch.empty(2, H, 3 * H, device=dev)
    sc2 = torch.empty(2, 2, B, H, device=dev)

    def bwdc2():
        ops.gru_biseq_bwd_chain(T, B, H, w_hh, w_hh, hs, hs, gates, gates, dhs, dhs, dG, dG2, wT2)

    def bwd2():
        call("cpg_gru_biseq_bwd", T, B, H, _p(w_hh), _p(w_hh), _p(hs), _p(hs), _p(gates), _p(gates), _p(dhs), _p(dhs), None, None, _p(dG), _p(dG2),
             _p(sc2[0]), _p(sc2[1]), _p(wT2[0]), _p(wT2[1]), _stream())

    big_ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)  # large enough for any split the knobs select

    def wgrad():
        call("cpg_gru_wgrad_hh", T, B, H, 0, _p(dG), _p(hs), _p(dw), _p(db), 0, _p(big_ws), big_ws.numel(), _stream())

    sweeps = [("base", {})]
    for k in a.knob:
        name, vals = k.split("=")
        for v in vals.split(","):
            sweeps.append((f"{name}={v}", {name: v}))
    fl_step = 2.0 * B * H * 3 * H
    for rnd in range(2):
        for label, env in sweeps:
            for k, v in env.items():
                os.environ[k] = v
            f = timeit(fwd, a.iters) / T
            fp = timeit(fwdp, a.iters) / T if ops.persistent_fits(B, H) else float("nan")
            bp = timeit(bwdp, a.iters) / (T + 1) if ops.persistent_fits(B, H) else float("nan")
            b = timeit(bwd, a.iters) / (T + 1)
            chain_ok = ops.chain_bwd_covers(T, B, H)
            bc = timeit(bwdc, a.iters) / (T + 1) if chain_ok else float("nan")
            bc2 = timeit(bwdc2, a.iters) / T if chain_ok else float("nan")
            b2 = timeit(bwd2, a.iters) / T
            w = timeit(wgrad, a.iters)
            for k in env:
                os.environ.pop(k, None)
            print(f"[{rnd}] {label:24s} fwd-persistent {fp:7.1f} us/step ({fl_step / fp / 1e6:6.1f} TF)  bwd-persistent {bp:7.1f} us/step ({fl_step / bp / 1e6:6.1f} TF)")
            print(f"[{rnd}] {label:24s} bwd-chain {bc:7.1f} us/step  pair: chain {bc2:7.1f} us/step, per-step launches {b2:7.1f} us/step")
            print(f"[{rnd}] {label:24s} fwd {f:7.1f} us/step ({fl_step / f / 1e6:6.1f} TF)  bwd {b:7.1f} us/step "
                  f"({fl_step / b / 1e6:6.1f} TF)  wgrad {w:8.1f} us ({fl_step * T / w / 1e6:6.1f} TF)")


if __name__ == "__main__":
    main()
