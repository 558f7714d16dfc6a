#!/usr/bin/env python3
"""Benchmark of the MI355X peptide-WAE training step (BASELINE.json metric: peptide-seq/s per WAE step).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one full train_vae iteration on one synthetic batch already resident in HBM: encoder + reparameterisation +
teacher-forced decoder, recon CE + KL + full-kernel MMD + random-feature MMD, backward, flat-gradient all-reduce
(N>1), global-norm clip + Adam; all random draws (eps, c, both dropout masks, z_prior) are generated on the device
inside the step.  Prints ONE JSON line (rank 0) with the `roofline` and `cpu_baseline` objects described in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "controlled-peptide-generation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X f32-input MFMA = f32 vector peak (MI355X_MICROARCH.md, chip-level parameters)
# HBM bytes per launch of the dominant kernel from PMC counters (separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
# passes over tools/kbench.py at the bench shapes; (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 read-side correction
# of MI355X_MICROARCH.md section HBM).  Counters cannot be collected from inside this script; the numbers and commands are
# in profiles/r01_bench_n1_summary_final.md.  Only valid for the default GRU / B=2048 / H=512 workload.
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide); the forward step executes 6 bf16 MFMA flops per f32 flop
PMC_TRAFFIC_BYTES = {("gru", 2048, 512): (2 * 21.1 + 21.4) * 1024 * 1024}


def model_kwargs(z_dim, enc_h, enc_layers=1, emb_dim=150, cell='gru'):
    return dict(
        z_dim=z_dim, c_dim=2, emb_dim=emb_dim, pretrained_emb=None, freeze_embeddings=False, flow=0, flow_type='',
        E_args=dict(h_dim=enc_h, biGRU=True, layers=enc_layers, p_dropout=0.0, cell=cell),
        G_args=dict(G_class='gru', GRU_args=dict(p_word_dropout=0.3, p_out_dropout=0.3, skip_connetions=False, cell=cell),
                    deconv_args=dict()),
        C_args=dict(min_filter_width=3, max_filter_width=5, num_filters=100, dropout=0.5))


def train_flops_per_seq(T, E, He, Z, V, R, B, gates=3):
    """Executed FLOPs (2*MAC) of one training step per sequence on this implementation (token-table form), for the
    whole-step rate quoted in `extra`; the algorithmic count with the dense W_ih product is SURVEY 8d's."""
    Hd = Z + 2
    rec = 2 * T * 2 * gates * He * He + T * 2 * gates * Hd * Hd    # recurrent products, fwd
    small = 2 * gates * Hd * Hd + 2 * 2 * 2 * He * Z + T * 2 * Hd * V + 2 * 2 * Z * R
    fwd = rec + small
    return fwd + 2 * rec + 2 * small + 3 * 2 * B * Z                  # bwd: dh product + dW product (+ Gram MMD)


def cpu_baseline(sd, T, V, B_sample, steps):
    """The numpy oracle (a port of the reference's algorithm, oracle/wae.py + oracle/optim.py) timed on the host cores
    on a bounded sample of the same workload: same model dimensions, a smaller batch."""
    from oracle import wae, optim
    P = {k: v.detach().cpu().numpy().copy() for k, v in sd.items() if not k.startswith("classifier")}
    Z = P["encoder.q_mu.weight"].shape[0]
    Hd = Z + 2
    rs = np.random.RandomState(0)
    ids = torch.randint(4, V, (B_sample, T)).numpy()
    ids[:, 0] = 2
    ids[:, -2] = 3
    ids[:, -1] = 1
    opt = optim.AdamClip(P, lr=1e-3, max_norm=5.0)

    def draw():
        c = np.zeros((B_sample, 2), np.float32)
        c[np.arange(B_sample), rs.randint(0, 2, B_sample)] = 1
        return dict(eps=rs.randn(B_sample, Z).astype(np.float32), c=c,
                    wd_mask=(rs.rand(B_sample, T) < 0.3).astype(np.uint8),
                    out_mask=(rs.rand(B_sample, T, Hd) >= 0.3).astype(np.uint8),
                    z_prior_full=rs.randn(B_sample, Z).astype(np.float32),
                    z_prior_rf=rs.randn(B_sample, Z).astype(np.float32),
                    rf_w=rf_w, rf_b=rf_b)
    rf_w = rs.randn(Z, 500).astype(np.float32)
    rf_b = (2 * np.pi * rs.rand(500)).astype(np.float32)
    times = []
    for i in range(steps + 1):
        rnd = draw()
        t0 = time.perf_counter()
        terms, G, _ = wae.train_loss_and_grads(P, ids, rnd, 1.0, 0.0, 1e-3, "mmdrf")
        opt.step(P, G)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times[1:]))
    return B_sample / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2048, help="sequences per GPU per step")
    ap.add_argument("--hidden", type=int, default=512, help="encoder h_dim and decoder hidden (z_dim = hidden-2)")
    ap.add_argument("--seq-len", type=int, default=25)
    ap.add_argument("--enc-layers", type=int, default=1, help="encoder biGRU layers (BASELINE.json configs[4] uses 2; the decoder stays 1 layer as in the reference)")
    ap.add_argument("--cell", default="gru", choices=["gru", "lstm"], help="gru = the reference's cell (parity pinned); lstm = extension")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-class", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=256)
    args = ap.parse_args()

    from cpg import dist as cdist
    world, rank, local = cdist.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs; there is no CPU path"
    local = cdist.local_device(local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import cfg
    import losses
    from cpg import ops
    from cpg.synth import synth_ids
    from models.model import RNN_VAE
    import train_vae as tv

    T, V, B, Hh = args.seq_len, 24, args.batch, args.hidden
    Z, E, R = Hh - 2, 150, 500
    torch.manual_seed(1238)
    model = RNN_VAE(n_vocab=V, max_seq_len=T, **model_kwargs(Z, Hh, enc_layers=args.enc_layers, cell=args.cell)).to(dev)
    model.device = dev
    losses.rf.clear()
    losses._rf_basis(torch.zeros(1, Z, device=dev), R, False)          # same basis on every rank (same seed)
    model.use_device_rng(1238 + 7919 * rank)                            # rank-distinct draws
    losses.set_prior_sampler(lambda z: model._randn(z.shape[0], z.shape[1]))
    cdist.broadcast_params(model.parameters())
    reduce_fn = cdist.allreduce_sum if world > 1 else None
    if world > 1:
        losses.set_distributed(cdist.allreduce_sum, world)
    cfgv = cfg.Bunch(lr=1e-3, clip_grad=5.0, z_regu_loss='mmdrf', lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
                     beta=cfg.Bunch(start=cfg.Bunch(val=1.0, iter=0), end=cfg.Bunch(val=2.0, iter=40000)))
    trainer = tv.make_optimizer(cfgv, model, reduce_fn, world)
    g = torch.Generator().manual_seed(1238 + rank)
    pool = [synth_ids(B, T, V, g).to(dev) for _ in range(8)]

    def step(it):
        return tv.train_step(cfgv, model, trainer, pool[it % len(pool)], it)

    for it in range(args.warmup):
        step(it)
    torch.cuda.synchronize()
    cdist.barrier()
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    for it in range(args.steps):
        out = step(args.warmup + it)
    torch.cuda.synchronize()
    cdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(out["L_vae"].item())
    assert np.isfinite(loss_val), "non-finite loss in the timed region"
    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    seq_per_s = B * world * args.steps / dt

    # dominant kernel: the fused GRU forward step (one launch per time step; decoder sequence = H 512, B rows)
    gates = 3 if args.cell == "gru" else 4
    recs = [r for r in prof if r[0] == args.cell + "_step_fwd" and r[5] == Hh]
    tot_ms = sum(r[1].elapsed_time(r[2]) for r in recs)
    launches = sum(r[3] for r in recs)
    avg_us = tot_ms * 1e3 / max(launches, 1)
    flops_launch = 2.0 * B * Hh * gates * Hh
    achieved = flops_launch / (avg_us * 1e-6) / 1e12 if launches else 0.0
    kname = ("gru_step_fwd_kernel<TileCfg<64,96,32,2,2,3>,true>" if args.cell == "gru"
             else "lstm_step_fwd_kernel<TileCfg<64,128,32,2,2,4>,true>")
    # achieved / peak are quoted on the ALGORITHMIC f32 product (2*B*H*gates*H per launch) against the f32 MFMA peak: the
    # result is an f32-grade product.  The GRU kernel executes it as six bf16 MFMAs on 3-way split operands,
    # i.e. 6x the algorithmic flops on the bf16 pipe: that fraction is reported beside it.
    roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2),
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                "traffic": PMC_TRAFFIC_BYTES.get((args.cell, B, Hh)), "avg_launch_us": round(avg_us, 2), "launches_timed": launches,
                "flops_per_launch": flops_launch}
    if args.cell == "gru":
        roofline["product_form"] = "f32 in/out, 6 x v_mfma_f32_16x16x32_bf16 on 3-way split operands (f32-grade)"
        roofline["executed_bf16_tflops"] = round(6 * achieved, 1)
        roofline["executed_frac_of_bf16_peak"] = round(6 * achieved / PEAK_BF16_MFMA_TFLOPS, 4)
    step_tflops = train_flops_per_seq(T, E, Hh, Z, V, R, B, gates) * B / (ms * 1e-3) / 1e12
    extra = {"loss_last_step": round(loss_val, 4), "executed_step_tflops_per_gpu": round(step_tflops, 2),
             "executed_step_frac_of_f32_peak": round(step_tflops / PEAK_F32_MFMA_TFLOPS, 4)}

    cfg_tag = ("BASELINE.json configs[1]" if (Hh, T, args.enc_layers, B) == (512, 25, 1, 2048)
               else "BASELINE.json configs[4] dimensions, GRU cells, 1-layer decoder as in the reference"
               if (Hh, T, args.enc_layers) == (1024, 50, 2) else "non-default dimensions")
    line = {
        "metric": "peptide-seq/s per WAE training step", "value": round(seq_per_s, 1), "unit": "seq/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"WAE train step ({cfg_tag}): biGRU encoder h={Hh} {args.enc_layers} layer, z={Z}, GRU decoder "
                               f"h={Hh}, emb 150, vocab 24, batch {B}/GPU, seq_len {T}; "
                               + ("GRU cell = the reference's cell, parity pinned (the reference has no LSTM; --cell lstm runs "
                                  "the LSTM extension)" if args.cell == "gru" else
                                  "LSTM cell (extension, torch.nn.LSTM semantics; parity unpinned against the GRU-only reference)")
                               + ", f32 storage; recurrent products on the MFMA units in f32-grade forms (exact-f32 MFMA, or six bf16 MFMAs on 3-way "
                                 "split operands)",
                   "global_batch": B * world, "seq_len": T, "parallelism": f"dp{world}"},
        "roofline": roofline, "extra": extra,
    }
    if world == 1 and not args.no_cpu_baseline:
        v, sdt = cpu_baseline(model.state_dict(), T, V, args.cpu_sample_batch, 9)   # ~12 s of host work
        line["cpu_baseline"] = {"value": round(v, 1), "unit": "seq/s", "cores": os.cpu_count(), "kind": "port",
                                "sample": f"numpy oracle (oracle/wae.py + oracle/optim.py), same model dims, batch "
                                          f"{args.cpu_sample_batch}, median of 9 steps after 1 warm-up ({sdt:.2f} s/step), all four "
                                          f"regularisers incl. the [N,N,D] full-kernel MMD"}
    if world == 1 and not args.no_class:
        line["extra"]["class"] = class_bench(dev)
    print(json.dumps(line))


def class_bench(dev, N=262144):
    """CLaSS inner loop at the reference's default dims (config A): z-space LR scoring + accept, greedy decode of all z."""
    from cpg import class_sampler, ops
    from models.model import RNN_VAE
    torch.manual_seed(1238)
    m = RNN_VAE(n_vocab=24, max_seq_len=25, **model_kwargs(100, 80)).to(dev)
    m.device = dev
    z = ops.rng_normal((N, 100), 99, 0, dev)
    c = torch.zeros(N, 2, device=dev)
    c[:, 1] = 1
    coef = torch.randn(2, 100, device=dev, dtype=torch.float64) / 10
    icpt = torch.zeros(2, device=dev, dtype=torch.float64)
    tgt = torch.tensor([1, 0], device=dev, dtype=torch.int32)
    u = ops.rng_uniform((N,), 5, 0, dev, dtype=torch.float64)
    for _ in range(2):
        ids, _, _ = m.generate_sentences(N, z, c, sample_mode='greedy')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    probs, accum, acc = class_sampler.lr_score_accept(z, coef, icpt, tgt, u)
    ids, _, _ = m.generate_sentences(N, z, c, sample_mode='greedy')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = 25  # the device loop always runs max_seq_len steps
    # beam-5 / n_best-3 decode (the reference's decode_from_z mode), incl. the hypothesis walk-back and its D2H copy
    from cpg import decode as cdecode
    Nb = 131072
    cdecode.decode_beam_raw(m.decoder, z[:1024], c[:1024], 25)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hyps, lens, _ = cdecode.decode_beam_arrays(m.decoder, z[:Nb], c[:Nb], 25, beam_size=5, n_best=3)
    dtb = time.perf_counter() - t0
    return {"workload": f"config A (z=100, dec h=102), {N} z: LR score+accept, greedy decode of all z; beam-5 on {Nb} z",
            "z_per_s": round(N / dt, 1), "decoder_evals_per_s": round(N * steps / dt, 1),
            "accepted_per_s": round(float(acc.sum().item()) / dt, 1),
            "beam5_z_per_s": round(Nb / dtb, 1), "beam5_decoder_evals_per_s": round(Nb * 5 * (hyps.shape[2] - 1) / dtb, 1)}


if __name__ == "__main__":
    main()
