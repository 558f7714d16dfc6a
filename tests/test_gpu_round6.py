"""Round-6 parity additions.

  * the f16-pair engines' weight range: rounds 4-5 multiplied weights by a fixed 2^8 before the f16 split, so
    a weight of magnitude >= 256 became an infinity.  Every engine now takes the power of two from the weight matrix' own largest
    magnitude (csrc/gemm_core.h: weight_exp_from_parts) - tested here with weights of 300 / -400 / 5000 planted into every recurrent
    matrix, at the benchmarked size (B = 2048, h = 512: persistent forward, f16-pair BPTT chain, all-T planes, plane decode step,
    per-step decode kernel) and at the reference's own size (whole-sequence training kernels, fused greedy / beam decode), against
    the float64 oracle - results must stay FINITE and inside the usual bars;
  * the two benchmarked shapes that were parity-tested below their benchmarked size: configs[4] as named
    (2-layer biLSTM encoder + 2-layer LSTM decoder) at bench.py's batch of 1024, and the plane-step CLaSS chain at 131 072 rows.
"""
import os

import numpy as np
import pytest
import torch

from helpers import cu, set_losses_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


# ------------------------------------------------------------------------------------------------ weight range of the f16-pair engines
def _plant(m, P, values):
    """Overwrite a few entries of every recurrent weight matrix (W_hh of each GRU, the vocabulary projection) with `values`, in the
    model and in the oracle's dict - entries spread over different rows / columns so that different workgroup slices see them."""
    with torch.no_grad():
        for k, prm in m.named_parameters():
            if ".rnn.weight_hh" in k or k == "decoder.fc.1.weight":
                R, C = prm.shape
                for j, v in enumerate(values):
                    r, c = (7 + 37 * j) % R, (3 + 11 * j) % C
                    prm[r, c] = float(v)
                    P[k][r, c] = np.float32(v)


@pytest.mark.parametrize("values", [(300.0, -400.0, 5000.0)], ids=["300,-400,5000"])   # one case: the float64 oracle at this size takes ~25 s of host time
def test_large_recurrent_weights_at_config_b_vs_f64_oracle(values):
    """configs[1] dimensions (B = 2048, h = 512, T = 25) with weights far beyond the old fixed scale's range in every W_hh and in the
    vocabulary projection: loss terms / mu / logits / every gradient against the float64 oracle (bars of _check_step_vs_oracle, with
    the f32 restatement's own departure as slack where the planted weights make the recurrence ill-conditioned), greedy ids of 1024 z
    through the plane decode step and of 256 z through the per-step kernel bit-exact.  /root/reference/models/decoder.py:40-41,77."""
    import test_gpu_tiles as tt
    m, P, ids, rnd = tt._random_case(2048, 25, 24, 510, 512, 1, seed=int(70 + abs(values[0])) % 1000)
    _plant(m, P, values)
    tt._check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf", f64=True, tag=f"config B, planted weights {values}")
    for prm in m.parameters():
        if prm.grad is not None:
            assert bool(torch.isfinite(prm.grad).all())
    tt._check_greedy_vs_oracle(m, P, 1024, 25, seed=81, f64=True, tag=f"config B planes step, planted weights {values}")
    tt._check_greedy_vs_oracle(m, P, 256, 25, seed=82, f64=True, tag=f"config B per-step kernel, planted weights {values}")


def test_large_recurrent_weights_at_reference_dims_vs_f64_oracle():
    """The reference's default sizes (encoder h = 80, decoder h = 102): the whole-sequence training kernels and the fused greedy /
    beam decode kernels hold W_hh (and the vocabulary projection) as f16-pair fragments in registers / LDS; with planted weights of
    300 / -400 the step matches the float64 oracle, greedy ids are bit-exact and beam hypotheses exact."""
    import test_gpu_tiles as tt
    from oracle import decode as odec
    m, P, ids, rnd = tt._random_case(64, 25, 24, 100, 80, 1, seed=91)
    _plant(m, P, (300.0, -400.0))
    tt._check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf", f64=True, tag="config A, planted weights")
    tt._check_greedy_vs_oracle(m, P, 256, 25, seed=92, f64=True, tag="config A fused greedy, planted weights")
    rs = np.random.RandomState(93)
    N = 48
    z = rs.randn(N, 100).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    m.eval()
    got, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='beam', beam_size=5, n_best=3)
    hyps, _, margins = odec.beam(P, z, c, 25, beam_size=5, n_best=3, return_margins=True)
    bad = [i for i in range(N) if [list(map(int, h)) for h in got[i]] != hyps[i]]
    assert all(margins[i] < 2e-5 for i in bad) and len(bad) <= 1, (bad, [margins[i] for i in bad])


def test_weight_exponent_record():
    """cpg_weight_exp: the record's maximum is the matrix' largest magnitude (any layout the callers use: contiguous, column blocks of a
    wider matrix, widths that are no multiple of four), zeros give exponent 0."""
    from cpg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    for shape, cols in (((1536, 512), None), ((306, 252), (150, 252)), ((24, 102), None), ((7, 5), None)):
        w = torch.randn(*shape, generator=g).to(dev)
        w[shape[0] // 2, shape[1] - 1] = -37.5
        v = w if cols is None else w[:, cols[0]:cols[1]]
        rec = ops.weight_exp(v).view(torch.float32)
        assert rec.numel() == 32 and float(rec.max()) == float(v.abs().max())
    assert float(ops.weight_exp(torch.zeros(64, 64, device=dev)).view(torch.float32).max()) == 0.0


# ------------------------------------------------------------------------------------------------ benchmarked shapes at their benchmarked size
def test_configs4_lstm_two_layer_decoder_at_bench_batch_vs_torch_ref():
    """`extra.config_c_lstm` of the bench line - BASELINE.json configs[4] as named: 2-layer bidirectional LSTM encoder + 2-layer LSTM
    decoder, h = 1024, T = 50 - at the batch bench.py times it on (B = 1024: above 512 rows the persistent forward runs as row-range
    launches) against torch.nn.LSTM(num_layers=2) + autograd with every draw injected.  Extension: parity unpinned against the reference
    (it has no LSTM and a one-layer decoder)."""
    import test_gpu_round5 as t5
    set_losses_cfg()
    m, P, ids, rnd = t5._case(1024, 50, 24, 1022, 1024, 2, 2, "lstm", seed=1961)
    beta, lam_l1, lam_kl = 1.5, 0.0, 1e-3
    terms, aux, G = t5._ref_step(P, ids, rnd, "lstm", beta, lam_l1, lam_kl)
    m, out, mu, logits = t5._hip_step(m, ids, rnd, beta, lam_l1, lam_kl)
    for name in ("recon", "kl", "mmdrf", "l1", "klmu", "total"):
        want = float(terms[name].detach())
        assert abs(out[name].item() - want) < 1e-4 * max(1.0, abs(want)), (name, out[name].item(), want)
    np.testing.assert_allclose(mu.detach().cpu().numpy(), aux["mu"].detach().numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), aux["logits"].detach().numpy(), atol=1e-4, rtol=0)
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        want, got = G[k], prm.grad.cpu().numpy()
        np.testing.assert_allclose(got, want, atol=2e-6 + 1e-4 * np.abs(want).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize("mode", ["greedy", "beam"])
def test_class_round_at_config_b_width_131072_rows(mode):
    """`class.config_b_width` of the bench line at its own size: ONE round of 131 072 proposals at z = 510 / decoder h = 512 through
    sample_pipeline.sample_round_arrays - the per-step decode chain on plane images (cpg_gru_step_fwd_planes; beam: re-gather folded
    into the operand loads).  (i) 1/64 shards of the round - drawn, scored and DECODED ALONE (2 048 rows: still the plane step) - equal
    those rows of the big round at its start, middle and very end; (ii) decoded residues of rows spread over the round equal
    oracle.decode's greedy / beam-5 of the same z.  /root/reference/sample_pipeline.py:129-139, models/model.py:225-385."""
    import sample_pipeline as sp
    import test_gpu_round5 as t5
    from bench import model_kwargs
    from cpg import decode as cdecode
    from cpg.synth import SyntheticPeptideLoader
    from density_modeling import mogQ
    from models.model import RNN_VAE
    from models.mutils import EOS_IDX
    from oracle import decode as odec
    dev = torch.device("cuda")
    Z, T = 510, 25
    torch.manual_seed(77)
    m = RNN_VAE(n_vocab=24, max_seq_len=T, **model_kwargs(Z, 32)).to(dev)
    m.device = dev
    with torch.no_grad():
        m.decoder.fc[1].bias[EOS_IDX] += 0.5       # hypotheses of several lengths
    m.eval()
    P = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    rs = np.random.RandomState(4)
    Q = mogQ.from_params(np.ones(4) / 4, 0.3 * rs.randn(4, Z), np.full((4, Z), 0.9))
    Q.init_attr_classifiers({'amp': t5._clf(0.1 * rs.randn(1, Z), np.zeros(1)), 'tox': t5._clf(0.1 * rs.randn(1, Z), np.zeros(1))},
                            clf_targets={'amp': 1, 'tox': 0})
    Q.rng = 'device'
    ds = SyntheticPeptideLoader(4, T, 'cuda', size=16)
    n, W = 1 << 17, 64
    K = 5 if mode == "beam" else 1
    assert cdecode.PlaneStep(m.decoder, K * n, Z + 2, False, nsent=n).ok and cdecode.PlaneStep(m.decoder, K * (n // W), Z + 2, False, nsent=n // W).ok
    Q._philox = [2026, 0]
    frame, st = sp.sample_round_arrays(m, ds, Q, n, sample_mode=mode)
    assert st['proposed'] == n and st['decoded'] == n
    for r in (0, 31, 63):
        Q._philox = [2026, 0]
        f, _ = sp.sample_round_arrays(m, ds, Q, n, sample_mode=mode, shard=(r, W))
        lo, hi = r * n // W, (r + 1) * n // W
        for k in frame:
            a, b = f[k], frame[k][lo:hi]
            if k == 'letters':
                w = min(a.shape[1], b.shape[1])
                assert bool((a[:, w:] == 0).all()) and bool((b[:, w:] == 0).all())
                a, b = a[:, :w], b[:, :w]
            assert torch.equal(a, b), (k, r)
    per = 24 if mode == "beam" else 48
    rows = np.concatenate([np.arange(per), n // 2 + np.arange(per), n - per + np.arange(per)])
    sel = torch.from_numpy(rows).cuda()
    zz = frame['z'][sel].cpu().numpy()
    cbit = frame['c'][sel].cpu().numpy()
    c = np.zeros((len(rows), 2), np.float32)
    c[np.arange(len(rows)), cbit] = 1
    if mode == "greedy":
        ids = odec.greedy(P, zz, c, T)
    else:
        hyps, _ = odec.beam(P, zz, c, T, beam_size=5, n_best=3)
        L = max(len(h[0]) for h in hyps)
        ids = np.full((len(rows), L), -1, np.int64)
        for i, h in enumerate(hyps):
            ids[i, :len(h[0])] = h[0]
    ref_letters, ref_n = sp.residue_rows(torch.from_numpy(ids), ds.n_vocab)
    got_letters, got_n = frame['letters'][sel].cpu().numpy(), frame['n_res'][sel].cpu().numpy()
    assert np.array_equal(ref_n.numpy(), got_n)
    w = min(ref_letters.shape[1], got_letters.shape[1])
    assert np.array_equal(ref_letters.numpy()[:, :w], got_letters[:, :w])
    assert len(set(got_n.tolist())) >= 3


# ------------------------------------------------------------------------------------------------ round-6 launch fusions
@pytest.mark.parametrize("form", ["NT", "NN", "TN"])
def test_gemm_group_vs_f64(form):
    """cpg_gemm_group: several problems of one form in one launch, chained segments, split destinations, bias, accumulation, widths that
    are no multiple of anything (z_dim = 510) - against float64 products of the same f32 inputs."""
    from cpg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    F = {"NT": ops.NT, "NN": ops.NN, "TN": ops.TN}[form]
    probs, checks, alive = [], [], []      # (a problem record holds raw pointers: the operands must outlive the launch)
    shapes = [(2048, 510, 512, 512), (24, 1536, 150, 0), (300, 70, 33, 96), (2048, 1024, 510, 510)]
    for pi, (M, N, K0, K1) in enumerate(shapes):
        segs, ref = [], torch.zeros(M, N, dtype=torch.float64)
        for K in (K0, K1):
            if K == 0:
                continue
            if form == "NT":
                A, Bm = rn(M, K), rn(N, K)
                ref += A.double().cpu() @ Bm.double().cpu().T
                segs.append((A, K, Bm, K, K))
            elif form == "NN":
                A, Bm = rn(M, K), rn(K, N)
                ref += A.double().cpu() @ Bm.double().cpu()
                segs.append((A, K, Bm, N, K))
            else:
                A, Bm = rn(K, M), rn(K, N)
                ref += A.double().cpu().T @ Bm.double().cpu()
                segs.append((A, M, Bm, N, K))
        bias = rn(N) if form != "TN" and pi % 2 == 0 else None
        alive.append((segs, bias))
        acc = pi == 2
        split = pi == 3
        C = rn(M, N if not split else 512) if acc or split else torch.full((M, N), float("nan"), device=dev)
        C2 = torch.full((M, N - 512), float("nan"), device=dev) if split else None
        if split:
            C = torch.full((M, 512), float("nan"), device=dev)
        c0 = C.clone()
        if bias is not None:
            ref += bias.double().cpu()[None, :]
        if acc:
            ref += c0.double().cpu()
        probs.append(ops.gemm_prob(segs, C, M, N, bias=bias, accumulate=acc, C2=C2, n_split=512 if split else 0))
        checks.append((C, C2, ref, sum(s[4] for s in segs)))
    ops.gemm_group(F, probs)
    torch.cuda.synchronize()
    for C, C2, ref, K in checks:
        got = C.double().cpu() if C2 is None else torch.cat([C.double().cpu(), C2.double().cpu()], 1)
        assert torch.isfinite(got).all()
        tol = (3e-6 if form != "TN" else 2e-5) * (K ** 0.5) * max(1.0, ref.abs().max().item() / (K ** 0.5))
        assert (got - ref).abs().max().item() < tol, ((got - ref).abs().max().item(), tol)


def _grads(loss, params):
    gs = torch.autograd.grad(loss, params, allow_unused=True)
    return [None if g is None else g.detach().clone() for g in gs]


def test_token_tables_and_heads_fns_vs_linear_fns():
    """ops.TokenTablesFn / ops.EncoderHeadsFn (grouped launches) against the one-product-per-launch Functions they replace (LinearFn /
    LinearColsFn + torch.cat): values and every gradient, through plain autograd AND with the direct accumulation of
    FusedAdamClip.backward (gradients added straight into existing .grad buffers, embedding row PAD skipped)."""
    from cpg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(21)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev).requires_grad_()
    V, E, H, Z, B = 24, 150, 96, 70, 200
    emb = rn(V, E)
    w_f, b_f, w_r, b_r = rn(3 * H, E), rn(3 * H), rn(3 * H, E), rn(3 * H)
    w_d, b_d = rn(3 * H, E + Z), rn(3 * H)
    up = [torch.randn(V, 3 * H, generator=g).to(dev) for _ in range(3)]

    def tables(fused):
        ew = ops.ZeroRowGradFn.apply(emb, 1)
        if fused:
            tf, tr = ops.TokenTablesFn.apply(ew, (0, E, None, None), w_f, b_f, w_r, b_r)
            td, = ops.TokenTablesFn.apply(ew, (0, E, None, None), w_d, b_d)
        else:
            tf, tr = ops.LinearFn.apply(ew, w_f, b_f), ops.LinearFn.apply(ew, w_r, b_r)
            td = ops.LinearColsFn.apply(ew, w_d, b_d, 0, E)
        return (tf, tr, td), sum((t * u).sum() for t, u in zip((tf, tr, td), up))
    P = [emb, w_f, b_f, w_r, b_r, w_d, b_d]
    (ta, la), (tb, lb) = tables(True), tables(False)
    for x, y in zip(ta, tb):
        assert torch.allclose(x, y, atol=2e-5, rtol=0)
    GA, GB = _grads(la, P), _grads(lb, P)
    for ga, gb in zip(GA, GB):
        assert torch.allclose(ga, gb, atol=2e-5 * max(1.0, gb.abs().max().item()), rtol=0)
    assert float(GA[0][1].abs().max()) == 0.0     # row PAD of the embedding
    # heads
    hf, hr = rn(B, H), rn(B, H)
    wm, bm, wl, bl = rn(Z, 2 * H), rn(Z), rn(Z, 2 * H), rn(Z)
    um, ul = torch.randn(B, Z, generator=g).to(dev), torch.randn(B, Z, generator=g).to(dev)

    def heads(fused):
        if fused:
            mu, lv = ops.EncoderHeadsFn.apply(hf, hr, wm, bm, wl, bl)
        else:
            h = torch.cat([hf, hr], 1)
            mu, lv = ops.LinearFn.apply(h, wm, bm), ops.LinearFn.apply(h, wl, bl)
        return (mu, lv), (mu * um).sum() + (lv * ul).sum()
    Ph = [hf, hr, wm, bm, wl, bl]
    (ha, la), (hb, lb) = heads(True), heads(False)
    for x, y in zip(ha, hb):
        assert torch.allclose(x, y, atol=3e-5, rtol=0)
    for ga, gb in zip(_grads(la, Ph), _grads(lb, Ph)):
        assert torch.allclose(ga, gb, atol=3e-5 * max(1.0, gb.abs().max().item()), rtol=0)
    # direct accumulation (inside backward_scope): existing .grad buffers receive the sums, the pad row of the embedding stays untouched
    for prm in P + Ph:
        prm.grad = torch.full_like(prm, 0.25)
    ew = ops.tag_emb(ops.ZeroRowGradFn.apply(emb, 1), emb, 1)
    tf, tr = ops.TokenTablesFn.apply(ew, (0, E, emb, 1), w_f, b_f, w_r, b_r)
    td, = ops.TokenTablesFn.apply(ew, (0, E, emb, 1), w_d, b_d)
    mu, lv = ops.EncoderHeadsFn.apply(hf, hr, wm, bm, wl, bl)
    loss = sum((t * u).sum() for t, u in zip((tf, tr, td), up)) + (mu * um).sum() + (lv * ul).sum()
    want = _grads(tables(False)[1] + heads(False)[1], P + Ph)
    with ops.backward_scope():
        loss.backward()
    torch.cuda.synchronize()
    for prm, w in zip(P + Ph, want):
        assert torch.allclose(prm.grad - 0.25, w, atol=4e-5 * max(1.0, w.abs().max().item()), rtol=0)
    assert bool((emb.grad[1] == 0.25).all())


def test_latent_fn_vs_unfused_and_stream_equivalence():
    """ops.LatentFn (reparameterisation + class prior + [z;c] + the three analytic penalties in one node) against ReparamFn /
    LatentTermsFn / torch.cat with injected draws - values and gradients of mu, logvar - and, with the draws made INSIDE the kernel,
    against DeviceRng.normal / onehot2 of the same stream position (bit-identical eps and c)."""
    from cpg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(31)
    B, Z = 300, 510
    mu = (torch.randn(B, Z, generator=g) * 0.5).to(dev).requires_grad_()
    lv = (torch.randn(B, Z, generator=g) * 0.5).to(dev).requires_grad_()
    eps = torch.randn(B, Z, generator=g).to(dev)
    c = torch.zeros(B, 2, device=dev)
    c[torch.arange(B), torch.randint(0, 2, (B,), generator=g)] = 1
    uz, uzc = torch.randn(B, Z, generator=g).to(dev), torch.randn(B, Z + 2, generator=g).to(dev)
    wts = (0.7, 1e-3, 0.1)
    z, zc, c2, kl, klmu, l1, s5 = ops.LatentFn.apply(mu, lv, eps, c, None)
    la = (z * uz).sum() + (zc * uzc).sum() + wts[0] * kl + wts[1] * klmu + wts[2] * l1
    zb = ops.ReparamFn.apply(mu, lv, eps)
    zcb = torch.cat([zb, c], 1)
    klb, klmub, l1b = ops.LatentTermsFn.apply(mu, lv, B)
    lb = (zb * uz).sum() + (zcb * uzc).sum() + wts[0] * klb + wts[1] * klmub + wts[2] * l1b
    assert torch.equal(z, zb) and torch.equal(zc, zcb) and torch.equal(c2, c)
    for a, b in ((kl, klb), (klmu, klmub), (l1, l1b)):
        assert abs(a.item() - b.item()) < 1e-5 * max(1.0, abs(b.item()))
    for ga, gb in zip(_grads(la, [mu, lv]), _grads(lb, [mu, lv])):
        assert torch.allclose(ga, gb, atol=1e-6 * max(1.0, gb.abs().max().item()), rtol=0)
    r1, r2 = ops.DeviceRng(77), ops.DeviceRng(77)
    for rr in (r1, r2):
        rr.normal((5,), dev)       # both streams stand at the same, non-zero position
    with torch.no_grad():
        z, zc, cc, *_ = ops.LatentFn.apply(mu.detach(), lv.detach(), None, None, r1)
        e2 = r2.normal((B, Z), dev)
        c_ref = r2.onehot2(B, 0.5, dev)
        assert torch.equal(z, ops.ReparamFn.apply(mu.detach(), lv.detach(), e2)) and torch.equal(cc, c_ref)
        assert torch.equal(zc, torch.cat([z, c_ref], 1)) and r1.offset == r2.offset


@pytest.mark.parametrize("B,He,Z,T", [(2048, 512, 510, 25), (32, 80, 100, 25)], ids=["config-B", "config-A"])
def test_fused_train_forward_vs_oracle(B, He, Z, T):
    """The trainer's forward (model.fused_train + decoder.recon_targets, what train_vae.train_step runs: ops.LatentFn, ops.VocabReconFn,
    the grouped tables / heads) against the oracle: loss terms 1e-4 and EVERY parameter gradient at 2e-6 + 1e-4 max|g| - at configs[1]
    dimensions and at the reference's defaults.  /root/reference/train_vae.py:24-42."""
    import losses
    import test_gpu_tiles as tt
    from cpg.ops import WeightedSumFn
    from oracle import wae
    m, P, ids, rnd = tt._random_case(B, T, 24, Z, He, 1, seed=7 + B)
    set_losses_cfg()
    beta, lam_l1, lam_kl = 1.5, 0.1, 1e-3
    terms, G, aux = wae.train_loss_and_grads(P, ids, rnd, beta, lam_l1, lam_kl, "mmdrf")
    losses.rf.clear()
    losses.rf['gaussian'] = (cu(rnd["rf_w"]), cu(rnd["rf_b"]))
    idt = cu(ids)
    rc = dict(eps=cu(rnd["eps"]), c=cu(rnd["c"]), wd_mask=cu(rnd["wd_mask"]), out_mask=cu(rnd["out_mask"]))
    m.fused_train, m.decoder.recon_targets = True, idt
    try:
        (mu, lv), (z, c), logits = m(idt, q_c='prior', sample_z=1, rnd=rc)
    finally:
        m.fused_train, m.decoder.recon_targets = False, None
    assert hasattr(logits, "_cpg_recon") and hasattr(mu, "_cpg_latent")
    recon = losses.recon_dec(idt, logits)
    kl, klmu, l1 = losses.latent_terms(mu, lv)
    mmdrf = losses.wae_mmd_gaussianprior(z, method='rf', z_prior=cu(rnd["z_prior_rf"]))
    loss = WeightedSumFn.apply((1.0, beta, lam_l1, lam_kl), recon, mmdrf, l1, klmu)
    loss.backward()
    torch.cuda.synchronize()
    slack = {}
    if lam_l1 > 0:
        tt._l1_kink_correction(G, aux, lv.detach().cpu().numpy(), P, lam_l1, slack, "fused")
    for name, got in (("recon", recon), ("kl", kl), ("mmdrf", mmdrf), ("klmu", klmu), ("l1", l1), ("total", loss)):
        ref = float(terms[name])
        assert abs(got.item() - ref) < 1e-4 * max(1.0, abs(ref)), (name, got.item(), ref)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), aux["logits"], atol=1e-4, rtol=0)
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        ref, got = G[k], prm.grad.cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=2e-6 + 1e-4 * np.abs(ref).max() + slack.get("g." + k, 0.0), rtol=0, err_msg=k)


def test_fused_optimizer_step_matches_the_per_segment_launches():
    """FusedAdamClip.step in its two-launch form (cpg_sumsq_segs + cpg_adam_step_segs, csrc/optim.hip) against the per-segment launches
    it replaces, on a model with the reference's duplicate embedding entry, clip active: parameters after five steps agree to
    float rounding of the norm's summation order; the iteration counter and the Philox base advance together."""
    import copy
    import test_gpu_tiles as tt
    from cpg.optim import FusedAdamClip
    m, P, ids, rnd = tt._random_case(64, 25, 24, 100, 80, 1, seed=5)
    m2 = copy.deepcopy(m)
    outs = []
    for mod, fused in ((m, True), (m2, False)):
        opt = FusedAdamClip(mod.vae_params(), lr=1e-3, max_norm=0.05)
        assert opt._fused_ok and opt._ndup == 1
        opt._fused_ok = fused
        g = torch.Generator().manual_seed(3)
        for it in range(5):
            opt.zero_grad()
            for prm in opt.order:
                prm.grad.copy_(torch.randn(prm.shape, generator=g).to(prm.device) * 0.01)
            opt.step()
        torch.cuda.synchronize()
        outs.append(([prm.detach().clone() for prm in opt.order], opt.grad_norm().item(), int(opt.iter_dev.item())))
    (pa, na, ia), (pb, nb, ib) = outs
    assert ia == ib == 5 and abs(na - nb) < 1e-6 * nb
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() < 2e-7


# ------------------------------------------------------------------------------------------------ small recurrences: 16-row tiles
@pytest.mark.parametrize("B,H,T,reverse", [(32, 80, 25, False), (40, 102, 9, True), (2048, 102, 25, False), (5, 64, 4, True)])
def test_small_recurrence_tile_rows_bitwise(B, H, T, reverse):
    """Whole-sequence launches of small GRU recurrences (csrc/decode_fused.hip) tile the batch by 16 rows while every tile still gets a CU
    of its own, by 32 above that (option small_seq_rows).  Rows are independent and both forms run the same instruction sequence per
    row: states, saved gates, gate gradients and the initial-state gradient must be BIT-identical - ragged last tiles included."""
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 3 + H)
    V = 24
    w_hh = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(dev)
    b_hh = (torch.randn(3 * H, generator=g) * 0.1).to(dev)
    tab = (torch.randn(V, 3 * H, generator=g) * 0.4).to(dev)
    rowc = (torch.randn(B, 3 * H, generator=g) * 0.4).to(dev)
    tok = torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev)
    h0 = (torch.randn(B, H, generator=g) * 0.5).to(dev)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    dlast = (torch.randn(B, H, generator=g) * 0.1).to(dev)
    out = {}
    try:
        for rows in (16, 32):
            ops.set_option("small_seq_rows", rows)
            hs = torch.full((T + 1, B, H), float("nan"), device=dev)
            hs[T if reverse else 0] = h0
            gates = torch.full((T, 4, B, H), float("nan"), device=dev)
            call("cpg_gru_seq_fwd", T, B, H, int(reverse), _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), None, _p(hs), _p(gates), 0, B, None,
                 _p(ops.weight_exp(w_hh)), _stream())
            dG = torch.full((T, B, 4 * H), float("nan"), device=dev)
            dh0 = torch.full((B, H), float("nan"), device=dev)
            call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(w_hh), _p(hs), _p(gates), _p(dhs), _p(dlast), _p(dG), _p(torch.empty(2, B, H, device=dev)),
                 _p(dh0), 0, B, None, None, None, 0, _stream())
            torch.cuda.synchronize()
            out[rows] = (hs, gates, dG, dh0)
    finally:
        ops.set_option("small_seq_rows", None)
    for a, b in zip(out[16], out[32]):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ persistent kernels' cell nonlinearities
def test_persistent_cell_nonlinearities_ulp_vs_f64():
    """The persistent sequence kernels evaluate sigmoid / tanh with hardware exp2 / rcp forms (csrc/cpg_common.h: 4 and 15 VALU
    instructions) where torch.nn.GRU's cell (models/encoder.py:25-30, models/decoder.py:40-41) calls the library functions.  Bars:
    sigmoid within 2.5 ulp of 1 in absolute terms (2^-24 per ulp: its outputs are O(1) gate values), tanh within 4 ulp of the
    result's own magnitude; exact saturation and signs at the extremes; NaN stays NaN."""
    from cpg import ops
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(1 << 18, generator=g) * 3.0, torch.randn(1 << 16, generator=g) * 0.2,
                   torch.linspace(-0.26, 0.26, 1 << 14), torch.linspace(-30.0, 30.0, 1 << 14),
                   torch.tensor([0.0, -0.0, 0.25, -0.25, 1e-20, -1e-20, 88.0, -88.0, 200.0, -200.0, float("inf"), float("-inf")])]).cuda()
    sg, th = torch.empty_like(x), torch.empty_like(x)
    ops.call("cpg_persistent_cell_probe", ops._p(x), ops._p(sg), ops._p(th), x.numel(), ops._stream())
    xd = x.double().cpu()
    sref, tref = torch.sigmoid(xd), torch.tanh(xd)
    ulp = 2.0 ** -24
    es = ((sg.double().cpu() - sref).abs() / ulp).max().item()
    fin = torch.isfinite(xd)
    et = ((th.double().cpu() - tref).abs()[fin] / (tref.abs()[fin].clamp_min(1e-30) * 2 * ulp)).max().item()
    assert es <= 2.5, es
    assert et <= 4.0, et
    tail = th[-12:].cpu().tolist()
    assert tail[0] == 0.0 and tail[1] == 0.0 and np.signbit(tail[1]) and tail[6] == 1.0 and tail[7] == -1.0 and tail[10] == 1.0 and tail[11] == -1.0
    assert sg[-1].item() == 0.0 and sg[-2].item() == 1.0 and sg[-3].item() == 0.0 and sg[-4].item() == 1.0
    nan = torch.full((64,), float("nan"), device="cuda")
    ops.call("cpg_persistent_cell_probe", ops._p(nan), ops._p(sg), ops._p(th), 64, ops._stream())
    assert torch.isnan(sg[:64]).all() and torch.isnan(th[:64]).all()
