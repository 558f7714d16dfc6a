"""LSTM cell: no counterpart in the reference (parity unpinned there) - the numpy restatement is pinned to
torch.nn.LSTM on CPU here, and the HIP kernels are compared with the restatement in the GPU test below."""
import numpy as np
import pytest
import torch

from oracle import lstm as olstm


def _torch_ref(x, h0, c0, w_ih, w_hh, b_ih, b_hh, dhs, reverse):
    B, T, I = x.shape
    H = h0.shape[1]
    m = torch.nn.LSTM(I, H, batch_first=True, bidirectional=reverse)
    sfx = "_reverse" if reverse else ""
    with torch.no_grad():
        for n, v in (("weight_ih_l0", w_ih), ("weight_hh_l0", w_hh), ("bias_ih_l0", b_ih), ("bias_hh_l0", b_hh)):
            getattr(m, n + sfx).copy_(torch.from_numpy(v))
    xt = torch.from_numpy(x).requires_grad_()
    h0t, c0t = torch.from_numpy(h0).requires_grad_(), torch.from_numpy(c0).requires_grad_()
    if reverse:
        h0f = torch.stack([torch.zeros_like(h0t), h0t])
        c0f = torch.stack([torch.zeros_like(c0t), c0t])
        out, _ = m(xt, (h0f, c0f))
        out = out[:, :, H:]
    else:
        out, _ = m(xt, (h0t.unsqueeze(0), c0t.unsqueeze(0)))
    (out * torch.from_numpy(dhs)).sum().backward()
    g = {n: getattr(m, n + sfx).grad.numpy() for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")}
    return out.detach().numpy(), xt.grad.numpy(), h0t.grad.numpy(), c0t.grad.numpy(), g


@pytest.mark.parametrize("reverse", [False, True])
def test_oracle_lstm_matches_torch(reverse):
    rs = np.random.RandomState(3)
    B, T, I, H = 5, 7, 6, 9
    x = rs.randn(B, T, I).astype(np.float32)
    h0, c0 = rs.randn(B, H).astype(np.float32) * 0.5, rs.randn(B, H).astype(np.float32) * 0.5
    w_ih, w_hh = rs.randn(4 * H, I).astype(np.float32) * 0.3, rs.randn(4 * H, H).astype(np.float32) * 0.3
    b_ih, b_hh = rs.randn(4 * H).astype(np.float32) * 0.1, rs.randn(4 * H).astype(np.float32) * 0.1
    dhs = rs.randn(B, T, H).astype(np.float32)
    out, dx, dh0, dc0, g = _torch_ref(x, h0, c0, w_ih, w_hh, b_ih, b_hh, dhs, reverse)
    gi = (x.reshape(B * T, I) @ w_ih.T + b_ih).reshape(B, T, 4 * H)
    hs, _, _, caches = olstm.lstm_seq_fwd(gi, h0, c0, w_hh, b_hh, reverse)
    np.testing.assert_allclose(hs, out, atol=2e-6)
    dG, odh0, odc0, dW, db = olstm.lstm_seq_bwd(dhs, caches, w_hh, reverse)
    np.testing.assert_allclose(odh0, dh0, atol=5e-6)
    np.testing.assert_allclose(odc0, dc0, atol=5e-6)
    np.testing.assert_allclose(dW, g["weight_hh_l0"], atol=1e-5)
    np.testing.assert_allclose(db, g["bias_hh_l0"], atol=1e-5)
    flat = dG.reshape(B * T, 4 * H)
    np.testing.assert_allclose(flat.T @ x.reshape(B * T, I), g["weight_ih_l0"], atol=1e-5)
    np.testing.assert_allclose((flat @ w_ih).reshape(B, T, I), dx, atol=5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,H,V,reverse", [(7, 6, 12, 9, False), (70, 5, 36, 24, True), (130, 4, 64, 24, False),
                                             # BASELINE.json configs[1] size ("hidden=512 1-layer LSTM, batch=2048, seq_len<=25"): here the
                                             # launcher itself picks the persistent forward, the direct-to-LDS backward step and the
                                             # 256 x 128 split-K dW tiles - against oracle/lstm.py (itself pinned to torch.nn.LSTM)
                                             (2048, 25, 512, 24, False), (2048, 25, 512, 24, True),
                                             # BASELINE.json configs[4] width and length (hidden=1024, seq_len<=50): per-step
                                             # forward kernels (the LSTM slice of that width does not fit a CU's LDS)
                                             (128, 50, 1024, 24, False), (128, 50, 1024, 24, True)])
def test_hip_lstm_sequence_matches_oracle(B, T, H, V, reverse):
    from cpg import ops
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    rs = np.random.RandomState(B + H)
    tab = (rs.randn(V, 4 * H) * 0.5).astype(np.float32)
    rowc = (rs.randn(B, 4 * H) * 0.5).astype(np.float32)
    tok = rs.randint(0, V, (T, B)).astype(np.int32)
    h0, c0 = (rs.randn(B, H) * 0.5).astype(np.float32), (rs.randn(B, H) * 0.5).astype(np.float32)
    w_hh = (rs.randn(4 * H, H) / H ** 0.5).astype(np.float32)
    b_hh = (rs.randn(4 * H) * 0.1).astype(np.float32)
    dslab = (rs.randn(T + 1, B, H) * 0.3).astype(np.float32)
    cu = lambda a: torch.from_numpy(a).cuda()
    tt = {k: cu(v).requires_grad_() for k, v in dict(tab=tab, rowc=rowc, h0=h0, c0=c0, w_hh=w_hh, b_hh=b_hh).items()}
    if H == 512:
        assert ops.lstm_persistent_fits(B, H)
    slab = ops.LstmSeqFn.apply(cu(tok), tt["tab"], tt["rowc"], None, tt["h0"], tt["c0"], tt["w_hh"], tt["b_hh"], T, reverse)
    slab.backward(cu(dslab))
    ops.check_persistent()
    gi = (tab[tok] + rowc[None]).transpose(1, 0, 2)  # [B,T,4H]
    hs, _, _, caches = olstm.lstm_seq_fwd(gi, h0, c0, w_hh, b_hh, reverse)
    got = slab.detach().cpu().numpy()
    out_t = got[1:] if not reverse else got[:T]
    np.testing.assert_allclose(out_t.transpose(1, 0, 2), hs, atol=2e-5)
    dext = (dslab[1:] if not reverse else dslab[:T]).transpose(1, 0, 2)
    dG, dh0, dc0, dW, db = olstm.lstm_seq_bwd(dext, caches, w_hh, reverse)
    dh0 = dh0 + (dslab[T] if reverse else dslab[0])
    tol = lambda ref: 5e-6 + 2e-4 * np.abs(ref).max()
    np.testing.assert_allclose(tt["h0"].grad.cpu().numpy(), dh0, atol=tol(dh0))
    np.testing.assert_allclose(tt["c0"].grad.cpu().numpy(), dc0, atol=tol(dc0))
    np.testing.assert_allclose(tt["w_hh"].grad.cpu().numpy(), dW, atol=tol(dW))
    np.testing.assert_allclose(tt["b_hh"].grad.cpu().numpy(), db, atol=tol(db))
    drowc = dG.sum(1)
    np.testing.assert_allclose(tt["rowc"].grad.cpu().numpy(), drowc, atol=tol(drowc))
    dtab = np.zeros_like(tab)
    np.add.at(dtab, tok.T.reshape(-1), dG.reshape(B * T, 4 * H))
    np.testing.assert_allclose(tt["tab"].grad.cpu().numpy(), dtab, atol=tol(dtab))


@pytest.mark.gpu
def test_lstm_model_trains_and_decodes():
    """End-to-end plumbing of cell='lstm' (extension; parity unpinned): loss finite and decreasing, greedy decode runs."""
    import bench
    import cfg
    import losses
    import train_vae as tv
    from cpg.synth import synth_ids
    from models.model import RNN_VAE
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = RNN_VAE(n_vocab=24, max_seq_len=25, **bench.model_kwargs(30, 32, cell='lstm')).to(dev)
    m.device = dev
    m.use_device_rng(5)
    losses.rf.clear()
    losses.set_prior_sampler(lambda z: m._randn(z.shape[0], z.shape[1]))
    cfgv = cfg.Bunch(lr=1e-3, clip_grad=5.0, z_regu_loss='mmdrf', lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
                     beta=cfg.Bunch(start=cfg.Bunch(val=1.0, iter=0), end=cfg.Bunch(val=2.0, iter=1000)))
    opt = tv.make_optimizer(cfgv, m)
    ids = synth_ids(64, 25, 24, torch.Generator().manual_seed(1)).to(dev)
    first = last = None
    for it in range(30):
        out = tv.train_step(cfgv, m, opt, ids, it)
        v = out["L_vae_recon"].item()
        assert np.isfinite(v)
        first = v if first is None else first
        last = v
    assert last < first - 0.1
    z = torch.randn(16, 30, device=dev)
    c = torch.eye(2, device=dev)[torch.zeros(16, dtype=torch.long)]
    s, _, _ = m.generate_sentences(16, z, c, sample_mode='greedy')
    assert s.shape[0] == 16 and s.dtype == torch.int64 and (s[:, 0] == 2).all()
    losses.rf.clear()
    losses.set_prior_sampler(None)


@pytest.mark.gpu
def test_lstm_beam_and_greedy_vs_oracle():
    """Beam-5 / n-best-3 and greedy decoding with cell='lstm' against the numpy restatement (oracle/decode.py with the
    torch.nn.LSTM-pinned cell of oracle/lstm.py): hypotheses exact.  Extension: parity unpinned against the reference."""
    import bench
    from models.model import RNN_VAE
    from oracle import decode as odec
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    dev = torch.device("cuda")
    torch.manual_seed(3)
    m = RNN_VAE(n_vocab=24, max_seq_len=25, **bench.model_kwargs(46, 32, cell='lstm')).to(dev)
    m.device = dev
    with torch.no_grad():
        m.decoder.fc[1].weight.mul_(5.0)
        m.decoder.fc[1].bias[3] += 1.0
    P = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    rs = np.random.RandomState(2)
    N = 48
    z = rs.randn(N, 46).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    zt, ct = torch.from_numpy(z).to(dev), torch.from_numpy(c).to(dev)
    hyps, _, _ = m.generate_sentences(N, zt, ct, sample_mode='beam', beam_size=5, n_best=3)
    ref, _ = odec.beam(P, z, c, 25, beam_size=5, n_best=3, cell="lstm")
    for i in range(N):
        for j in range(3):
            assert hyps[i][j] == [int(t) for t in ref[i][j]], (i, j)


# ---- persistent whole-sequence LSTM forward kernel (csrc/lstm_persist.hip) against the per-step kernels
def _lstm_inputs(B, H, T, V, seed, dense=False, rowc=True):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda")
    return dict(
        w_hh=(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev), b_hh=(torch.randn(4 * H, generator=g) * 0.1).to(dev),
        tab=(torch.randn(V, 4 * H, generator=g) * 0.3).to(dev),
        rowc=(torch.randn(B, 4 * H, generator=g) * 0.3).to(dev) if rowc else None,
        dense=(torch.randn(T, B, 4 * H, generator=g) * 0.3).to(dev) if dense else None,
        tok=torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev),
        h0=(torch.randn(B, H, generator=g) * 0.5).to(dev), c0=(torch.randn(B, H, generator=g) * 0.5).to(dev))


def _lstm_run(d, B, H, T, reverse, persistent):
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    hs, cs = torch.zeros(T + 1, B, H, device=dev), torch.zeros(T + 1, B, H, device=dev)
    hs[T if reverse else 0] = d["h0"]
    cs[T if reverse else 0] = d["c0"]
    gates = torch.zeros(T, 4, B, H, device=dev)
    if persistent:
        assert ops.lstm_persistent_fits(B, H)
        ops.lstm_seq_fwd_persistent(T, B, H, reverse, d["w_hh"], d["b_hh"], d["tok"], d["tab"], d["rowc"], d["dense"], hs, cs, gates)
        ops.check_persistent()
    else:
        call("cpg_lstm_seq_fwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(d["b_hh"]), _p(d["tok"]), _p(d["tab"]), _p(d["rowc"]),
             _p(d["dense"]), _p(hs), _p(cs), _p(gates), _stream())
    torch.cuda.synchronize()
    return hs, cs, gates


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,reverse,dense,rowc", [
    (2048, 512, 25, False, False, True),    # bench decoder shape
    (2048, 512, 25, True, False, False),    # bench encoder shape, reverse direction
    (200, 96, 6, False, False, True),       # partial row tile, three k-blocks
    (333, 128, 9, True, True, False),       # dense input term (upper layers), reverse
    (64, 512, 50, False, False, True),
    (1000, 256, 3, False, True, True),
    (256, 1024, 4, False, False, True),     # configs[4] width: covered since the f16-pair planes (32 x 1024 x 4 B = 131 KB of LDS; the bf16 triple needed 196)
    (1024, 1024, 3, True, True, False),     # ... up to 1024 rows per launch (256 co-resident workgroups)
])
def test_lstm_persistent_forward_matches_per_step(B, H, T, reverse, dense, rowc):
    """State slabs (h, c) and saved gates of every step against the per-step kernels.  Option f32_engine = bf16x3: same split
    products and cell formulas, sums over k-blocks in the same order - agreement to f32 rounding of the six-term order (<= 5e-6 on
    O(1) values).  Default engine (f16 pair, csrc/gemm_core.h): another f32-grade decomposition of the same products (closer to
    the exact sums: tests/test_gpu_persistent.py::test_persistent_engines_vs_f64), the two differ by the sum of their roundings."""
    from cpg import ops
    d = _lstm_inputs(B, H, T, 24, seed=B + H + T, dense=dense, rowc=rowc)
    hq, cq, gq = _lstm_run(d, B, H, T, reverse, False)
    for engine, k in (("f16x2", 3.0), ("bf16x3", 1.0)):
        with ops.options(f32_engine=engine):
            if not ops.lstm_persistent_fits(B, H):    # three bf16 planes do not fit the LDS at H = 1024
                assert engine == "bf16x3" and H > 512
                continue
            hp, cp, gp = _lstm_run(d, B, H, T, reverse, True)
        assert torch.isfinite(hp).all() and torch.isfinite(cp).all()
        assert (hp - hq).abs().max().item() < k * 5e-6, engine
        assert (cp - cq).abs().max().item() < k * 2e-5, engine
        assert (gp - gq).abs().max().item() < k * 5e-6, engine


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,reverse,dense,rowc,ng", [
    (2048, 512, 25, False, False, True, 4),   # bench decoder shape: four groups of 8 units per workgroup (32 units, 128 rows)
    (2048, 512, 25, True, False, False, 4),
    (200, 96, 6, False, False, True, 4),      # partial row tiles of 16 rows
    (333, 128, 9, True, True, False, 4),
    (1000, 80 * 2, 5, False, True, True, 4),  # H = 160: 5 column tiles of 32 units
    (512, 48 * 2, 4, False, False, True, 4),
    (300, 16 * 6, 7, True, False, True, 4),
    (1024, 8 * 36, 3, False, False, True, 4), # H = 288
    (256, 8 * 20, 3, False, False, True, 4),
    (640, 16 * 10, 4, True, False, False, 4),
])
def test_lstm_persistent_forward_bf16_mode_groups(B, H, T, reverse, dense, rowc, ng):
    """bf16 compute mode: the persistent kernel keeps ONE W_hh plane and so holds up to four 8-unit groups per workgroup
    (lstm_seq_fwd_persist_kernel<1, NG>, round 4).  Every group count computes each unit's gates with the same k order from the
    same bf16-rounded operands, so the wide forms equal the one-group form (option lstm_persist_groups=1) BIT FOR BIT; and the
    mode's result stays within its bar of the f32-grade per-step kernels (whose step product is never rounded to bf16).  The kernel
    name the launcher reports carries the group count."""
    import ctypes
    from cpg import ops, lib
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    d = _lstm_inputs(B, H, T, 24, seed=B + H + T, dense=dense, rowc=rowc)
    ops.set_compute_mode('bf16')
    try:
        buf = ctypes.create_string_buffer(128)
        lib().dll.cpg_lstm_persistent_kernel_name(B, H, buf, 128)
        name = buf.value.decode()
        assert name.startswith("lstm_seq_fwd_persist_kernel<1, "), name
        wide = int(name.split(",")[1].strip(" >"))
        assert wide in (1, 2, 4) and H % (8 * wide) == 0
        hp, cp, gp = _lstm_run(d, B, H, T, reverse, True)
        outs = {}
        for cap in (1, 2):
            with ops.options(lstm_persist_groups=cap):
                lib().dll.cpg_lstm_persistent_kernel_name(B, H, buf, 128)
                got = int(buf.value.decode().split(",")[1].strip(" >"))
                assert got <= cap
                outs[cap] = _lstm_run(d, B, H, T, reverse, True)
        hq, cq, gq = _lstm_run(d, B, H, T, reverse, False)
    finally:
        ops.set_compute_mode('f32')
    assert torch.isfinite(hp).all() and torch.isfinite(cp).all()
    for cap, (h1, c1, g1) in outs.items():
        assert torch.equal(hp, h1) and torch.equal(cp, c1) and torch.equal(gp, g1), cap
    assert (hp - hq).abs().max().item() < 2e-2
    assert (cp - cq).abs().max().item() < 4e-2
    assert (gp - gq).abs().max().item() < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("B,H", [(333, 128), (77, 256), (1001, 96)])
def test_lstm_persistent_odd_batch_changing_data(B, H):
    """Odd batch sizes, NEW data on the same scratch every launch (a repeat of the same inputs hides a stale read: the stale bytes
    are the right ones).  Round 4 regression: with odd B the last row of one k-block of an exchange plane and row 0 of the next
    shared a 128-byte line - two row tiles with their own arrival counters - and row 0's tile could read a copy cached before its
    producers wrote it; exchange rows are now padded to an even count."""
    from cpg import ops
    T = 9
    for mode in ("f32", "bf16"):
        ops.set_compute_mode(mode)
        try:
            for seed in range(5):
                d = _lstm_inputs(B, H, T, 24, seed=100 + seed, dense=bool(seed & 1), rowc=not (seed & 1))
                hp, cp, gp = _lstm_run(d, B, H, T, bool(seed & 2), True)
                hq, cq, gq = _lstm_run(d, B, H, T, bool(seed & 2), False)
                tol = 2e-5 if mode == "f32" else 3e-2
                assert (hp - hq).abs().max().item() < tol, (mode, seed)
                assert (gp - gq).abs().max().item() < tol, (mode, seed)
        finally:
            ops.set_compute_mode("f32")


@pytest.mark.gpu
def test_lstm_persistent_forward_vs_oracle_and_repeatable():
    B, H, T, V = 130, 64, 7, 24
    d = _lstm_inputs(B, H, T, V, seed=5)
    hs, cs, gates = _lstm_run(d, B, H, T, False, True)
    gi = (d["tab"].cpu().numpy()[d["tok"].cpu().numpy().T] + d["rowc"].cpu().numpy()[:, None, :]).astype(np.float32)   # [B,T,4H]
    hs_ref, _, _, _ = olstm.lstm_seq_fwd(gi, d["h0"].cpu().numpy(), d["c0"].cpu().numpy(), d["w_hh"].cpu().numpy(), d["b_hh"].cpu().numpy())
    np.testing.assert_allclose(hs[1:].permute(1, 0, 2).cpu().numpy(), hs_ref, atol=3e-6)
    d2 = _lstm_inputs(2048, 512, 25, V, seed=2)
    a = _lstm_run(d2, 2048, 512, 25, False, True)
    for _ in range(2):
        b = _lstm_run(d2, 2048, 512, 25, False, True)
        assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
def test_lstm_persistent_limits_and_option():
    from cpg import ops
    assert ops.lstm_persistent_fits(2048, 512)
    assert not ops.lstm_persistent_fits(2048, 100)     # H % 32 != 0
    assert not ops.lstm_persistent_fits(8192, 512)     # 1024 workgroups
    assert ops.lstm_persistent_fits(1024, 1024)        # configs[4] width (f16-pair planes: round 4)
    assert not ops.lstm_persistent_fits(2048, 1024)    # 512 workgroups of 131 KB LDS
    with ops.options(f32_engine="bf16x3"):
        assert not ops.lstm_persistent_fits(1024, 1024)    # three bf16 planes of 32 gate rows x 1024: 196 KB
    with ops.options(lstm_persist=0):
        assert not ops.lstm_persistent_fits(2048, 512)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,reverse", [(128, 64, 5, False), (192, 96, 4, True), (2048, 512, 6, False)])
def test_lstm_backward_direct_to_lds_kernel_is_the_step_arithmetic(B, H, T, reverse):
    """lstm_step_bwd_dl_kernel (global_load_lds ring, W_hh^T) against lstm_step_bwd_kernel: same products in the same contraction
    order; the cell backward is written on 4-vectors there and on scalars here (fused-multiply-add contraction may differ): dG,
    dh0, dc0 within 2e-6 of their scale."""
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    d = _lstm_inputs(B, H, T, 24, seed=B + H)
    hs, cs, gates = _lstm_run(d, B, H, T, reverse, False)
    g = torch.Generator().manual_seed(3)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    out = []
    from cpg import ops
    for dl in (1, 0):
        ops.set_option("lstm_bwd_dl", dl)
        dG = torch.zeros(T, B, 4 * H, device=dev)
        scr = torch.empty(2, B, H, device=dev)
        dh0, dc0 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
        wT = torch.empty(H, 4 * H, device=dev)
        call("cpg_lstm_seq_bwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(cs), _p(gates), _p(dhs), _p(dG), _p(scr), _p(dh0), _p(dc0),
             _p(wT), None, _stream())
        torch.cuda.synchronize()
        out.append((dG, dh0, dc0))
    ops.set_option("lstm_bwd_dl", None)
    for a, b in zip(*out):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,reverse", [(128, 64, 5, False), (192, 96, 4, True), (2048, 512, 25, False), (2048, 512, 25, True)])
def test_lstm_backward_f16_pair_step_and_wgrad_vs_exact(B, H, T, reverse):
    """The f16-pair form of the LSTM backward step (lstm_step_bwd_dl_kernel<.., 3>, csrc/pair_engine.h: as the GRU's) and the dW_hh
    product on f16 pairs behind it, against the exact-f32 step / the split-engine product: dG, dh0, dc0 and dW_hh within a few
    1e-6 of their scale, incl. gradients whose magnitude differs by many orders between column groups."""
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    dev = torch.device("cuda")
    d = _lstm_inputs(B, H, T, 24, seed=B + H + 1)
    hs, cs, gates = _lstm_run(d, B, H, T, reverse, False)
    g = torch.Generator().manual_seed(3)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    colexp = torch.randint(-12, 4, (H // 32,), generator=g).repeat_interleave(32).float().to(dev)
    for wild in (False, True):
        dh = dhs * (10.0 ** colexp)[None, None, :] if wild else dhs
        res = []
        for pair in (True, False):
            dG = torch.zeros(T, B, 4 * H, device=dev)
            scr = torch.empty(2, B, H, device=dev)
            dh0, dc0 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
            wT = torch.empty(H, 4 * H, device=dev)
            ps = ops._pair_scratch(B, H, 1, dev, lstm=True) if pair else None
            assert not pair or ps is not None
            if ps is not None:
                ps.fill_(0xFF)
            call("cpg_lstm_seq_bwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(cs), _p(gates), _p(dh), _p(dG), _p(scr), _p(dh0), _p(dc0),
                 _p(wT), _p(ps), _stream())
            ws = torch.empty(query("cpg_gru_wgrad_workspace", T, B, H, 24), device=dev, dtype=torch.uint8)
            dw = torch.zeros(4 * H, H, device=dev)
            call("cpg_lstm_wgrad_hh", T, B, H, int(reverse), _p(dG), _p(hs), _p(dw), None, 0, _p(ws), ws.numel(), _p(ps), _stream())
            torch.cuda.synchronize()
            res.append((dG, dh0, dc0, dw))
        (dG, dh0, dc0, dw), (rG, r0, rc0, rw) = res
        assert torch.isfinite(dG).all() and torch.isfinite(dw).all()
        blk = rG.abs().view(T, B // 32, 32, 4 * H).amax(dim=(2, 3), keepdim=True).expand(T, B // 32, 32, 4 * H).reshape(T, B, 4 * H)
        assert ((dG - rG).abs() <= 4e-6 * blk + 1e-37).all(), wild
        assert (dh0 - r0).abs().max().item() <= 4e-6 * r0.abs().max().item()
        assert (dc0 - rc0).abs().max().item() <= 4e-6 * rc0.abs().max().item()
        # dW_hh: against an f64 sum over the pair run's own dG, per 32-row group of the gate axis
        hprev = (hs[1:] if reverse else hs[:-1]).reshape(T * B, H).double()
        ref = dG.reshape(T * B, 4 * H).double().T @ hprev
        grp = ref.abs().view(4, H // 32, 32, H).amax(dim=(2, 3), keepdim=True).expand(4, H // 32, 32, H).reshape(4 * H, H)
        assert ((dw.double() - ref).abs() <= 3e-6 * grp + 1e-37).all(), wild


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,T,reverse", [(128, 128, 5, False), (256, 128, 4, True), (2048, 512, 25, False), (2048, 512, 25, True), (1024, 1024, 6, True)])
def test_lstm_all_t_planes_form_vs_ping_pong(B, H, T, reverse):
    """All-T planes form of the LSTM chain (cpg_lstm_seq_bwd_ap / cpg_lstm_biseq_bwd_ap / cpg_lstm_wgrad_hh_ap, round 5): dG, dh0, dc0
    are bit-identical to the ping-pong chain (same kernels, same arithmetic: only where the images live changes); dW_hh against an f64
    sum over that dG per 32-unit group of the gate axis, with gradient magnitudes spread over 16 orders, an all-zero column group and
    an all-zero row block (images of all-zero groups must read as zeros); the paired launches equal the single ones bit for bit."""
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    dev = torch.device("cuda")
    assert query("cpg_lstm_ap_bytes", T, B, H) > 0
    d = _lstm_inputs(B, H, T, 24, seed=B + H + 5)
    hs, cs, gates = _lstm_run(d, B, H, T, reverse, False)
    g = torch.Generator().manual_seed(5)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    colexp = torch.randint(-12, 4, (H // 32,), generator=g).repeat_interleave(32).float().to(dev)
    dh = dhs * (10.0 ** colexp)[None, None, :]
    dh[:, 32:64, :] = 0.0          # a row block without any gradient ...
    dh[:, :, 64:96] = 0.0          # ... and a column group that receives none from outside
    ws = torch.empty(query("cpg_gru_wgrad_workspace", T, B, H, 24), device=dev, dtype=torch.uint8)

    def run(allt):
        dG = torch.zeros(T, B, 4 * H, device=dev)
        scr = torch.empty(2, B, H, device=dev)
        dh0, dc0 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
        wT = torch.empty(H, 4 * H, device=dev)
        dw = torch.full((4 * H, H), float("nan"), device=dev)
        if allt:
            ap = ops._ap_scratch(T, B, H, 1, dev, lstm=True)
            ap.fill_(0xFF)   # NaN halves / garbage exponents wherever the chain does not write
            call("cpg_lstm_seq_bwd_ap", T, B, H, int(reverse), _p(d["w_hh"]), _p(cs), _p(gates), _p(dh), _p(dG), _p(scr), _p(dh0), _p(dc0),
                 _p(wT), _p(ap), _stream())
            call("cpg_lstm_wgrad_hh_ap", T, B, H, int(reverse), _p(ap), _p(hs), _p(dw), 0, _p(ws), ws.numel(), _stream())
        else:
            ps = ops._pair_scratch(B, H, 1, dev, lstm=True)
            call("cpg_lstm_seq_bwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(cs), _p(gates), _p(dh), _p(dG), _p(scr), _p(dh0), _p(dc0),
                 _p(wT), _p(ps), _stream())
            call("cpg_lstm_wgrad_hh", T, B, H, int(reverse), _p(dG), _p(hs), _p(dw), None, 0, _p(ws), ws.numel(), _p(ps), _stream())
        torch.cuda.synchronize()
        return dG, dh0, dc0, dw

    (dG, dh0, dc0, dw), (rG, r0, rc0, rw) = run(True), run(False)
    assert torch.equal(dG, rG) and torch.equal(dh0, r0) and torch.equal(dc0, rc0)
    assert torch.isfinite(dw).all()
    hprev = (hs[1:] if reverse else hs[:-1]).reshape(T * B, H).double()
    ref = dG.reshape(T * B, 4 * H).double().T @ hprev
    grp = ref.abs().view(4, H // 32, 32, H).amax(dim=(2, 3), keepdim=True).expand(4, H // 32, 32, H).reshape(4 * H, H)
    assert ((dw.double() - ref).abs() <= 3e-6 * grp + 1e-37).all()
    # accumulate = 1 adds onto what is there
    dw2 = torch.ones(4 * H, H, device=dev)
    ap = ops._ap_scratch(T, B, H, 1, dev, lstm=True)
    dG2 = torch.zeros(T, B, 4 * H, device=dev)
    scr, wT = torch.empty(2, B, H, device=dev), torch.empty(H, 4 * H, device=dev)
    a0, c0 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    call("cpg_lstm_seq_bwd_ap", T, B, H, int(reverse), _p(d["w_hh"]), _p(cs), _p(gates), _p(dh), _p(dG2), _p(scr), _p(a0), _p(c0), _p(wT), _p(ap), _stream())
    call("cpg_lstm_wgrad_hh_ap", T, B, H, int(reverse), _p(ap), _p(hs), _p(dw2), 1, _p(ws), ws.numel(), _stream())
    torch.cuda.synchronize()
    assert ((dw2.double() - 1.0 - ref).abs() <= 3e-6 * grp + 2.5e-7).all()
    # dG = null: the images are the only copy - same dW_hh bit for bit; the input-side reductions read them (cpg_lstm_dgi_reduce_ap)
    ap3 = ops._ap_scratch(T, B, H, 1, dev, lstm=True)
    ap3.fill_(0xFF)
    a1, c1 = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    call("cpg_lstm_seq_bwd_ap", T, B, H, int(reverse), _p(d["w_hh"]), _p(cs), _p(gates), _p(dh), None, _p(scr), _p(a1), _p(c1), _p(wT), _p(ap3), _stream())
    dw3 = torch.zeros(4 * H, H, device=dev)
    call("cpg_lstm_wgrad_hh_ap", T, B, H, int(reverse), _p(ap3), _p(hs), _p(dw3), 0, _p(ws), ws.numel(), _stream())
    dtab, dsum, drowc = torch.zeros(24, 4 * H, device=dev), torch.zeros(4 * H, device=dev), torch.zeros(B, 4 * H, device=dev)
    call("cpg_lstm_dgi_reduce_ap", T, B, H, _p(ap3), _p(d["tok"]), 24, _p(dtab), _p(dsum), _p(drowc), 0, _p(ws), ws.numel(), _stream())
    torch.cuda.synchronize()
    assert torch.equal(dw3, dw) and torch.equal(a1, dh0) and torch.equal(c1, dc0)
    G64 = dG.double()
    oh = torch.nn.functional.one_hot(d["tok"].long().reshape(-1), 24).double()
    rt, rs_, rr = oh.T @ G64.reshape(T * B, 4 * H), G64.sum((0, 1)), G64.sum(0)
    at, as_, ar = oh.T @ G64.abs().reshape(T * B, 4 * H), G64.abs().sum((0, 1)), G64.abs().sum(0)
    # the images hold dG to 2^-22 of each 32 x 32 group's largest value (f16 pairs); sums in f32
    gmax = dG.abs().view(T, B // 32, 32, 4, H // 32, 32).amax(dim=(2, 5), keepdim=True).expand(T, B // 32, 32, 4, H // 32, 32).reshape(T, B, 4 * H).double()
    qt, qs, qr = oh.T @ gmax.reshape(T * B, 4 * H), gmax.sum((0, 1)), gmax.sum(0)
    assert ((dtab.double() - rt).abs() <= 3e-6 * at + 5e-7 * qt + 1e-37).all()
    assert ((dsum.double() - rs_).abs() <= 3e-6 * as_ + 5e-7 * qs + 1e-37).all()
    assert ((drowc.double() - rr).abs() <= 3e-6 * ar + 5e-7 * qr + 1e-37).all()
    if reverse:
        return
    # paired launches: the forward direction of a (this sequence, its own copy run as the reverse direction of another) pair
    d2 = _lstm_inputs(B, H, T, 24, seed=B + H + 6)
    hs2, cs2, gates2 = _lstm_run(d2, B, H, T, True, False)
    apb = ops._ap_scratch(T, B, H, 2, dev, lstm=True)
    dGf, dGr = torch.zeros(T, B, 4 * H, device=dev), torch.zeros(T, B, 4 * H, device=dev)
    sc = torch.empty(2, 2, B, H, device=dev)
    wT2 = torch.empty(2, H, 4 * H, device=dev)
    call("cpg_lstm_biseq_bwd_ap", T, B, H, _p(d["w_hh"]), _p(d2["w_hh"]), _p(cs), _p(cs2), _p(gates), _p(gates2), _p(dh), _p(dh), None, None,
         _p(dGf), _p(dGr), _p(sc[0]), _p(sc[1]), _p(wT2[0]), _p(wT2[1]), _p(apb[0]), _p(apb[1]), _stream())
    dwf = torch.zeros(4 * H, H, device=dev)
    call("cpg_lstm_wgrad_hh_ap", T, B, H, 0, _p(apb[0]), _p(hs), _p(dwf), 0, _p(ws), ws.numel(), _stream())
    torch.cuda.synchronize()
    assert torch.equal(dGf, dG) and torch.equal(dwf, dw)


@pytest.mark.gpu
@pytest.mark.parametrize("B,He,Z,T,layers", [(64, 32, 30, 9, 1), (2048, 512, 510, 25, 1), (256, 128, 126, 12, 2)])
def test_lstm_model_step_vs_torch_ref(B, He, Z, T, layers):
    """A whole WAE training step of the LSTM extension (BASELINE.json configs[1] names an LSTM: encoder biLSTM, LSTM decoder with
    h0 = [z;c], c0 = 0) against oracle/torch_ref.RefWAE(cell='lstm') - torch.nn.LSTM + autograd on the CPU - with every random draw
    injected: loss terms 1e-4, mu / logits, and EVERY parameter gradient at 2e-6 + 1e-4 max|g|, incl. configs[1] size
    (B = 2048, h = 512, T = 25).  **Parity unpinned against the reference** (it has no LSTM): this pins the model-level
    composition - token tables, final-state read-out, decoder initial state, weight-gradient accumulation - to torch.nn.LSTM."""
    import sys
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__)))
    from bench import model_kwargs
    from cpg.synth import synth_ids
    from helpers import cu, set_losses_cfg
    from models.model import RNN_VAE
    from oracle import torch_ref
    import losses
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    set_losses_cfg()
    V = 24
    torch.manual_seed(500 + B)
    m = RNN_VAE(n_vocab=V, max_seq_len=T, **model_kwargs(Z, He, enc_layers=layers, cell='lstm'))
    P = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    ref = torch_ref.RefWAE.from_state(P, cell="lstm")
    rs = np.random.RandomState(B)
    ids = synth_ids(B, T, V, torch.Generator().manual_seed(B))
    c = np.zeros((B, 2), np.float32)
    c[np.arange(B), rs.randint(0, 2, B)] = 1
    rnd = dict(eps=rs.randn(B, Z).astype(np.float32), c=c, wd_mask=(rs.rand(B, T) < 0.3).astype(np.uint8),
               out_mask=(rs.rand(B, T, Z + 2) >= 0.3).astype(np.uint8), z_prior_rf=rs.randn(B, Z).astype(np.float32),
               rf_w=rs.randn(Z, 500).astype(np.float32), rf_b=(2 * np.pi * rs.rand(500)).astype(np.float32))
    beta, lam_l1, lam_kl = 1.5, 0.05, 1e-3
    torch.set_num_threads(min(32, __import__("os").cpu_count() or 1))
    rt = {k: torch.from_numpy(v) for k, v in rnd.items()}
    terms, aux = torch_ref.train_loss(ref, ids, rt, beta, lam_l1, lam_kl, "mmdrf", full_mmd=False)
    terms["total"].backward()
    G = {ref.ref_name(k): p.grad.numpy() for k, p in ref.named_parameters()}
    m = m.cuda()
    m.device = torch.device("cuda")
    losses.rf.clear()
    losses.rf['gaussian'] = (cu(rnd["rf_w"]), cu(rnd["rf_b"]))
    idt = ids.cuda()
    (mu, lv), (z, cc), logits = m(idt, q_c='prior', sample_z=1,
                                  rnd=dict(eps=cu(rnd["eps"]), c=cu(rnd["c"]), wd_mask=cu(rnd["wd_mask"]), out_mask=cu(rnd["out_mask"])))
    recon = losses.recon_dec(idt, logits)
    mmdrf = losses.wae_mmd_gaussianprior(z, method='rf', z_prior=cu(rnd["z_prior_rf"]))
    l1, klmu, kl = losses.logvar_l1(lv), losses.kl_gaussian_sharedmu(mu, lv), losses.kl_gaussianprior(mu, lv)
    loss = recon + beta * mmdrf + lam_l1 * l1 + lam_kl * klmu
    loss.backward()
    torch.cuda.synchronize()
    for name, got in (("recon", recon), ("kl", kl), ("mmdrf", mmdrf), ("l1", l1), ("klmu", klmu), ("total", loss)):
        want = float(terms[name].detach()) if torch.is_tensor(terms[name]) else float(terms[name])
        assert abs(got.item() - want) < 1e-4 * max(1.0, abs(want)), (name, got.item(), want)
    np.testing.assert_allclose(mu.detach().cpu().numpy(), aux["mu"].detach().numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), aux["logits"].detach().numpy(), atol=1e-4, rtol=0)
    checked = 0
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        want, got = G[k], prm.grad.cpu().numpy()
        np.testing.assert_allclose(got, want, atol=2e-6 + 1e-4 * np.abs(want).max(), rtol=0, err_msg=k)
        checked += 1
    assert checked == len(G)


@pytest.mark.gpu
def test_lstm_soft_modes_and_forward_sample_vs_oracle():
    """The soft sampling modes (models/model.py:337-359) and the single decode step forward_sample (models/decoder.py:86-109) with
    cell='lstm': ids exact and soft rows 2e-5 against oracle.decode.soft_sample(cell='lstm'); forward_sample takes and returns the
    state as torch.nn.LSTM's (h, c) pair and one call equals one step of the greedy decode.  Extension: parity unpinned."""
    import bench
    from models.model import RNN_VAE
    from oracle import decode as odec
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    dev = torch.device("cuda")
    torch.manual_seed(5)
    m = RNN_VAE(n_vocab=24, max_seq_len=25, **bench.model_kwargs(46, 32, cell='lstm')).to(dev)
    m.device = dev
    with torch.no_grad():
        m.decoder.fc[1].weight.mul_(6.0)
        m.decoder.fc[1].bias[3] += 1.0
    P = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    rs = np.random.RandomState(4)
    N = 40
    z = rs.randn(N, 46).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    zt, ct = torch.from_numpy(z).to(dev), torch.from_numpy(c).to(dev)
    for mode, temp in (("greedy_softmax", 1.0), ("greedy_softmax", 0.7), ("none_softmax", 1.0)):
        (ids, soft), _, _ = m.generate_sentences(N, zt, ct, sample_mode=mode, temp=temp)
        ref_ids, ref_soft = odec.soft_sample(P, z, c, 25, mode, temp=temp, cell="lstm")
        assert np.array_equal(ids.cpu().numpy(), ref_ids), mode
        np.testing.assert_allclose(soft.cpu().numpy(), ref_soft, atol=2e-5, err_msg=mode)
    # forward_sample: (h, c) in, (h, c) out; first greedy step
    m.eval()
    h0 = m.decoder.init_hidden(zt, ct).unsqueeze(0)
    tok = torch.full((N,), 2, device=dev, dtype=torch.long)
    logits, (h1, c1) = m.decoder.forward_sample(None, tok, zt, ct, (h0, torch.zeros_like(h0)))
    ref_logits, ref_h, ref_c = odec.lstm_decoder_step(P, np.full(N, 2), np.concatenate([z, c], 1), np.concatenate([z, c], 1), np.zeros((N, 48), np.float32))
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits, atol=2e-5)
    np.testing.assert_allclose(h1[0].cpu().numpy(), ref_h, atol=5e-6)
    np.testing.assert_allclose(c1[0].cpu().numpy(), ref_c, atol=5e-6)
    with pytest.raises(AssertionError):
        m.decoder.forward_sample(None, tok, zt, ct, h0)          # a bare h is the GRU decoder's form


@pytest.mark.gpu
def test_lstm_bf16_mode_step_agreement():
    """The LSTM extension in the bf16 compute mode (BASELINE.json configs[1] as named: LSTM + bf16): forward / backward step products
    and dW_hh with bf16-rounded operands (round 4: the backward step too - lstm_step_bwd_dl_kernel<.., 1>), f32 accumulation and
    storage.  Agreement against torch.nn.LSTM on the CPU, the bf16 mode's own bars (tests/test_gpu_bf16.py): loss terms within 2e-3,
    every gradient within 3 % relative L2."""
    import os
    from bench import model_kwargs
    from cpg import ops
    from cpg.synth import synth_ids
    from helpers import cu, set_losses_cfg
    from models.model import RNN_VAE
    from oracle import torch_ref
    import losses
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    set_losses_cfg()
    B, He, Z, T, V = 256, 128, 126, 12, 24
    torch.manual_seed(77)
    m = RNN_VAE(n_vocab=V, max_seq_len=T, **model_kwargs(Z, He, cell='lstm'))
    P = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    ref = torch_ref.RefWAE.from_state(P, cell="lstm")
    rs = np.random.RandomState(7)
    ids = synth_ids(B, T, V, torch.Generator().manual_seed(7))
    c = np.zeros((B, 2), np.float32)
    c[np.arange(B), rs.randint(0, 2, B)] = 1
    rnd = dict(eps=rs.randn(B, Z).astype(np.float32), c=c, wd_mask=(rs.rand(B, T) < 0.3).astype(np.uint8),
               out_mask=(rs.rand(B, T, Z + 2) >= 0.3).astype(np.uint8), z_prior_rf=rs.randn(B, Z).astype(np.float32),
               rf_w=rs.randn(Z, 500).astype(np.float32), rf_b=(2 * np.pi * rs.rand(500)).astype(np.float32))
    terms, _ = torch_ref.train_loss(ref, ids, {k: torch.from_numpy(v) for k, v in rnd.items()}, 1.5, 0.0, 1e-3, "mmdrf", full_mmd=False)
    terms["total"].backward()
    G = {ref.ref_name(k): p.grad.numpy() for k, p in ref.named_parameters()}
    m = m.cuda()
    m.device = torch.device("cuda")
    ops.set_compute_mode('bf16')
    try:
        losses.rf.clear()
        losses.rf['gaussian'] = (cu(rnd["rf_w"]), cu(rnd["rf_b"]))
        idt = ids.cuda()
        (mu, lv), (z, cc), logits = m(idt, q_c='prior', sample_z=1,
                                      rnd=dict(eps=cu(rnd["eps"]), c=cu(rnd["c"]), wd_mask=cu(rnd["wd_mask"]), out_mask=cu(rnd["out_mask"])))
        recon = losses.recon_dec(idt, logits)
        loss = recon + 1.5 * losses.wae_mmd_gaussianprior(z, method='rf', z_prior=cu(rnd["z_prior_rf"])) + 1e-3 * losses.kl_gaussian_sharedmu(mu, lv)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.set_compute_mode('f32')
    assert abs(recon.item() - float(terms["recon"])) < 2e-3 and abs(loss.item() - float(terms["total"])) < 2e-3
    worst = 0.0
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        want = G[k].astype(np.float64)
        worst = max(worst, float(np.linalg.norm(prm.grad.cpu().numpy() - want) / max(np.linalg.norm(want), 1e-30)))
    assert 1e-6 < worst < 3e-2, worst
