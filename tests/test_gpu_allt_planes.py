"""All-T planes form of the f16-pair BPTT chain (round 5; include/cpg_api.h "All-T planes form", csrc/pair_engine.h ApScratch,
csrc/pair_tn.h): cpg_gru_seq_bwd_ap / _biseq_bwd_ap keep the recurrent gate-gradient blocks of every step ONLY as f16-pair planes;
cpg_gru_wgrad_hh_ap and cpg_gru_dgi_reduce_ap read them.  Checked here, through the C ABI, against the exact-f32 chain
(cpg_gru_seq_bwd without pair scratch -> f32 dG -> f64 sums on the host): the planes dequantise to dG, dh0 / dN agree, dW_hh, the
token-table gradient, the column sums and the sums over time agree - also with gradient magnitudes spread over 30 orders of
magnitude, an all-zero column group and an all-zero row block.  The model-level parity tests (tests/test_gpu_tiles.py part 2,
tests/test_gpu_parity.py) run through this form wherever it covers the shape (config B / C token-table layers)."""
import numpy as np
import pytest
import torch

from test_gpu_persistent import _bwd, _bwd_inputs, _inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


def _dequant(ap, T, B, H):
    """(dG recurrent blocks [T,B,3H] f64, state planes [T,B,H] f64, exponents) from the raw scratch bytes (layout: pair_engine.h)."""
    n_pl, n_hp = T * B * 6 * H, T * B * 2 * H
    raw = ap.cpu().numpy()
    planes = raw[:2 * n_pl].view(np.float16).reshape(T, B, H // 32, 3, 2, 32).astype(np.float64)
    hp = raw[2 * n_pl:2 * (n_pl + n_hp)].view(np.float16).reshape(T, B, H // 32, 2, 32).astype(np.float64)
    ex = raw[2 * (n_pl + n_hp):2 * (n_pl + n_hp) + 4 * T * (B // 32) * (H // 32)].view(np.int32).reshape(T, B // 32, H // 32)
    emin = raw[2 * (n_pl + n_hp) + 4 * T * (B // 32) * (H // 32):][:4 * (H // 32)].view(np.int32)
    e = np.repeat(ex, 32, axis=1).astype(np.float64)                         # [T,B,H/32]
    live = e < 2 ** 30
    sc = np.where(live, 2.0 ** (-np.where(live, e, 0.0)), 0.0)
    v = (planes[..., 0, :] + planes[..., 1, :]) * sc[..., None, None]        # [T,B,cg,q,32]
    dg = v.transpose(0, 1, 3, 2, 4).reshape(T, B, 3 * H)
    h = (hp[..., 0, :] + hp[..., 1, :]).reshape(T, B, H)
    return dg, h, ex, emin


@pytest.mark.parametrize("B,H,T,reverse,wild", [(256, 128, 6, False, False), (256, 128, 6, True, True), (2048, 512, 25, False, False),
                                                (2048, 512, 25, True, True), (512, 1024, 5, False, True)])
def test_all_t_planes_chain_vs_exact_f32(B, H, T, reverse, wild):
    from cpg import ops
    if B * H < 512 * 64 * 64:    # small launches: the policy picks 32-row tiles (no f16-pair step); force the bench shapes' tiles
        with ops.options(gru_bwd_tile="64x32" if wild else "64x64"):
            _chain_vs_exact(B, H, T, reverse, wild)
    else:
        _chain_vs_exact(B, H, T, reverse, wild)


def _chain_vs_exact(B, H, T, reverse, wild):
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    dev = torch.device("cuda")
    V = 24
    d = _inputs(B, H, T, V, seed=B + H + T + 9)
    hs, gates, dhs, last = _bwd_inputs(d, B, H, T, reverse, seed=7)
    if wild:
        g = torch.Generator().manual_seed(11)
        colexp = torch.randint(-14, 6, (H // 32,), generator=g).repeat_interleave(32).float()
        rowexp = torch.randint(-8, 6, (B // 32,), generator=g).repeat_interleave(32).float()
        scale = (10.0 ** colexp)[None, None, :] * (10.0 ** rowexp)[None, :, None]
        scale[:, :, 32:64] = 0.0
        scale[:, 32:64, :] = 0.0
        dhs, last = dhs * scale.to(dev), last * scale[0].to(dev)
    with ops.options(gru_bwd_engine="exact"):
        ref, ref0 = _bwd(d, B, H, T, reverse, hs, gates, dhs, last)        # exact-f32 products, f32 dG [T,B,4H]
    assert query("cpg_gru_ap_bytes", T, B, H, 1) > 0
    ap = ops._ap_scratch(T, B, H, 1, dev)
    ap.fill_(0xFF)                                                          # NaN patterns: nothing may be read before it is written
    dN = torch.full((T, B, H), float("nan"), device=dev)
    dh0 = torch.zeros(B, H, device=dev)
    scr = torch.empty(2, B, H, device=dev)
    wT = torch.empty(H, 3 * H, device=dev)
    call("cpg_gru_seq_bwd_ap", T, B, H, int(reverse), _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(dN), _p(scr), _p(dh0), _p(wT),
         _p(ap), _stream())
    torch.cuda.synchronize()
    refn = ref.cpu().numpy().astype(np.float64)
    blk = np.abs(refn).reshape(T, B // 32, 32, 4 * H).max(axis=(2, 3), keepdims=True).repeat(32, 2).reshape(T, B, 1)
    dg, hpl, ex, emin = _dequant(ap, T, B, H)
    assert np.isfinite(dg).all()
    assert (np.abs(dg - refn[:, :, :3 * H]) <= 4e-6 * blk).all()
    assert (np.abs(dN.cpu().numpy() - refn[:, :, 3 * H:]) <= 4e-6 * blk).all()
    r0 = ref0.cpu().numpy()
    blk0 = np.abs(r0).reshape(B // 32, 32, H).max(axis=(1, 2), keepdims=True).repeat(32, 1).reshape(B, 1)
    assert (np.abs(dh0.cpu().numpy() - r0) <= 4e-6 * blk0).all()
    hprev = (hs[1:] if reverse else hs[:-1]).cpu().numpy().astype(np.float64)
    assert np.abs(hpl - hprev).max() <= 2.0 ** -21 * max(1.0, np.abs(hprev).max())    # unscaled f16 pair of the state
    live = ex < 2 ** 30
    assert (emin == np.where(live, ex, 2 ** 31 - 1).min(axis=(0, 1))).all()
    if wild:
        t_first = 0 if reverse else T - 1      # the step the chain starts from: its gradient is the external one alone
        assert not live[:, 1, :].any() and not live[t_first, :, 1].any()    # the all-zero row block / column group are marked
    # ---- consumers: dW_hh, token-table gradient, column sums, sums over time against f64 sums over the chain's OWN gate gradients
    # (the dequantised images + dN: what the consumers are handed), per 32-column group of the gradient relative to the group's own
    # largest value - the per-segment exponents must keep small column groups exact next to large ones; the chain itself was
    # compared with the exact-f32 chain above (relative to each row block's largest value, the bar of the f16-pair step)
    refn = np.concatenate([dg, dN.cpu().numpy().astype(np.float64)], 2)
    hprev = hpl
    ws = torch.empty(query("cpg_gru_wgrad_workspace", T, B, H, V), device=dev, dtype=torch.uint8)
    dw = torch.full((3 * H, H), 3.0, device=dev)
    call("cpg_gru_wgrad_hh_ap", T, B, H, _p(ap), None, _p(dw), 0, _p(ws), ws.numel(), _stream())
    torch.cuda.synchronize()
    dw2 = dw.clone()
    call("cpg_gru_wgrad_hh_ap", T, B, H, _p(ap), None, _p(dw2), 1, _p(ws), ws.numel(), _stream())      # accumulate: twice the product
    with ops.options(tn_split=1):
        dw3 = torch.zeros(3 * H, H, device=dev)
        call("cpg_gru_wgrad_hh_ap", T, B, H, _p(ap), None, _p(dw3), 0, _p(ws), ws.numel(), _stream())  # no split over the rows
    torch.cuda.synchronize()
    want = refn[:, :, :3 * H].reshape(T * B, 3 * H).T @ hprev.reshape(T * B, H)
    grp = np.abs(want).reshape(3, H // 32, 32, H).max(axis=(2, 3), keepdims=True).repeat(32, 2).reshape(3 * H, 1)
    # (f32 accumulation over up to 51 200 rows: the unsplit launch adds them in ONE chain per element)
    for got, bar in ((dw.cpu().numpy(), 6e-6), (0.5 * dw2.cpu().numpy(), 6e-6), (dw3.cpu().numpy(), 2e-5)):
        assert np.isfinite(got).all()
        assert (np.abs(got - want) <= bar * np.maximum(grp, 1e-300)).all(), float((np.abs(got - want) / np.maximum(grp, 1e-300)).max())
    dtab = torch.zeros(V, 3 * H, device=dev)
    dsum = torch.zeros(4 * H, device=dev)
    drowc = torch.zeros(B, 3 * H, device=dev)
    call("cpg_gru_dgi_reduce_ap", T, B, H, _p(ap), _p(dN), _p(d["tok"]), V, _p(dtab), _p(dsum), _p(drowc), 0, _p(ws), ws.numel(), _stream())
    torch.cuda.synchronize()
    dgi = np.concatenate([refn[:, :, :2 * H], refn[:, :, 3 * H:]], 2)       # input-side blocks dr, dz, dn
    tok = d["tok"].cpu().numpy()
    want_tab = np.zeros((V, 3 * H))
    np.add.at(want_tab, tok.reshape(-1), dgi.reshape(T * B, 3 * H))
    colmax = np.abs(dgi).reshape(T * B, 3, H // 32, 32).max(axis=(0, 3), keepdims=True).repeat(32, 3).reshape(1, 3 * H)
    assert (np.abs(dtab.cpu().numpy() - want_tab) <= 4e-6 * colmax * np.sqrt(T * B)).all()
    want_sum = refn.reshape(T * B, 4 * H).sum(0)
    cm4 = np.abs(refn).reshape(T * B, 4, H // 32, 32).max(axis=(0, 3), keepdims=True).repeat(32, 3).reshape(4 * H)
    assert (np.abs(dsum.cpu().numpy() - want_sum) <= 4e-6 * cm4 * np.sqrt(T * B)).all()
    want_rowc = dgi.sum(0)
    rb = np.abs(dgi).reshape(T, B // 32, 32, 3 * H).max(axis=(0, 2, 3), keepdims=True).repeat(32, 2).reshape(B, 1)
    assert (np.abs(drowc.cpu().numpy() - want_rowc) <= 4e-6 * rb * T).all()


def test_all_t_planes_bidirectional_launches_equal_single_direction():
    """cpg_gru_biseq_bwd_ap (both encoder directions in one launch per step) leaves exactly the images of two cpg_gru_seq_bwd_ap calls."""
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    dev = torch.device("cuda")
    B, H, T = 256, 128, 5
    ops.set_option("gru_bwd_tile", "64x64")
    try:
        _bidirectional(B, H, T)
    finally:
        ops.set_option("gru_bwd_tile", None)


def _bidirectional(B, H, T):
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    dev = torch.device("cuda")
    outs = []
    ds = [_inputs(B, H, T, 24, seed=70 + r) for r in range(2)]
    ins = [_bwd_inputs(ds[r], B, H, T, bool(r), seed=80 + r) for r in range(2)]
    singles = []
    for r in range(2):
        hs, gates, dhs, last = ins[r]
        ap = ops._ap_scratch(T, B, H, 1, dev)
        ap.zero_()
        dN, scr, wT = torch.zeros(T, B, H, device=dev), torch.empty(2, B, H, device=dev), torch.empty(H, 3 * H, device=dev)
        call("cpg_gru_seq_bwd_ap", T, B, H, r, _p(ds[r]["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(dN), _p(scr), None, _p(wT), _p(ap),
             _stream())
        singles.append((ap, dN))
    ap2 = ops._ap_scratch(T, B, H, 2, dev)
    ap2.zero_()
    dNf, dNr = torch.zeros(T, B, H, device=dev), torch.zeros(T, B, H, device=dev)
    sc, wT2 = torch.empty(2, 2, B, H, device=dev), torch.empty(2, H, 3 * H, device=dev)
    call("cpg_gru_biseq_bwd_ap", T, B, H, _p(ds[0]["w_hh"]), _p(ds[1]["w_hh"]), _p(ins[0][0]), _p(ins[1][0]), _p(ins[0][1]), _p(ins[1][1]),
         _p(ins[0][2]), _p(ins[1][2]), _p(ins[0][3]), _p(ins[1][3]), _p(dNf), _p(dNr), _p(sc[0]), _p(sc[1]), _p(wT2[0]), _p(wT2[1]),
         _p(ap2[0]), _p(ap2[1]), _stream())
    torch.cuda.synchronize()
    nb = query("cpg_gru_ap_bytes", T, B, H, 2)
    for r, dn in ((0, dNf), (1, dNr)):
        assert torch.equal(dn, singles[r][1])
        assert torch.equal(ap2[r][:nb], singles[r][0][:nb])


def test_all_t_planes_option_and_coverage():
    """cpg_gru_ap_bytes: 0 where a consumer has no form (H % 128, B % 128, bf16 compute mode) and under option gru_ap = 0; the
    entry points refuse such shapes loudly."""
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    assert query("cpg_gru_ap_bytes", 25, 2048, 512, 1) > 0 and query("cpg_gru_ap_bytes", 25, 2048, 512, 2) > 0
    assert query("cpg_gru_ap_bytes", 25, 2048, 96, 1) == 0 and query("cpg_gru_ap_bytes", 25, 2000, 512, 1) == 0
    with ops.options(gru_ap=0):
        assert query("cpg_gru_ap_bytes", 25, 2048, 512, 1) == 0
    ops.set_compute_mode('bf16')
    try:   # bf16 compute mode: the form is the bf16 state copy next to the mode's bf16 gate gradients
        assert query("cpg_gru_ap_bytes", 25, 2048, 512, 1) == 25 * 2048 * 512 * 2
        with ops.options(bf16_dg=0):
            assert query("cpg_gru_ap_bytes", 25, 2048, 512, 1) == 0
    finally:
        ops.set_compute_mode('f32')
    x = torch.zeros(64, device="cuda")
    with pytest.raises(ops.CpgError):
        call("cpg_gru_wgrad_hh_ap", 4, 64, 96, _p(x), None, _p(x), 0, _p(x), 256, _stream())


@pytest.mark.parametrize("B,H,T,reverse", [(256, 128, 6, False), (2048, 512, 25, True)])
def test_bf16_mode_all_t_form(B, H, T, reverse):
    """bf16 compute mode: cpg_gru_seq_bwd_ap keeps the mode's bf16 gate gradients (bit-identical to cpg_gru_seq_bwd(dg_bf16 = 1)) and
    leaves h_prev of every step rounded to bf16; cpg_gru_wgrad_hh_ap(ap, dG) - one bf16 MFMA per block on operands that ARE bf16 in
    memory - against an f64 sum over exactly those bf16 operands (1e-5 of each 32-row group's largest value: f32 accumulation only) and
    against the mode's register-staged product of the same call (which rounds h to bf16 only in its large-shape one-plane form: 1e-2,
    the size of that rounding)."""
    from cpg import ops
    from cpg.ops import _p, _stream, call, query
    dev = torch.device("cuda")
    V = 24
    ops.set_compute_mode('bf16')
    try:
        if B * H < 512 * 64 * 64:
            ops.set_option("gru_bwd_tile", "64x64")
        d = _inputs(B, H, T, V, seed=B + H + T + 19)
        hs, gates, dhs, last = _bwd_inputs(d, B, H, T, reverse, seed=17)
        assert gates.dtype == torch.bfloat16 and query("cpg_gru_dg_bf16", B, H, 0, V) == 1 and query("cpg_gru_ap_bytes", T, B, H, 1) == T * B * H * 2
        scr, wT = torch.empty(2, B, H, device=dev), torch.empty(H, 3 * H, device=dev)
        ref = torch.zeros(T, B, 4 * H, device=dev, dtype=torch.bfloat16)
        r0 = torch.zeros(B, H, device=dev)
        call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(ref), _p(scr), _p(r0), 0, B, None,
             _p(wT), None, 1, _stream())
        ap = ops._ap_scratch(T, B, H, 1, dev)
        ap.fill_(0xFF)
        dG = torch.zeros(T, B, 4 * H, device=dev, dtype=torch.bfloat16)
        dh0 = torch.zeros(B, H, device=dev)
        call("cpg_gru_seq_bwd_ap", T, B, H, int(reverse), _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(dG), _p(scr), _p(dh0), _p(wT),
             _p(ap), _stream())
        torch.cuda.synchronize()
        assert torch.equal(dG.view(torch.int16), ref.view(torch.int16)) and torch.equal(dh0, r0)
        hb = ap[:T * B * H * 2].view(torch.bfloat16).view(T, B, H)
        hprev = hs[1:] if reverse else hs[:-1]
        assert torch.equal(hb.view(torch.int16), hprev.to(torch.bfloat16).view(torch.int16))           # RNE, as the staged product rounds
        ws = torch.empty(query("cpg_gru_wgrad_workspace", T, B, H, V), device=dev, dtype=torch.uint8)
        dw, dwo = torch.zeros(3 * H, H, device=dev), torch.zeros(3 * H, H, device=dev)
        call("cpg_gru_wgrad_hh_ap", T, B, H, _p(ap), _p(dG), _p(dw), 0, _p(ws), ws.numel(), _stream())
        call("cpg_gru_wgrad_hh", T, B, H, int(reverse), _p(dG), _p(hs), _p(dwo), None, 0, _p(ws), ws.numel(), None, 1, _stream())
        torch.cuda.synchronize()
        want = dG[:, :, :3 * H].reshape(T * B, 3 * H).double().T @ hb.reshape(T * B, H).double()
        grp = want.abs().view(3, H // 32, 32, H).amax(dim=(2, 3), keepdim=True).expand(3, H // 32, 32, H).reshape(3 * H, H)
        assert ((dw.double() - want).abs() <= 1e-5 * grp + 1e-30).all(), float(((dw.double() - want).abs() / (grp + 1e-30)).max())
        assert ((dw - dwo).abs().double() <= 1e-2 * grp + 1e-30).all()
    finally:
        ops.set_option("gru_bwd_tile", None)
        ops.set_compute_mode('f32')
