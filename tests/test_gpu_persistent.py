"""Persistent whole-sequence GRU kernels (csrc/gru_persist.hip) against the per-step kernels and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


@pytest.fixture(autouse=True)
def _per_step_kernels():
    """This module compares the per-step and the persistent kernels with each other, partly bit for bit, also at widths <= 128 where a
    whole-batch cpg_gru_seq_* call would take round 5's whole-sequence launch (tests/test_gpu_round5.py compares that one): off here."""
    from cpg import ops
    ops.set_option("gru_small_seq", 0)
    yield
    ops.set_option("gru_small_seq", None)


def _inputs(B, H, T, V, seed, dense=False, rowc=True):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda")
    d = dict(
        w_hh=(torch.randn(3 * H, H, generator=g) / H ** 0.5).to(dev), b_hh=(torch.randn(3 * H, generator=g) * 0.1).to(dev),
        tab=(torch.randn(V, 3 * H, generator=g) * 0.3).to(dev),
        rowc=(torch.randn(B, 3 * H, generator=g) * 0.3).to(dev) if rowc else None,
        dense=(torch.randn(T, B, 3 * H, generator=g) * 0.3).to(dev) if dense else None,
        tok=torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev), h0=torch.randn(B, H, generator=g).to(dev))
    return d


def _run(d, B, H, T, reverse, persistent):
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    hs = torch.zeros(T + 1, B, H, device=dev)
    hs[T if reverse else 0] = d["h0"]
    gates = torch.zeros(T, 4, B, H, device=dev, dtype=ops.gates_dtype(B, H))
    if persistent:
        assert ops.persistent_fits(B, H)
        ops.gru_seq_fwd_persistent(T, B, H, reverse, d["w_hh"], d["b_hh"], d["tok"], d["tab"], d["rowc"], d["dense"], hs, gates)
        ops.check_persistent()
    else:
        call("cpg_gru_seq_fwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(d["b_hh"]), _p(d["tok"]), _p(d["tab"]), _p(d["rowc"]),
             _p(d["dense"]), _p(hs), _p(gates), 0, B, None, _p(ops.weight_exp(d["w_hh"])), _stream())
    torch.cuda.synchronize()
    return hs, gates


@pytest.mark.parametrize("B,H,T,reverse,dense,rowc", [
    (2048, 512, 25, False, False, True),    # bench decoder shape
    (2048, 512, 25, True, False, False),    # bench encoder shape, reverse direction
    (200, 96, 6, False, False, True),       # partial row tile, H = 96 (3 k-blocks: odd count)
    (333, 128, 9, True, True, False),       # dense input term (upper encoder layers), ragged last tile
    (64, 512, 50, False, False, True),      # one row tile, T = 50
    (1000, 256, 12, False, True, True),
    (256, 1024, 6, False, False, True),     # configs[4] width: 8 hidden units per workgroup ([r|z], [n|n] column blocks)
    (300, 768, 5, True, True, False),       # H = 768 also takes the 8-unit form; partial row tile, dense input term
    (1100, 1024, 4, False, False, True),    # wider than one launch holds at H = 1024 (512 rows): consecutive row ranges
    (4096, 512, 3, True, False, False),     # the same at H = 512 (2048 rows per launch)
])
def test_persistent_forward_matches_per_step(B, H, T, reverse, dense, rowc):
    """Same cell formulas and the same product engine as the per-step kernel with 64-row split tiles (option f32_engine selects it for
    both: f16 pairs by default, three bf16 planes) - results agree to f32 rounding of the reordered k-block sums (bit-identical is not
    required: the per-step launcher may pick the exact engine for small shapes).  Across engines (persistent on pairs against per-step
    on the triple) the two f32-grade decompositions differ by the sum of their roundings (test_persistent_engines_vs_f64: the pair is
    the closer one to the exact sums)."""
    from cpg import ops
    d = _inputs(B, H, T, 24, seed=B + H + T, dense=dense, rowc=rowc)
    per_step = {}
    for engine in ("f16x2", "bf16x3"):
        with ops.options(f32_engine=engine):
            per_step[engine] = _run(d, B, H, T, reverse, False)
            hs_p, g_p = _run(d, B, H, T, reverse, True)
        hs_s, g_s = per_step[engine]
        assert torch.isfinite(hs_p).all()
        assert (hs_p - hs_s).abs().max().item() < 5e-6, engine
        assert (g_p - g_s).abs().max().item() < 5e-6, engine
        if engine == "f16x2":
            pair = (hs_p, g_p)
    assert (pair[0] - per_step["bf16x3"][0]).abs().max().item() < 1.5e-5
    assert (pair[1] - per_step["bf16x3"][1]).abs().max().item() < 1.5e-5


@pytest.mark.parametrize("B,H", [(2048, 512), (512, 1024), (256, 96)])
def test_persistent_engines_vs_f64(B, H):
    """One step, the linear output h . W_hn^T + b_hn (saved gate 3) against the same sum in f64: the f16-pair engine of the
    persistent kernel is at least as close as a plain f32 matrix product (torch.mm on the GPU) and as the bf16-triple engine."""
    from cpg import ops
    d = _inputs(B, H, 1, 24, seed=7)
    W, b, h0 = d["w_hh"].double().cpu(), d["b_hh"].double().cpu(), d["h0"].double().cpu()
    truth = h0 @ W[2 * H:].T + b[2 * H:]

    def err(x):
        e = (x.double().cpu() - truth).abs()
        return e.max().item(), (e ** 2).mean().sqrt().item()

    f32_max, f32_rms = err(d["h0"] @ d["w_hh"][2 * H:].T + d["b_hh"][2 * H:])
    res = {}
    for engine in ("f16x2", "bf16x3"):
        with ops.options(f32_engine=engine):
            res[engine] = err(_run(d, B, H, 1, False, True)[1][0, 3])
    assert res["f16x2"][1] <= 1.1 * f32_rms and res["f16x2"][0] <= 1.5 * f32_max, (res, f32_max, f32_rms)
    assert res["bf16x3"][1] <= 1.25 * f32_rms, (res, f32_rms)
    assert res["f16x2"][1] <= 1.05 * res["bf16x3"][1], res


@pytest.mark.parametrize("B,H", [(333, 128), (77, 256), (1001, 96), (131, 1024)])
def test_persistent_odd_batch_changing_data(B, H):
    """Odd batch sizes with NEW data on the same scratch every launch (repeating the same inputs would hide a stale read of the
    exchange).  Exchange rows are padded to an even count so that no 128-byte line holds rows of two row tiles (round 4, found in
    the LSTM kernel: tests/test_lstm.py::test_lstm_persistent_odd_batch_changing_data)."""
    from cpg import ops
    T = 9
    for mode in ("f32", "bf16"):
        ops.set_compute_mode(mode)
        try:
            for seed in range(5):
                d = _inputs(B, H, T, 24, seed=100 + seed, dense=bool(seed & 1), rowc=not (seed & 1))
                hs_p, g_p = _run(d, B, H, T, bool(seed & 2), True)
                hs_s, g_s = _run(d, B, H, T, bool(seed & 2), False)
                tol = 1.5e-5 if mode == "f32" else 6e-2
                assert (hs_p - hs_s).abs().max().item() < tol, (mode, seed)
                assert (g_p.float() - g_s.float()).abs().max().item() < tol, (mode, seed)
        finally:
            ops.set_compute_mode("f32")


def test_persistent_forward_vs_oracle():
    """Against the numpy GRU restatement (oracle/gru.py) directly."""
    from oracle.gru import gru_seq_fwd
    B, H, T, V = 130, 64, 7, 24
    d = _inputs(B, H, T, V, seed=5)
    hs, _ = _run(d, B, H, T, False, True)
    gi = (d["tab"].cpu().numpy()[d["tok"].cpu().numpy().T] + d["rowc"].cpu().numpy()[:, None, :]).astype(np.float32)  # [B,T,3H]
    ref, _, _ = gru_seq_fwd(gi, d["h0"].cpu().numpy(), d["w_hh"].cpu().numpy(), d["b_hh"].cpu().numpy())
    np.testing.assert_allclose(hs[1:].permute(1, 0, 2).cpu().numpy(), ref, atol=5e-6)


def test_persistent_is_deterministic_and_repeatable():
    B, H, T = 2048, 512, 25
    d = _inputs(B, H, T, 24, seed=1)
    a, ga = _run(d, B, H, T, False, True)
    for _ in range(3):
        b, gb = _run(d, B, H, T, False, True)
        assert torch.equal(a, b) and torch.equal(ga, gb)


def test_option_disables_persistent_path_and_limits():
    from cpg import ops
    with ops.options(gru_persist=0):
        assert not ops.persistent_fits(2048, 512)
    assert ops.persistent_fits(2048, 512) and ops.persistent_rows(512) == 2048     # 16 units x 256 rows per CU, 256 CUs
    assert ops.persistent_rows(1024) == 512 and ops.persistent_rows(768) >= 512  # 8 units per workgroup
    assert ops.persistent_fits(8192, 512)        # four launches over row ranges
    assert not ops.persistent_fits(2048, 102)    # H % 32 != 0
    assert not ops.persistent_fits(256, 2048)    # the 24 x H plane slice does not fit the LDS
    assert ops.query("cpg_gru_persistent_fits", 2048, 512) == 1 and ops.query("cpg_gru_persistent_fits", 4096, 512) == 0   # ONE launch
    with pytest.raises(ops.CpgError):
        ops.set_option("no_such_option", 1)


def test_persistent_timeout_is_loud():
    """A timed-out wave sets the sticky error word in the scratch AND a host-mapped copy: the host sees it without a copy or a
    synchronisation - at the next launch on that scratch, and in check_persistent()."""
    from cpg import ops
    B, H, T = 256, 64, 3
    d = _inputs(B, H, T, 24, seed=3)
    _run(d, B, H, T, False, True)
    key = next(k for k in ops._persist_scratch if k[0] == "gru" and k[3:] == (B, H))
    ent = ops._persist_scratch[key]
    assert ent[2].is_pinned() and ent[2][0] == 0
    try:
        ent[2][0] = 1                                  # what a timed-out wave writes through the host-mapped pointer
        with pytest.raises(ops.CpgError, match="timed out"):
            _run(d, B, H, T, False, True)              # noticed before the next launch on this scratch
        with pytest.raises(ops.CpgError, match="timed out"):
            ops.check_persistent()
    finally:
        del ops._persist_scratch[key]


# ---- backward step kernels: direct-to-LDS loop (every tile) and the two-K-halves form against the register-staged kernel
def _bwd_inputs(d, B, H, T, reverse, seed, dh_last=True):
    dev = torch.device("cuda")
    hs, gates = _run(d, B, H, T, reverse, False)
    g = torch.Generator().manual_seed(seed)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    last = (torch.randn(B, H, generator=g) * 0.1).to(dev) if dh_last else None
    return hs, gates, dhs, last


def _bwd(d, B, H, T, reverse, hs, gates, dhs, last, with_wT=True, pair=False):
    """pair: hand the scratch of the f16-pair step (the training path does, cpg/ops.py); without it the exact-f32 product runs."""
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    dG, dh0 = torch.zeros(T, B, 4 * H, device=dev), torch.zeros(B, H, device=dev)
    scr = torch.empty(2, B, H, device=dev)
    wT = torch.empty(H, 3 * H, device=dev) if with_wT else None
    ps = ops._pair_scratch(B, H, 1, dev) if pair else None
    assert not pair or ps is not None
    if ps is not None:
        ps.fill_(0xFF)   # NaN patterns: nothing may be read before it is written
    call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(dG), _p(scr), _p(dh0),
         0, B, None, _p(wT), _p(ps), 0, _stream())
    torch.cuda.synchronize()
    if pair == "keep":
        return dG, dh0, ps
    return dG, dh0


@pytest.mark.parametrize("B,H,T,reverse", [(2048, 512, 25, False), (2048, 512, 25, True), (256, 128, 6, False), (192, 96, 5, True)])
def test_backward_direct_to_lds_tiles_match_the_register_staged_kernel(B, H, T, reverse):
    """gru_step_bwd_dl_kernel on each of its four tiles (option gru_bwd_tile) against gru_step_bwd_kernel (gru_bwd_dl = 0): the same
    exact-f32 products in the same contraction order - dG of every step and dh0 bit-identical."""
    from cpg import ops
    d = _inputs(B, H, T, 24, seed=B + H + T + 2)
    hs, gates, dhs, last = _bwd_inputs(d, B, H, T, reverse, seed=3)
    with ops.options(gru_bwd_dl=0, gru_bwd_tile="32x32"):
        ref, ref0 = _bwd(d, B, H, T, reverse, hs, gates, dhs, last)
    assert torch.isfinite(ref).all() and ref.abs().max().item() > 0
    tiles = ["32x32", "64x32"] + (["32x64", "64x64"] if H % 64 == 0 else [])
    for t in tiles:
        with ops.options(gru_bwd_tile=t, gru_bwd_dl2=0):
            dG, dh0 = _bwd(d, B, H, T, reverse, hs, gates, dhs, last)
        bad = (dG != ref).nonzero()
        assert bad.numel() == 0, (t, bad[:8].tolist(), bad.shape[0], (dG - ref).abs().max().item())
        assert torch.equal(dh0, ref0), t
    dG, dh0 = _bwd(d, B, H, T, reverse, hs, gates, dhs, last)     # the launcher's own choice
    assert torch.equal(dG, ref) and torch.equal(dh0, ref0)


@pytest.mark.parametrize("B,H,T,reverse", [(2048, 512, 25, False), (2048, 512, 25, True), (256, 128, 6, False), (192, 96, 5, True),
                                             (1024, 1024, 4, False)])
def test_backward_f16_pair_step_vs_exact(B, H, T, reverse):
    """The f16-pair form of the direct-to-LDS step (PREC 3: dG handed from launch to launch as f16 pairs times a power of two per
    32 x 32 group, three f16 MFMAs per block) against the exact-f32 product, both tiles: dG of every step and dh0 within f32
    rounding of the products (the BPTT chain is 25 steps long: a few 1e-7 of the largest value).  Then the same with gradient
    magnitudes spread over 30 orders (column groups and rows scaled by powers of ten, one group and one row block all zero): every
    element must agree RELATIVE to the largest value of its own 32-row block - the per-group exponents, the accumulator rescaling
    and the skip of all-zero groups."""
    from cpg import ops
    d = _inputs(B, H, T, 24, seed=B + H + T + 2)
    hs, gates, dhs, last = _bwd_inputs(d, B, H, T, reverse, seed=3)
    ref, ref0 = _bwd(d, B, H, T, reverse, hs, gates, dhs, last)
    tiles = ["64x32"] + (["64x64"] if H % 64 == 0 else []) + (["64x64x8"] if H % 64 == 0 else []) + (["128x64"] if H % 64 == 0 and B % 128 == 0 else [])   # eight-wave forms (round 6)
    for t in tiles:
        with ops.options(gru_bwd_tile=t):
            buf = ctypes_name(1, B, H, 1)
            assert ", 3, 3, " in buf and buf.startswith("gru_step_bwd_dl_kernel<%s, " % t.split("x")[0]), buf
            dG, dh0 = _bwd(d, B, H, T, reverse, hs, gates, dhs, last, pair=True)
        assert torch.isfinite(dG).all()
        assert (dG - ref).abs().max().item() <= 2e-6 * ref.abs().max().item(), t
        assert (dh0 - ref0).abs().max().item() <= 2e-6 * ref0.abs().max().item(), t
    # gradients of very different magnitude side by side
    g = torch.Generator().manual_seed(11)
    colexp = torch.randint(-14, 6, (H // 32,), generator=g).repeat_interleave(32).float()
    rowexp = torch.randint(-8, 6, (B // 32,), generator=g).repeat_interleave(32).float()
    scale = (10.0 ** colexp)[None, None, :] * (10.0 ** rowexp)[None, :, None]
    scale[:, :, 32:64] = 0.0
    scale[:, 32:64, :] = 0.0
    dhs2 = dhs * scale.to(dhs.device)
    last2 = last * scale[0].to(dhs.device)
    ref, ref0 = _bwd(d, B, H, T, reverse, hs, gates, dhs2, last2)
    with ops.options(gru_bwd_tile=tiles[-1]):
        dG, dh0 = _bwd(d, B, H, T, reverse, hs, gates, dhs2, last2, pair=True)
    assert torch.isfinite(dG).all() and torch.isfinite(dh0).all()
    blk = ref.abs().view(T, B // 32, 32, 4 * H).amax(dim=(2, 3), keepdim=True).expand(T, B // 32, 32, 4 * H).reshape(T, B, 4 * H)
    assert ((dG - ref).abs() <= 4e-6 * blk).all()
    blk0 = ref0.abs().view(B // 32, 32, H).amax(dim=(1, 2), keepdim=True).expand(B // 32, 32, H).reshape(B, H)
    assert ((dh0 - ref0).abs() <= 4e-6 * blk0).all()
    assert torch.equal(dG[:, 32:64], ref[:, 32:64])   # the all-zero row block stays exactly zero


@pytest.mark.parametrize("B,H,T,reverse", [(2048, 512, 25, False), (2048, 512, 25, True), (1024, 1024, 8, False)])
def test_wgrad_hh_on_f16_pairs_vs_split_engine(B, H, T, reverse):
    """dW_hh = sum_t dG_t^T h_prev(t) with the pair scratch of the sequence's BPTT (gemm_kernel<256x128, ..., 8>: f16 pairs, the
    gate-gradient columns scaled by the power of two the backward steps recorded per 32-column group) against the same call without
    it (three bf16 planes, six MFMAs) and against an f64 sum: both f32-grade, the pair form no further from the f64 result.  Then
    with column groups of very different magnitude: every row group of dW_hh relative to its own largest value."""
    import ctypes
    from cpg import ops, lib
    from cpg.ops import _p, _stream, call, query
    d = _inputs(B, H, T, 24, seed=B + H + T + 4)
    hs, gates, dhs, last = _bwd_inputs(d, B, H, T, reverse, seed=5)
    buf = ctypes.create_string_buffer(160)
    lib().dll.cpg_gemm_tn_kernel_name(T * B, 3 * H, H, 1, buf, 160)
    assert buf.value.decode().endswith(", 8>"), buf.value
    dev = hs.device
    g = torch.Generator().manual_seed(13)
    colexp = torch.randint(-12, 4, (H // 32,), generator=g).repeat_interleave(32).float()
    for wild in (False, True):
        sc = (10.0 ** colexp)[None, None, :].to(dev) if wild else 1.0
        dG, dh0, ps = _bwd(d, B, H, T, reverse, hs, gates, dhs * sc, last * (sc[0] if wild else 1.0), pair="keep")
        ws = torch.empty(query("cpg_gru_wgrad_workspace", T, B, H, 24), device=dev, dtype=torch.uint8)
        outs = []
        for scratch in (ps, None):
            dw = torch.zeros(3 * H, H, device=dev)
            call("cpg_gru_wgrad_hh", T, B, H, int(reverse), _p(dG), _p(hs), _p(dw), None, 0, _p(ws), ws.numel(), _p(scratch), 0, _stream())
            torch.cuda.synchronize()
            outs.append(dw)
        hprev = (hs[1:] if reverse else hs[:-1]).reshape(T * B, H).double()
        ref = (dG[:, :, :3 * H].reshape(T * B, 3 * H).double().T @ hprev)
        grp = ref.abs().view(3, H // 32, 32, H).amax(dim=(2, 3), keepdim=True).expand(3, H // 32, 32, H).reshape(3 * H, H)
        err = [((o.double() - ref).abs() / grp.clamp_min(1e-300)).max().item() for o in outs]
        assert err[0] < 5e-6 and err[1] < 5e-6, (wild, err)   # f32 accumulation over up to 51 200 rows
        assert err[0] < 1.5 * err[1] + 1e-7, (wild, err)


def test_backward_f16_pair_step_edge_values():
    """Edge values through the f16-pair backward step: all-zero gradients (every group takes the all-zero path: results exactly
    zero, nothing read from the unwritten planes), a NaN and an infinity in the incoming gradient (they reach dh0 / dG: loud, as
    with the exact step), and recurrent weights beyond the range of rounds 4-5's fixed 2^8 image scale (|w| >= 256 overflowed the f16
    high half): the image's power of two now follows the matrix' largest magnitude (csrc/gemm_core.h: weight_exp_from_parts), so the
    step stays finite and agrees with the exact-f32 step."""
    from cpg import ops
    B, H, T = 256, 128, 5
    d = _inputs(B, H, T, 24, seed=77)
    hs, gates, dhs, last = _bwd_inputs(d, B, H, T, False, seed=8)
    with ops.options(gru_bwd_tile="64x64"):
        dG, dh0 = _bwd(d, B, H, T, False, hs, gates, torch.zeros_like(dhs), torch.zeros_like(last), pair=True)
        assert not dG.any() and not dh0.any()
        for bad in (float("nan"), float("inf")):
            x = dhs.clone()
            x[T - 1, 3, 5] = bad
            dG, dh0 = _bwd(d, B, H, T, False, hs, gates, x, last, pair=True)
            assert not torch.isfinite(dh0[3]).all() and not torch.isfinite(dG[0, 3]).all()
            assert torch.isfinite(dh0[4:]).all()          # other rows are independent recurrences
        for big in (300.0, -7.0e4, 2.0e6):
            d2 = dict(d)
            d2["w_hh"] = d["w_hh"].clone()
            d2["w_hh"][7, 9] = big
            d2["w_hh"][200, 100] = -big
            dG, dh0 = _bwd(d2, B, H, T, False, hs, gates, dhs, last, pair=True)
            dG_x, dh0_x = _bwd(d2, B, H, T, False, hs, gates, dhs, last, pair=False)
            assert torch.isfinite(dh0).all() and torch.isfinite(dG).all()
            # the pair images carry ONE power of two per 32 rows x 32 columns: an element is exact to 2^-22 of its block's largest
            # value (the engine's documented bar, per step: 4e-6 of the 32-row block's maximum), five steps deep here
            scale = dh0_x.abs().view(B // 32, 32, H).amax(dim=(1, 2), keepdim=True).expand(B // 32, 32, H).reshape(B, H).clamp_min(1e-30)
            # (weights more than 2^16 below the matrix' largest keep an ABSOLUTE precision - 2^-39 of the largest -, so with one
            # weight of 2e6 among weights of 0.1 those carry ~14 bits: the step stays finite and sane, the bar is wider)
            bar = 5e-5 if abs(big) < 1e6 else 4e-4
            assert ((dh0 - dh0_x).abs() / scale).max().item() < bar, (big, ((dh0 - dh0_x).abs() / scale).max().item())


def ctypes_name(kind, B, H, ndir):
    import ctypes
    from cpg import lib
    buf = ctypes.create_string_buffer(160)
    lib().dll.cpg_gru_step_kernel_name(kind, B, H, ndir, 1, buf, 160)
    return buf.value.decode()


@pytest.mark.parametrize("tile", ["32x32", "64x32", "32x64"])
def test_register_staged_backward_tiles_vs_oracle(tile):
    """The register-staged kernel (partial tiles: B = 130, H = 80) on each of its tiles against the numpy BPTT restatement."""
    from cpg import ops
    from oracle.gru import gru_seq_fwd, gru_seq_bwd
    B, H, T, V = 130, 80, 7, 24
    d = _inputs(B, H, T, V, seed=5)
    hs, gates, dhs, _ = _bwd_inputs(d, B, H, T, False, seed=9, dh_last=False)
    with ops.options(gru_bwd_tile=tile):
        dG, dh0 = _bwd(d, B, H, T, False, hs, gates, dhs, None, with_wT=False)
    gi = (d["tab"].cpu().numpy()[d["tok"].cpu().numpy().T] + d["rowc"].cpu().numpy()[:, None, :]).astype(np.float32)
    w, b = d["w_hh"].cpu().numpy(), d["b_hh"].cpu().numpy()
    _, _, caches = gru_seq_fwd(gi, d["h0"].cpu().numpy(), w, b)
    dgi, dh0_ref, dW, db = gru_seq_bwd(dhs.permute(1, 0, 2).cpu().numpy(), None, caches, w)
    got = dG.permute(1, 0, 2).cpu().numpy()                      # [B,T,4H]: dr, dz, dhn, dn_pre
    np.testing.assert_allclose(np.concatenate([got[:, :, :2 * H], got[:, :, 3 * H:]], 2), dgi, atol=2e-6)
    np.testing.assert_allclose(dh0.cpu().numpy(), dh0_ref, atol=2e-6)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,T", [(256, 128, 6), (128, 192, 4)])
def test_backward_two_k_halves_workgroup_matches_the_plain_direct_to_lds_kernel(B, H, T, mode):
    """gru_step_bwd_dl2_kernel (512 threads, two K-halves, partial blocks swapped through LDS; option gru_bwd_dl2 = 1 forces it)
    against gru_step_bwd_dl_kernel (= 0): the same products with one extra reassociation (half sums added once), both directions."""
    from cpg import ops
    d = _inputs(B, H, T, 24, seed=41)
    ops.set_compute_mode(mode)
    try:
        for reverse in (False, True):
            hs, gates, dhs, last = _bwd_inputs(d, B, H, T, reverse, seed=7)
            res = []
            for knob in (0, 1):
                with ops.options(gru_bwd_dl2=knob):
                    res.append(_bwd(d, B, H, T, reverse, hs, gates, dhs, last))
            # bf16 mode: a last-bit difference in dG can flip the bf16 rounding of the next step's operand (2^-9 of one term)
            tol = 2e-6 if mode == "f32" else 1e-3
            for a, b in zip(res[0], res[1]):
                assert not torch.equal(a, torch.zeros_like(a))
                assert (a - b).abs().max().item() <= tol * max(1.0, a.abs().max().item())
    finally:
        ops.set_compute_mode("f32")


# ---- bf16 compute mode: saved gates stored as bf16
@pytest.mark.parametrize("B,H,T", [(256, 128, 6), (2048, 512, 5)])
def test_bf16_mode_saves_gates_as_bf16(B, H, T):
    """In the bf16 compute mode the saved gates of a dense sequence are bf16 (cpg_gru_gates_bf16): the persistent and the per-step
    forward write the same values (RNE of their f32 gates), the direct-to-LDS backward on them stays within bf16 rounding of the
    backward on f32 gates (option bf16_store = 0), and a sequence saved one way is refused by a backward expecting the other."""
    from cpg import ops
    d = _inputs(B, H, T, 24, seed=77)
    ops.set_compute_mode("bf16")
    try:
        assert ops.gates_dtype(B, H) == torch.bfloat16 and ops.gates_dtype(B, H, ragged=True) == torch.float32
        assert ops.gates_dtype(130, 80) == torch.float32            # shape the direct-to-LDS backward does not cover
        hs_p, g_p = _run(d, B, H, T, False, True)
        hs_s, g_s = _run(d, B, H, T, False, False)
        assert g_p.dtype == torch.bfloat16 and g_p.shape == (T, 4, B, H)
        with ops.options(bf16_store=0):
            assert ops.gates_dtype(B, H) == torch.float32
            hs_f, g_f = _run(d, B, H, T, False, True)
        assert torch.equal(hs_p, hs_f)                              # the state does not depend on how the gates are stored
        # layout of the bf16 form: [T,B,H,4] - (r,z,n,hn) of an element adjacent; values = RNE of the f32 gates [T,4,B,H]
        assert torch.equal(g_p.view(T, B, H, 4), g_f.permute(0, 2, 3, 1).to(torch.bfloat16))
        assert (g_p.float() - g_s.float()).abs().max().item() <= 2 ** -7   # per-step kernel: same values up to its own last bits
        g = torch.Generator().manual_seed(5)
        dhs = (torch.randn(T, B, H, generator=g) * 0.1).cuda()
        last = (torch.randn(B, H, generator=g) * 0.1).cuda()
        dG_b, dh0_b = _bwd(d, B, H, T, False, hs_p, g_p, dhs, last)
        with ops.options(bf16_store=0):
            dG_f, dh0_f = _bwd(d, B, H, T, False, hs_f, g_f, dhs, last)
            with pytest.raises(ops.CpgError):                       # bf16 gates handed to a backward that expects f32: refused where visible
                ops._check_gates(g_p, B, H)
        scale = dG_f.abs().max().item()
        assert scale > 0 and (dG_b - dG_f).abs().max().item() <= 2e-2 * scale
        rel = ((dG_b - dG_f).norm() / dG_f.norm()).item()
        assert rel <= 1e-2, rel
        assert (dh0_b - dh0_f).abs().max().item() <= 2e-2 * max(1e-6, dh0_f.abs().max().item())
        with pytest.raises(ops.CpgError, match="bf16"):             # no transposed-weight scratch -> register-staged kernel: refused
            _bwd(d, B, H, T, False, hs_p, g_p, dhs, last, with_wT=False)
    finally:
        ops.set_compute_mode("f32")


def test_handoff_flavour_follows_the_reported_placement():
    """The persistent forward picks its hand-off per launch from the XCD ids the workgroups report: at B=2048, H=512 (one row group
    per XCD under the observed round-robin dispatch) the same-XCD form, at a shape whose six workgroups land on six XCDs the
    write-through form - either way the results are the per-step kernels' (asserted above for every shape); here the recorded
    choice is read back and must be one of the two, and is written to gpurun_out/ for the record."""
    import json
    import os
    from cpg import ops
    seen = {}
    for B, H, T in ((2048, 512, 4), (200, 96, 3)):
        d = _inputs(B, H, T, 24, seed=11)
        hs_p, _ = _run(d, B, H, T, False, True)
        hs_s, _ = _run(d, B, H, T, False, False)
        assert (hs_p - hs_s).abs().max().item() < 5e-6
        ent = next(v for k, v in ops._persist_scratch.items() if k[0] == "gru" and k[3:] == (B, H))
        off = ops.query("cpg_gru_persistent_path_offset", B)
        word = int(ent[0][off:off + 4].view(torch.int32).item())
        assert word in (1, 2), word
        seen[f"B{B}_H{H}"] = {1: "same-XCD (L2-local hand-off)", 2: "write-through"}[word]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(seen, open(os.path.join(ROOT, "gpurun_out", "persist_handoff_report.json"), "w"), indent=1)
