"""Persistent whole-sequence GRU kernels (csrc/gru_persist.hip) against the per-step kernels and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


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
    gates = torch.zeros(T, 4, B, H, device=dev)
    if persistent:
        assert ops.persistent_fits(B, H)
        ops.gru_seq_fwd_persistent(T, B, H, reverse, d["w_hh"], d["b_hh"], d["tok"], d["tab"], d["rowc"], d["dense"], hs, gates)
        ops.check_persistent()
    else:
        call("cpg_gru_seq_fwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(d["b_hh"]), _p(d["tok"]), _p(d["tab"]), _p(d["rowc"]),
             _p(d["dense"]), _p(hs), _p(gates), 0, B, None, _stream())
    torch.cuda.synchronize()
    return hs, gates


@pytest.mark.parametrize("B,H,T,reverse,dense,rowc", [
    (2048, 512, 25, False, False, True),    # bench decoder shape
    (2048, 512, 25, True, False, False),    # bench encoder shape, reverse direction
    (200, 96, 6, False, False, True),       # partial row tile, H = 96 (3 k-blocks: odd count)
    (333, 128, 9, True, True, False),       # dense input term (upper encoder layers), ragged last tile
    (64, 512, 50, False, False, True),      # one row tile, T = 50
    (1000, 256, 12, False, True, True),
])
def test_persistent_forward_matches_per_step(B, H, T, reverse, dense, rowc):
    """Same split, same MFMA order, same cell formulas as the per-step kernel with 64-row split tiles: results agree to
    f32 rounding of the reordered k-block sums (bit-identical is not required: the per-step launcher may pick the exact
    engine for small shapes), state and saved gates alike."""
    d = _inputs(B, H, T, 24, seed=B + H + T, dense=dense, rowc=rowc)
    hs_p, g_p = _run(d, B, H, T, reverse, True)
    hs_s, g_s = _run(d, B, H, T, reverse, False)
    assert torch.isfinite(hs_p).all()
    assert (hs_p - hs_s).abs().max().item() < 5e-6
    assert (g_p - g_s).abs().max().item() < 5e-6


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


def test_knob_disables_persistent_path():
    from cpg import ops
    os.environ["CPG_GRU_PERSIST"] = "0"
    try:
        assert not ops.persistent_fits(2048, 512)
    finally:
        os.environ.pop("CPG_GRU_PERSIST")
    assert ops.persistent_fits(2048, 512)
    assert not ops.persistent_fits(2048, 1024)   # W_hh slice does not fit LDS
    assert not ops.persistent_fits(2048, 102)    # H % 32 != 0


def _run_bwd(d, B, H, T, reverse, persistent, with_dh0, seed):
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    hs, gates = _run(d, B, H, T, reverse, False)          # forward with the per-step kernels: common saved tensors
    g = torch.Generator().manual_seed(seed)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    dG = torch.zeros(T, B, 4 * H, device=dev)
    dh0 = torch.zeros(B, H, device=dev) if with_dh0 else None
    if persistent:
        assert not ops.persistent_bwd_fits(T, B, H)   # policy: off by default (slower than the per-step kernels)
        ops.gru_seq_bwd_persistent(T, B, H, reverse, d["w_hh"], hs, gates, dhs, None, dG, dh0)
        ops.check_persistent()
    else:
        scr = torch.empty(2, B, H, device=dev)
        os.environ["CPG_GRU_BWD_EXACT"] = "1"
        try:
            call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), None, _p(dG), _p(scr), _p(dh0),
                 0, B, None, None, _stream())
        finally:
            os.environ.pop("CPG_GRU_BWD_EXACT")
    torch.cuda.synchronize()
    return dG, dh0, (hs, gates, dhs)


@pytest.mark.parametrize("B,H,T,reverse,with_dh0", [
    (2048, 512, 25, False, True),     # bench decoder shape (initial state carries a gradient)
    (2048, 512, 25, True, False),     # bench encoder shape, reverse direction
    (200, 96, 6, False, True),
    (333, 128, 9, True, False),
    (64, 512, 50, False, True),
    (1000, 256, 12, False, False),
])
def test_persistent_backward_matches_per_step(B, H, T, reverse, with_dh0):
    """dG of every step and dh0 against the per-step exact-f32 kernels: f32-grade agreement (different product engine:
    split-bf16 vs exact f32; gradients of size ~0.1 agree to ~1e-6)."""
    d = _inputs(B, H, T, 24, seed=B + H + T + 1)
    dG_p, dh0_p, _ = _run_bwd(d, B, H, T, reverse, True, with_dh0, seed=3)
    dG_s, dh0_s, _ = _run_bwd(d, B, H, T, reverse, False, with_dh0, seed=3)
    assert torch.isfinite(dG_p).all()
    scale = dG_s.abs().max().item()
    assert (dG_p - dG_s).abs().max().item() < 2e-6 + 2e-5 * scale
    if with_dh0:
        assert (dh0_p - dh0_s).abs().max().item() < 2e-6 + 2e-5 * dh0_s.abs().max().item()


def test_persistent_backward_vs_oracle():
    """Against the numpy BPTT restatement (oracle/gru.py)."""
    from oracle.gru import gru_seq_fwd, gru_seq_bwd
    B, H, T, V = 130, 64, 7, 24
    d = _inputs(B, H, T, V, seed=5)
    dG, dh0, (hs, gates, dhs) = _run_bwd(d, B, H, T, False, True, True, seed=9)
    gi = (d["tab"].cpu().numpy()[d["tok"].cpu().numpy().T] + d["rowc"].cpu().numpy()[:, None, :]).astype(np.float32)
    w, b = d["w_hh"].cpu().numpy(), d["b_hh"].cpu().numpy()
    _, _, caches = gru_seq_fwd(gi, d["h0"].cpu().numpy(), w, b)
    dgi, dh0_ref, dW, db = gru_seq_bwd(dhs.permute(1, 0, 2).cpu().numpy(), None, caches, w)
    got = dG.permute(1, 0, 2).cpu().numpy()                      # [B,T,4H]: dr, dz, dhn, dn_pre
    np.testing.assert_allclose(np.concatenate([got[:, :, :2 * H], got[:, :, 3 * H:]], 2), dgi, atol=2e-6)
    np.testing.assert_allclose(dh0.cpu().numpy(), dh0_ref, atol=2e-6)


def test_persistent_backward_deterministic():
    B, H, T = 2048, 512, 25
    d = _inputs(B, H, T, 24, seed=2)
    a, a0, _ = _run_bwd(d, B, H, T, False, True, True, seed=4)
    for _ in range(2):
        b, b0, _ = _run_bwd(d, B, H, T, False, True, True, seed=4)
        assert torch.equal(a, b) and torch.equal(a0, b0)


# ---- one-launch BPTT on the step kernel's tiles ("chain", csrc/gru.hip): same arithmetic as the per-step launches
def _run_chain(d, B, H, T, reverse, with_dh0, seed, dh_last=False):
    from cpg import ops
    dev = torch.device("cuda")
    hs, gates = _run(d, B, H, T, reverse, False)
    g = torch.Generator().manual_seed(seed)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    last = (torch.randn(B, H, generator=g) * 0.1).to(dev) if dh_last else None
    dG = torch.zeros(T, B, 4 * H, device=dev)
    dh0 = torch.zeros(B, H, device=dev) if with_dh0 else None
    assert ops.chain_bwd_covers(T, B, H) and not ops.chain_bwd_fits(T, B, H)   # policy: off by default (no faster)
    wT = torch.empty(H, 3 * H, device=dev)
    ops.gru_seq_bwd_chain(T, B, H, reverse, d["w_hh"], hs, gates, dhs, last, dG, dh0, wT)
    ops.check_persistent()
    torch.cuda.synchronize()
    return dG, dh0, (hs, gates, dhs, last)


@pytest.mark.parametrize("B,H,T,reverse,with_dh0", [
    (2048, 512, 25, False, True),     # bench decoder shape
    (2048, 512, 25, True, False),     # bench encoder shape, reverse direction
    (200, 96, 6, False, True),        # partial row tile
    (333, 80, 9, True, False),        # partial row and column tiles (H % 32 != 0)
    (64, 512, 50, False, True),
    (1000, 256, 1, False, True),      # T = 1: nothing to hand over
    (32, 100, 4, True, True),
])
def test_chain_backward_is_the_per_step_arithmetic(B, H, T, reverse, with_dh0):
    """Same tiles, same product engine, same summation order as the per-step launches: dG and dh0 bit-identical."""
    from cpg.ops import _p, _stream, call
    d = _inputs(B, H, T, 24, seed=B + H + T + 2)
    dG_c, dh0_c, (hs, gates, dhs, last) = _run_chain(d, B, H, T, reverse, with_dh0, seed=3, dh_last=True)
    dev = torch.device("cuda")
    dG_s = torch.zeros(T, B, 4 * H, device=dev)
    dh0_s = torch.zeros(B, H, device=dev) if with_dh0 else None
    scr = torch.empty(2, B, H, device=dev)
    os.environ["CPG_GRU_BWD_WIDE"] = "32"   # the chain kernel's tiles (small batches would take another tile / engine)
    try:
        call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(dG_s), _p(scr),
             _p(dh0_s), 0, B, None, None, _stream())
    finally:
        os.environ.pop("CPG_GRU_BWD_WIDE")
    torch.cuda.synchronize()
    assert torch.isfinite(dG_c).all()
    bad = (dG_c != dG_s).nonzero()
    assert bad.numel() == 0, (bad[:8].tolist(), bad.shape[0], (dG_c - dG_s).abs().max().item())
    if with_dh0:
        assert torch.equal(dh0_c, dh0_s)


@pytest.mark.parametrize("B,H,T", [(2048, 512, 25), (150, 96, 7), (64, 80, 3)])
def test_chain_backward_pair_matches_per_step_pair(B, H, T):
    """Both directions of a biGRU layer alternating inside the same workgroups against the paired per-step launches."""
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    df, dr = _inputs(B, H, T, 24, seed=11), _inputs(B, H, T, 24, seed=12)
    hs_f, gt_f = _run(df, B, H, T, False, False)
    hs_r, gt_r = _run(dr, B, H, T, True, False)
    g = torch.Generator().manual_seed(5)
    ext_f, ext_r = ((torch.randn(T, B, H, generator=g) * 0.1).to(dev) for _ in range(2))
    last_f, last_r = ((torch.randn(B, H, generator=g) * 0.1).to(dev) for _ in range(2))   # gradients on the final states
    out = []
    for chain in (True, False):
        dG_f, dG_r = torch.zeros(T, B, 4 * H, device=dev), torch.zeros(T, B, 4 * H, device=dev)
        if chain:
            ops.gru_biseq_bwd_chain(T, B, H, df["w_hh"], dr["w_hh"], hs_f, hs_r, gt_f, gt_r, ext_f, ext_r, dG_f, dG_r,
                                    torch.empty(2, H, 3 * H, device=dev), last_f, last_r)
            ops.check_persistent()
        else:
            sc = torch.empty(2, 2, B, H, device=dev)
            os.environ["CPG_GRU_BWD_WIDE"] = "32"
            try:
                call("cpg_gru_biseq_bwd", T, B, H, _p(df["w_hh"]), _p(dr["w_hh"]), _p(hs_f), _p(hs_r), _p(gt_f), _p(gt_r),
                     _p(ext_f), _p(ext_r), _p(last_f), _p(last_r), _p(dG_f), _p(dG_r), _p(sc[0]), _p(sc[1]), None, None, _stream())
            finally:
                os.environ.pop("CPG_GRU_BWD_WIDE")
        torch.cuda.synchronize()
        out.append((dG_f, dG_r))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_chain_backward_vs_oracle_and_repeatable():
    from oracle.gru import gru_seq_fwd, gru_seq_bwd
    B, H, T, V = 130, 64, 7, 24
    d = _inputs(B, H, T, V, seed=5)
    dG, dh0, (hs, gates, dhs, _) = _run_chain(d, B, H, T, False, True, seed=9)
    gi = (d["tab"].cpu().numpy()[d["tok"].cpu().numpy().T] + d["rowc"].cpu().numpy()[:, None, :]).astype(np.float32)
    w, b = d["w_hh"].cpu().numpy(), d["b_hh"].cpu().numpy()
    _, _, caches = gru_seq_fwd(gi, d["h0"].cpu().numpy(), w, b)
    dgi, dh0_ref, dW, db = gru_seq_bwd(dhs.permute(1, 0, 2).cpu().numpy(), None, caches, w)
    got = dG.permute(1, 0, 2).cpu().numpy()
    np.testing.assert_allclose(np.concatenate([got[:, :, :2 * H], got[:, :, 3 * H:]], 2), dgi, atol=2e-6)
    np.testing.assert_allclose(dh0.cpu().numpy(), dh0_ref, atol=2e-6)
    d2 = _inputs(2048, 512, 25, V, seed=2)
    a, a0, _ = _run_chain(d2, 2048, 512, 25, False, True, seed=4)
    for _ in range(3):
        b2, b0, _ = _run_chain(d2, 2048, 512, 25, False, True, seed=4)
        assert torch.equal(a, b2) and torch.equal(a0, b0)


def test_chain_knob_and_limits():
    from cpg import ops
    assert ops.chain_bwd_covers(25, 2048, 512)
    assert not ops.chain_bwd_covers(25, 2048, 102)      # 16-byte row layout needs H % 4 == 0
    assert not ops.chain_bwd_covers(25, 8192, 512)      # 4096 workgroups are not co-resident
    assert not ops.chain_bwd_fits(25, 2048, 512)        # policy: off by default
    os.environ["CPG_GRU_BWD_CHAIN"] = "1"
    try:
        assert ops.chain_bwd_fits(25, 2048, 512) and not ops.chain_bwd_fits(25, 8192, 512)
    finally:
        os.environ.pop("CPG_GRU_BWD_CHAIN")


def test_chain_backward_bf16_mode_matches_per_step_bf16():
    """bf16 compute mode: the chain runs the W_hh^T / 64x32 one-plane kernel of the per-step launches - identical results."""
    from cpg import ops
    from cpg.ops import _p, _stream, call
    B, H, T = 512, 256, 9
    dev = torch.device("cuda")
    d = _inputs(B, H, T, 24, seed=31)
    ops.set_compute_mode("bf16")
    try:
        dG_c, dh0_c, (hs, gates, dhs, last) = _run_chain(d, B, H, T, False, True, seed=3, dh_last=True)
        dG_s, dh0_s = torch.zeros(T, B, 4 * H, device=dev), torch.zeros(B, H, device=dev)
        scr, wT = torch.empty(2, B, H, device=dev), torch.empty(H, 3 * H, device=dev)
        call("cpg_gru_seq_bwd", T, B, H, 0, _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(dG_s), _p(scr), _p(dh0_s),
             0, B, None, _p(wT), _stream())
        torch.cuda.synchronize()
    finally:
        ops.set_compute_mode("f32")
    assert torch.equal(dG_c, dG_s) and torch.equal(dh0_c, dh0_s)
    dG_f, _, _ = _run_chain(d, B, H, T, False, True, seed=3, dh_last=True)      # f32-grade result: bf16 mode is close, not equal
    assert not torch.equal(dG_f, dG_c)
    assert (dG_f - dG_c).abs().max().item() < 2e-2 * dG_f.abs().max().item()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,T", [(256, 128, 6), (128, 192, 4)])
def test_backward_two_k_halves_workgroup_matches_the_plain_direct_to_lds_kernel(B, H, T, mode):
    """gru_step_bwd_dl2_kernel (512 threads, two K-halves, partial blocks swapped through LDS; CPG_GRU_BWD_DL2=1 forces it) against
    gru_step_bwd_dl_kernel (=0): the same products with one extra reassociation (half sums added once), forward and reverse."""
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    d = _inputs(B, H, T, 24, seed=41)
    ops.set_compute_mode(mode)
    saved = os.environ.get("CPG_GRU_BWD_DL2")
    try:
        for reverse in (0, 1):
            hs, gates = _run(d, B, H, T, bool(reverse), False)
            g = torch.Generator().manual_seed(7)
            dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
            last = (torch.randn(B, H, generator=g) * 0.1).to(dev)
            res = []
            for knob in ("0", "1"):
                os.environ["CPG_GRU_BWD_DL2"] = knob
                dG, dh0 = torch.zeros(T, B, 4 * H, device=dev), torch.zeros(B, H, device=dev)
                scr, wT = torch.empty(2, B, H, device=dev), torch.empty(H, 3 * H, device=dev)
                call("cpg_gru_seq_bwd", T, B, H, reverse, _p(d["w_hh"]), _p(hs), _p(gates), _p(dhs), _p(last), _p(dG), _p(scr), _p(dh0),
                     0, B, None, _p(wT), _stream())
                torch.cuda.synchronize()
                res.append((dG, dh0))
            # bf16 mode: a last-bit difference in dG can flip the bf16 rounding of the next step's operand (2^-9 of one term)
            tol = 2e-6 if mode == "f32" else 1e-3
            for a, b in zip(res[0], res[1]):
                assert not torch.equal(a, torch.zeros_like(a))
                assert (a - b).abs().max().item() <= tol * max(1.0, a.abs().max().item())
    finally:
        ops.set_compute_mode("f32")
        if saved is None:
            os.environ.pop("CPG_GRU_BWD_DL2", None)
        else:
            os.environ["CPG_GRU_BWD_DL2"] = saved
