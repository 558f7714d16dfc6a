"""Whole-loop decode kernels (cpg_decode_greedy_fused / cpg_decode_beam_fused) against the numpy oracle and against the
per-step launch chain, on seeded random decoders of several widths: ragged batch sizes (partial tiles, sentences that
straddle tiles), rows that finish early (<eos> biased up), min_length, beam widths other than 5.  Token ids and
hypotheses must match exactly; beam scores within 1e-5."""
import numpy as np
import pytest
import torch

from helpers import cu

pytestmark = pytest.mark.gpu
V, T, E = 24, 25, 150


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


def _model(z_dim, seed, eos_bias=0.0):
    from models.model import RNN_VAE
    torch.manual_seed(seed)
    m = RNN_VAE(n_vocab=V, max_seq_len=T, z_dim=z_dim, c_dim=2, emb_dim=E, pretrained_emb=None, freeze_embeddings=False,
                flow=0, flow_type='', E_args=dict(h_dim=16, biGRU=True, layers=1, p_dropout=0.0),
                G_args=dict(G_class='gru', GRU_args=dict(p_word_dropout=0.3, p_out_dropout=0.3, skip_connetions=False),
                            deconv_args=dict()),
                C_args=dict(min_filter_width=3, max_filter_width=5, num_filters=100, dropout=0.5))
    with torch.no_grad():
        # default init gives near-uniform logits; widen them so argmax margins are far above f32 summation noise
        m.decoder.fc[1].weight.mul_(6.0)
        m.decoder.fc[1].bias[3] += eos_bias
    m = m.cuda()
    m.device = torch.device("cuda")
    P = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    return m, P


def _zc(n, z_dim, seed):
    rs = np.random.RandomState(seed)
    z = rs.randn(n, z_dim).astype(np.float32)
    c = np.zeros((n, 2), np.float32)
    c[np.arange(n), rs.randint(0, 2, n)] = 1
    return z, c


@pytest.mark.parametrize("z_dim,n,eos_bias,min_length", [(100, 130, 1.5, 1), (100, 1, 0.0, 1), (16, 63, 2.0, 1),
                                                         (40, 65, 1.0, 4), (68, 200, 1.5, 1), (90, 64, 3.0, 1)])
def test_fused_greedy_matches_oracle(z_dim, n, eos_bias, min_length):
    from oracle import decode as odecode
    from cpg import decode as cdecode
    m, P = _model(z_dim, 7 + z_dim, eos_bias)
    assert cdecode.fused_greedy_fits(z_dim + 2, V, V)
    z, c = _zc(n, z_dim, n)
    ref = odecode.greedy(P, z, c, T, min_length=min_length)
    ids, _, _ = m.generate_sentences(n, cu(z), cu(c), sample_mode='greedy', min_length=min_length)
    assert ids.dtype == torch.int64
    assert np.array_equal(ids.cpu().numpy(), ref)
    if eos_bias >= 1.5 and n > 1:
        assert (ref == 3).any()  # the case really exercises finished rows


def test_fused_greedy_equals_step_path_large():
    from cpg import decode as cdecode
    m, _ = _model(100, 3, 1.0)
    z, c = _zc(5000, 100, 11)
    a, _, _ = m.generate_sentences(5000, cu(z), cu(c), sample_mode='greedy')
    cdecode.FUSED_GREEDY = False
    try:
        b, _, _ = m.generate_sentences(5000, cu(z), cu(c), sample_mode='greedy')
    finally:
        cdecode.FUSED_GREEDY = True
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("z_dim,n,K,n_best,eos_bias,min_length", [(100, 13, 5, 3, 1.0, 1), (100, 25, 3, 2, 2.0, 1),
                                                                  (16, 9, 5, 3, 0.5, 5), (68, 17, 8, 4, 1.5, 1),
                                                                  (40, 7, 2, 1, 2.5, 1)])
def test_fused_beam_matches_oracle(z_dim, n, K, n_best, eos_bias, min_length):
    from oracle import decode as odecode
    from cpg import decode as cdecode
    m, P = _model(z_dim, 21 + z_dim, eos_bias)
    assert cdecode.fused_beam_fits(z_dim + 2, V, V, K)
    z, c = _zc(n, z_dim, 5 * n)
    ref_h, ref_s = odecode.beam(P, z, c, T, K, n_best, min_length)
    m.eval()
    hyps, lens, sc = cdecode.decode_beam_arrays(m.decoder, cu(z), cu(c), T, K, n_best, min_length)
    for i in range(n):
        for j in range(n_best):
            assert hyps[i, j, :lens[i, j]].tolist() == [int(t) for t in ref_h[i][j]], (i, j)
            assert abs(sc[i, j] - ref_s[i][j]) < 1e-5 * max(1.0, abs(ref_s[i][j]))
    out, _, _ = m.generate_sentences(n, cu(z), cu(c), sample_mode='beam', beam_size=K, n_best=n_best, min_length=min_length)
    assert out[0][0] == [int(t) for t in ref_h[0][0]]


def test_fused_beam_equals_step_path_large():
    from cpg import decode as cdecode
    m, _ = _model(100, 5, 1.5)
    m.eval()
    z, c = _zc(3001, 100, 13)
    a = cdecode.decode_beam_arrays(m.decoder, cu(z), cu(c), T)
    cdecode.FUSED_GREEDY = False
    try:
        b = cdecode.decode_beam_arrays(m.decoder, cu(z), cu(c), T)
    finally:
        cdecode.FUSED_GREEDY = True
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], b[0])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-5, atol=1e-5)


def test_fused_limits_reported():
    """Shapes beyond the fused kernels fall back to the per-step chain (and the C entry refuses them loudly)."""
    from cpg import decode as cdecode, ops
    assert not cdecode.fused_greedy_fits(512, V, V)
    assert not cdecode.fused_greedy_fits(126, V, V)   # W_hh fits the registers but the tile state exceeds 160 KB of LDS
    assert not cdecode.fused_beam_fits(102, V, V, 9)
    assert ops.query("cpg_decode_greedy_fused_lds_bytes", 102, V, V) <= 160 * 1024
    d = torch.device("cuda")
    f = lambda *s: torch.zeros(*s, device=d)
    ids = torch.zeros(4, T + 1, device=d, dtype=torch.int64)
    unf = torch.zeros(T, device=d, dtype=torch.int32)
    with pytest.raises(ops.CpgError):
        ops.call("cpg_decode_greedy_fused", ops._p(f(4, 126)), ops._p(f(4, 378)), ops._p(f(V, 378)), V, ops._p(f(378, 126)),
                 ops._p(f(378)), ops._p(f(V, 126)), ops._p(f(V)), 4, 126, V, T, 2, 1, 3, ops._p(ids), T + 1, ops._p(unf),
                 ops._stream())


@pytest.mark.parametrize("Tn,B,H", [(7, 96, 20), (7, 96, 64), (25, 256, 128), (3, 32, 192), (9, 384, 64), (1, 128, 64)])
def test_dgi_reduce_one_pass_sums(Tn, B, H):
    """cpg_gru_dgi_reduce: token-grouped sums, column sums and sums over time of dG vs float64 sums - through the one-hot
    product + over-time pass (H % 64 != 0), the fused single pass (dgi_fused_kernel: H % 64 == 0, B % 32 == 0) and the
    matrix-core single pass (dgi_mfma_kernel: H % 64 == 0, B % 128 == 0)."""
    from cpg import ops
    rs = np.random.RandomState(0)
    dG = rs.randn(Tn, B, 4 * H).astype(np.float32)
    tok = rs.randint(0, V, size=(Tn, B)).astype(np.int32)
    d = torch.device("cuda")
    dtab = torch.empty(V, 3 * H, device=d)
    dsum = torch.empty(4 * H, device=d)
    drowc = torch.empty(B, 3 * H, device=d)
    ws = ops.workspace(ops.query("cpg_gru_wgrad_workspace", Tn, B, H, V), d)
    dG_d, tok_d = cu(dG), cu(tok)  # keep the device tensors alive across the asynchronous call
    ops.call("cpg_gru_dgi_reduce", Tn, B, H, ops._p(dG_d), ops._p(tok_d), V, ops._p(dtab), ops._p(dsum), ops._p(drowc), 0,
             ops._p(ws), ws.numel(), 0, ops._stream())
    flat = dG.reshape(-1, 4 * H).astype(np.float64)
    dgi = np.concatenate([flat[:, :2 * H], flat[:, 3 * H:]], 1)
    ref_tab = np.zeros((V, 3 * H))
    np.add.at(ref_tab, tok.reshape(-1), dgi)
    np.testing.assert_allclose(dtab.cpu().numpy(), ref_tab, atol=1e-4)
    np.testing.assert_allclose(dsum.cpu().numpy(), flat.sum(0), atol=2e-4)
    np.testing.assert_allclose(drowc.cpu().numpy(), dgi.reshape(Tn, B, 3 * H).sum(0), atol=1e-4)


@pytest.mark.parametrize("lstm", [0, 1])
def test_dgi_reduce_matrix_core_pass_accumulates(lstm):
    """dgi_mfma_kernel with accumulate = 1, with and without the over-time sums, GRU and LSTM gate layouts; ids >= V (none in
    the reference's data) count in the column sums only."""
    from cpg import ops
    Tn, B, H = 6, 256, 64
    NC = 4 * H if lstm else 3 * H
    rs = np.random.RandomState(3)
    dG = rs.randn(Tn, B, 4 * H).astype(np.float32)
    tok = rs.randint(0, V + 2, size=(Tn, B)).astype(np.int32)
    d = torch.device("cuda")
    flat = dG.reshape(-1, 4 * H).astype(np.float64)
    dgi = flat if lstm else np.concatenate([flat[:, :2 * H], flat[:, 3 * H:]], 1)
    ref_tab = np.zeros((V + 2, NC))
    np.add.at(ref_tab, tok.reshape(-1), dgi)
    name = "cpg_lstm_dgi_reduce" if lstm else "cpg_gru_dgi_reduce"
    ws = ops.workspace(ops.query("cpg_gru_wgrad_workspace", Tn, B, H, V), d)
    dG_d, tok_d = cu(dG), cu(tok)
    for with_rowc in (True, False):
        dtab, dsum, drowc = torch.ones(V, NC, device=d), torch.full((4 * H,), 2.0, device=d), torch.full((B, NC), 3.0, device=d)
        ops.call(name, Tn, B, H, ops._p(dG_d), ops._p(tok_d), V, ops._p(dtab), ops._p(dsum), ops._p(drowc) if with_rowc else None, 1,
                 ops._p(ws), ws.numel(), *(() if lstm else (0,)), ops._stream())
        np.testing.assert_allclose(dtab.cpu().numpy(), 1.0 + ref_tab[:V], atol=1e-4)
        np.testing.assert_allclose(dsum.cpu().numpy(), 2.0 + flat.sum(0), atol=2e-4)
        if with_rowc:
            np.testing.assert_allclose(drowc.cpu().numpy(), 3.0 + dgi.reshape(Tn, B, NC).sum(0), atol=1e-4)
