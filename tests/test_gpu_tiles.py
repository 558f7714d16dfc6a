"""Every compiled tile instantiation of the recurrent kernels against the reference's vectors, and the instantiations the
launcher picks at BASELINE.json's sizes against the oracle.

The golden batches are small (B <= 16), so left alone the launcher only ever picks its small-problem tiles for them; the
benchmark shapes (B=2048, H=512 / H=1024, T=50) select different template instantiations.  Part 1 forces each instantiation
through the launch-time knobs (CPG_GRU_FWD_BM, CPG_GRU_BWD_BM, CPG_GRU_BWD_WIDE, CPG_TN_TILE) and repeats the golden
checks; part 2 runs the real sizes - where the launcher itself chooses - against the numpy oracle (oracle/wae.py,
oracle/decode.py) on seeded inputs.  Bars: losses 1e-4, gradients 2e-6 + 1e-4 max|g|, greedy ids bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from helpers import (check_decoder_teacher_forced_golden, check_encoder_golden, check_losses_and_grads_golden,
                     check_train_trajectory_golden, cu)

pytestmark = pytest.mark.gpu
MODELS = ["A", "micro", "enc2"]
KNOBS = ("CPG_GRU_FWD_BM", "CPG_GRU_BWD_BM", "CPG_GRU_BWD_WIDE", "CPG_GRU_BWD_TILE", "CPG_TN_TILE", "CPG_TN_SPLIT",
         "CPG_GRU_BWD_CHAIN", "CPG_GRU_BWD_STAGGER")


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


@pytest.fixture
def knobs():
    """Set launch-time tile knobs (read with getenv by the launchers at every call) and restore them afterwards."""
    saved = {k: os.environ.get(k) for k in KNOBS}

    def set_(**kw):
        for k, v in kw.items():
            assert k in KNOBS, k
            os.environ[k] = str(v)
    yield set_
    torch.cuda.synchronize()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


# ------------------------------------------------------------------------------------------------ part 1: forced tiles
@pytest.mark.parametrize("bm", [32, 64, 128])
@pytest.mark.parametrize("name", MODELS)
def test_forward_tiles_golden(golden, knobs, name, bm):
    """gru_step_fwd_kernel<GF32|GF64|GF128> (GF64 is the bench's forward instantiation)."""
    knobs(CPG_GRU_FWD_BM=bm)
    g = golden("model_" + name)
    check_encoder_golden(g)
    check_decoder_teacher_forced_golden(g)


BWD_VARIANTS = [dict(CPG_GRU_BWD_BM=32), dict(CPG_GRU_BWD_BM=64), dict(CPG_GRU_BWD_BM=128),
                dict(CPG_GRU_BWD_WIDE=32), dict(CPG_GRU_BWD_WIDE=64), dict(CPG_GRU_BWD_WIDE=128)] + \
               [dict(CPG_GRU_BWD_TILE=t) for t in ("64x32", "32x64", "64x64", "128x32", "128x64", "32x32")] + \
               [dict(CPG_GRU_BWD_WIDE=3264), dict(CPG_GRU_BWD_STAGGER=0), dict(CPG_GRU_BWD_STAGGER=2), dict(CPG_GRU_BWD_CHAIN=1)]


@pytest.mark.parametrize("variant", BWD_VARIANTS, ids=lambda v: "-".join(f"{k[12:]}{x}" for k, x in v.items()))
@pytest.mark.parametrize("name", MODELS)
def test_backward_tiles_golden(golden, knobs, name, variant):
    """gru_step_bwd_kernel<GB32|GB64|GB128|GB32N|GB64W|GB128W> on the exact-f32 path (BM / WIDE knobs; GB32N = 32x32 tiles
    is the bench's) and on the split-bf16 path with W_hh^T handed over (TILE knob); 64-deep slabs (WIDE=3264); the staggered
    epilogue-operand fetch at other spacings; the one-launch BPTT (CHAIN=1: whole model through cpg_gru_*_bwd_chain)."""
    knobs(**variant)
    check_losses_and_grads_golden(golden("model_" + name))


@pytest.mark.parametrize("tile", ["256x128", "128x64", "64x64", "128x128", "128x32", "32x128"])
@pytest.mark.parametrize("split", [1, 3])
@pytest.mark.parametrize("name", MODELS)
def test_wgrad_tiles_golden(golden, knobs, name, tile, split):
    """dW = dY^T X products (the dW_hh product and every nn.Linear weight gradient): each tile shape, with and without
    split-K (256x128 split-K, 512-thread workgroups, is the bench's dW_hh instantiation)."""
    knobs(CPG_TN_TILE=tile, CPG_TN_SPLIT=split)
    check_losses_and_grads_golden(golden("model_" + name))


@pytest.mark.parametrize("variant", [dict(CPG_GRU_FWD_BM=64, CPG_GRU_BWD_WIDE=32, CPG_TN_TILE="128x64", CPG_TN_SPLIT=4),
                                     dict(CPG_GRU_FWD_BM=128, CPG_GRU_BWD_BM=128, CPG_TN_TILE="128x128")],
                         ids=["bench-tiles", "large-tiles"])
@pytest.mark.parametrize("name", ["micro_clip", "A_clip"])
def test_train_trajectory_forced_tiles(golden, knobs, name, variant):
    knobs(**variant)
    check_train_trajectory_golden(golden("train_" + name))


# ------------------------------------------------------------------------------------------------ part 2: real sizes
def _random_case(B, T, V, Z, He, enc_layers, seed):
    from bench import model_kwargs
    from cpg.synth import synth_ids
    from models.model import RNN_VAE
    torch.manual_seed(seed)
    m = RNN_VAE(n_vocab=V, max_seq_len=T, **model_kwargs(Z, He, enc_layers=enc_layers))
    P = {k: v.detach().numpy().copy() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    rs = np.random.RandomState(seed)
    ids = synth_ids(B, T, V, torch.Generator().manual_seed(seed)).numpy()
    c = np.zeros((B, 2), np.float32)
    c[np.arange(B), rs.randint(0, 2, B)] = 1
    rnd = dict(eps=rs.randn(B, Z).astype(np.float32), c=c, wd_mask=(rs.rand(B, T) < 0.3).astype(np.uint8),
               out_mask=(rs.rand(B, T, Z + 2) >= 0.3).astype(np.uint8), z_prior_full=rs.randn(B, Z).astype(np.float32),
               z_prior_rf=rs.randn(B, Z).astype(np.float32), rf_w=rs.randn(Z, 500).astype(np.float32),
               rf_b=(2 * np.pi * rs.rand(500)).astype(np.float32))
    m = m.cuda()
    m.device = torch.device("cuda")
    return m, P, ids, rnd


def _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf", beta=1.5, lam_l1=0.1, lam_kl=1e-3):
    import losses
    from helpers import set_losses_cfg
    from oracle import wae
    set_losses_cfg()
    terms, G, aux = wae.train_loss_and_grads(P, ids, rnd, beta, lam_l1, lam_kl, regu)
    losses.rf.clear()
    losses.rf['gaussian'] = (cu(rnd["rf_w"]), cu(rnd["rf_b"]))
    idt = cu(ids)
    rc = dict(eps=cu(rnd["eps"]), c=cu(rnd["c"]), wd_mask=cu(rnd["wd_mask"]), out_mask=cu(rnd["out_mask"]))
    (mu, lv), (z, c), logits = m(idt, q_c='prior', sample_z=1, rnd=rc)
    recon = losses.recon_dec(idt, logits)
    kl = losses.kl_gaussianprior(mu, lv)
    mmd = losses.wae_mmd_gaussianprior(z, method='full_kernel', z_prior=cu(rnd["z_prior_full"]))
    mmdrf = losses.wae_mmd_gaussianprior(z, method='rf', z_prior=cu(rnd["z_prior_rf"]))
    l1 = losses.logvar_l1(lv)
    klmu = losses.kl_gaussian_sharedmu(mu, lv)
    loss = recon + beta * {'kl': kl, 'mmd': mmd, 'mmdrf': mmdrf}[regu] + lam_l1 * l1 + lam_kl * klmu
    z.retain_grad()
    loss.backward()
    torch.cuda.synchronize()
    for name, got in (("recon", recon), ("kl", kl), ("mmd", mmd), ("mmdrf", mmdrf), ("klmu", klmu), ("total", loss)):
        ref = float(terms[name])
        assert abs(got.item() - ref) < 1e-4 * max(1.0, abs(ref)), (name, got.item(), ref)
    assert abs(l1.item() - float(terms["l1"])) < 1e-4 * max(1.0, abs(float(terms["l1"])))
    np.testing.assert_allclose(mu.detach().cpu().numpy(), aux["mu"], atol=2e-5)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), aux["logits"], atol=1e-4)
    ref = aux["dz"]
    np.testing.assert_allclose(z.grad.cpu().numpy(), ref, atol=2e-6 + 1e-4 * np.abs(ref).max(), rtol=0)
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        ref, got = G[k], prm.grad.cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=2e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


def _check_greedy_vs_oracle(m, P, N, T, seed):
    """Token ids bit-exact against the oracle; a row may only differ from the step on at which the ORACLE's own top-2 logit
    margin is below 1e-5 (an f32 tie: either argmax is a correct evaluation) - and such rows must be rare."""
    from oracle import decode
    rs = np.random.RandomState(seed)
    Z = m.z_dim
    z = rs.randn(N, Z).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    ref, ref_logits = decode.greedy(P, z, c, T, return_logits=True)
    ids, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='greedy')
    got = ids.cpu().numpy()
    assert got.shape[1] <= T + 1 and ref.shape[1] <= T + 1
    w = min(got.shape[1], ref.shape[1])
    ties = 0
    for i in np.nonzero((got[:, :w] != ref[:, :w]).any(1))[0]:
        s = int(np.nonzero(got[i, :w] != ref[i, :w])[0][0]) - 1    # decode step that produced the first differing column
        top2 = np.sort(ref_logits[i, s])[-2:]
        assert top2[1] - top2[0] < 1e-5, (i, s, got[i], ref[i], top2)
        ties += 1
    assert ties <= max(1, N // 256), ties
    if ties == 0:
        assert np.array_equal(got, ref)


def test_config_b_step_vs_oracle():
    """BASELINE.json configs[1] dimensions (biGRU encoder h=512, z=510, decoder h=512, T=25, B=2048): the launcher itself
    picks the bench's instantiations (64-row split-bf16 forward tiles, exact-f32 32x32 backward tiles, 128x64 split-K dW)."""
    m, P, ids, rnd = _random_case(2048, 25, 24, 510, 512, 1, seed=11)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf")
    _check_greedy_vs_oracle(m, P, 512, 25, seed=12)


def test_config_b_step_one_launch_bptt_vs_oracle(knobs):
    """The same step with both backward recurrences (decoder, encoder pair) as one-launch chains (CPG_GRU_BWD_CHAIN=1)."""
    knobs(CPG_GRU_BWD_CHAIN=1)
    m, P, ids, rnd = _random_case(1024, 25, 24, 510, 512, 1, seed=21)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf")
    from cpg import ops
    ops.check_persistent()


def test_config_b_full_mmd_regulariser_vs_oracle():
    """Same dimensions at B=512 with the full-kernel MMD as the regulariser (its gradient then drives dz)."""
    m, P, ids, rnd = _random_case(512, 25, 24, 510, 512, 1, seed=13)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmd", beta=2.0, lam_l1=0.0)


def test_config_c_step_vs_oracle():
    """BASELINE.json configs[4] dimensions as far as the reference defines them: 2-layer biGRU encoder h=1024
    (models/encoder.py:27,46-47), z=1022, decoder h=1024 (1 layer: models/model.py:283-284), T=50; B=256."""
    m, P, ids, rnd = _random_case(256, 50, 24, 1022, 1024, 2, seed=17)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf")
    _check_greedy_vs_oracle(m, P, 128, 50, seed=18)


def test_other_seq_len_vs_oracle():
    """T != 25 at reference-default widths (He=80, Z=100): T=50 and T=7, B=192 (partial row tiles)."""
    for T, seed in ((50, 21), (7, 22)):
        m, P, ids, rnd = _random_case(192, T, 24, 100, 80, 1, seed=seed)
        _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf")
        _check_greedy_vs_oracle(m, P, 256, T, seed=seed + 100)


def test_minimal_seq_len_vs_oracle(monkeypatch):
    """The shortest sequences the path can be given: T = 1 (only <start>: every target is PAD), 2, 3, with rows of every live
    length, at B = 5 (one partial row tile) and B = 64 - losses and gradients against the oracle."""
    import cpg.synth as synth

    def tiny_ids(B, T, V, gen):   # <start>=2, <eos>=3, <pad>=1, residues from 4
        ids = torch.full((B, T), 1, dtype=torch.int64)
        ids[:, 0] = 2
        for b in range(B):
            n = 1 + (b % T)
            for j in range(1, n):
                ids[b, j] = 4 + (7 * b + j) % (V - 4)
            if n < T:
                ids[b, n] = 3
        return ids
    monkeypatch.setattr(synth, "synth_ids", tiny_ids)
    for T in (1, 2, 3):
        for B in (5, 64):
            m, P, ids, rnd = _random_case(B, T, 24, 100, 80, 1, seed=30 + T)
            _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf")
