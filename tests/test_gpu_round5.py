"""Round-5 parity additions: every shape bench.py times is compared at the size it is timed at.

  * the multi-layer decoder EXTENSION (BASELINE.json configs[4]: "2-layer biLSTM enc + 2-layer dec, hidden=1024, seq_len<=50") -
    model-level training step against oracle/torch_ref.RefWAE(dec_layers=2) (torch.nn.GRU / nn.LSTM(num_layers=2) + autograd on the
    CPU), greedy / beam / soft decodes and forward_sample against oracle/decode.py's multi-layer restatement (itself pinned to
    torch.nn.GRU / nn.LSTM in tests/test_oracle_golden.py).  The reference's decoder is hard-wired to one layer
    (models/decoder.py:40-41, models/model.py:283-284): PARITY UNPINNED against the reference for everything in this group;
  * the LSTM extension in the bf16 mode at configs[1] size (B = 2048, h = 512) and a bf16 deviation record at configs[4] size;
  * beam search through the per-step chain with hypotheses of 2 ... >= 15 tokens at config-B / config-C width;
  * one CLaSS round of 10^6 proposals: rows of the big round equal the same rows drawn / scored / decoded alone, accept mask and
    LR probabilities equal the oracle's on a 10 k slice, decoded residues equal the oracle's on a small slice.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import build_model, cu, set_losses_cfg
from conftest import weights_of

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")


def _write_report(name, rows):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(rows, open(os.path.join(d, name), "w"), indent=1)
    except OSError:
        pass


def _case(B, T, V, Z, He, enc_layers, dec_layers, cell, seed):
    from bench import model_kwargs
    from cpg.synth import synth_ids
    from models.model import RNN_VAE
    torch.manual_seed(seed)
    m = RNN_VAE(n_vocab=V, max_seq_len=T, **model_kwargs(Z, He, enc_layers=enc_layers, cell=cell, dec_layers=dec_layers))
    P = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    rs = np.random.RandomState(seed)
    ids = synth_ids(B, T, V, torch.Generator().manual_seed(seed))
    c = np.zeros((B, 2), np.float32)
    c[np.arange(B), rs.randint(0, 2, B)] = 1
    rnd = dict(eps=rs.randn(B, Z).astype(np.float32), c=c, wd_mask=(rs.rand(B, T) < 0.3).astype(np.uint8),
               out_mask=(rs.rand(B, T, Z + 2) >= 0.3).astype(np.uint8), z_prior_rf=rs.randn(B, Z).astype(np.float32),
               rf_w=rs.randn(Z, 500).astype(np.float32), rf_b=(2 * np.pi * rs.rand(500)).astype(np.float32))
    return m, P, ids, rnd


def _ref_step(P, ids, rnd, cell, beta, lam_l1, lam_kl):
    from oracle import torch_ref
    ref = torch_ref.RefWAE.from_state(P, cell=cell)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    terms, aux = torch_ref.train_loss(ref, ids, {k: torch.from_numpy(v) for k, v in rnd.items()}, beta, lam_l1, lam_kl, "mmdrf",
                                      full_mmd=False)
    terms["total"].backward()
    G = {ref.ref_name(k): p.grad.numpy() for k, p in ref.named_parameters()}
    return terms, aux, G


def _hip_step(m, ids, rnd, beta, lam_l1, lam_kl):
    import losses
    m = m.cuda()
    m.device = torch.device("cuda")
    losses.rf.clear()
    losses.rf['gaussian'] = (cu(rnd["rf_w"]), cu(rnd["rf_b"]))
    idt = ids.cuda()
    (mu, lv), (z, cc), logits = m(idt, q_c='prior', sample_z=1,
                                  rnd=dict(eps=cu(rnd["eps"]), c=cu(rnd["c"]), wd_mask=cu(rnd["wd_mask"]), out_mask=cu(rnd["out_mask"])))
    out = dict(recon=losses.recon_dec(idt, logits), mmdrf=losses.wae_mmd_gaussianprior(z, method='rf', z_prior=cu(rnd["z_prior_rf"])),
               l1=losses.logvar_l1(lv), klmu=losses.kl_gaussian_sharedmu(mu, lv), kl=losses.kl_gaussianprior(mu, lv))
    out["total"] = out["recon"] + beta * out["mmdrf"] + lam_l1 * out["l1"] + lam_kl * out["klmu"]
    out["total"].backward()
    torch.cuda.synchronize()
    return m, out, mu, logits


@pytest.mark.parametrize("cell,B,He,Z,T,enc_layers", [("gru", 64, 32, 30, 9, 1), ("lstm", 64, 32, 30, 9, 1),
                                                      ("gru", 256, 1024, 1022, 50, 2), ("lstm", 256, 1024, 1022, 50, 2)],
                         ids=["gru-small", "lstm-small", "gru-configs4", "lstm-configs4"])
def test_two_layer_decoder_model_step_vs_torch_ref(cell, B, He, Z, T, enc_layers):
    """A whole training step with a 2-layer decoder - at BASELINE.json configs[4]'s dimensions AS NAMED (2-layer bidirectional
    encoder h = 1024, 2-layer decoder h = 1024, T = 50; GRU and LSTM cells) and at a small shape - against torch.nn.GRU / nn.LSTM
    (num_layers=2) + autograd on the CPU with every draw injected: loss terms 1e-4, mu 2e-5, logits 1e-4, EVERY parameter gradient
    (incl. decoder.rnn.*_l1) at 2e-6 + 1e-4 max|g|.  Extension: parity unpinned against the reference."""
    set_losses_cfg()
    m, P, ids, rnd = _case(B, T, 24, Z, He, enc_layers, 2, cell, seed=900 + B + (1 if cell == "lstm" else 0))
    assert "decoder.rnn.weight_hh_l1" in P and "decoder.rnn.weight_ih_l1" in P
    beta, lam_l1, lam_kl = 1.5, 0.0, 1e-3
    terms, aux, G = _ref_step(P, ids, rnd, cell, beta, lam_l1, lam_kl)
    m, out, mu, logits = _hip_step(m, ids, rnd, beta, lam_l1, lam_kl)
    for name in ("recon", "kl", "mmdrf", "l1", "klmu", "total"):
        want = float(terms[name].detach())
        assert abs(out[name].item() - want) < 1e-4 * max(1.0, abs(want)), (name, out[name].item(), want)
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


@pytest.mark.parametrize("cell", ["gru", "lstm"])
@pytest.mark.parametrize("Z,He,T", [(46, 32, 25), (510, 64, 25)], ids=["h48", "h512"])
def test_two_layer_decoder_decodes_vs_oracle(cell, Z, He, T):
    """sample_G with a 2-layer decoder (per-step chain: the whole-loop kernels cover one layer): greedy ids BIT-EXACT, beam-5 /
    n-best-3 hypotheses EXACT (every layer's state follows the back-pointers, models/model.py:378-385), greedy_softmax ids exact and
    soft rows 2e-5, one forward_sample call = one step of the oracle (state [layers,N,H]; the LSTM's (h, c) pair)."""
    from bench import model_kwargs
    from models.model import RNN_VAE
    from models.mutils import EOS_IDX, START_IDX
    from oracle import decode as odec
    dev = torch.device("cuda")
    torch.manual_seed(31 + Z)
    m = RNN_VAE(n_vocab=24, max_seq_len=T, **model_kwargs(Z, He, cell=cell, dec_layers=2)).to(dev)
    m.device = dev
    with torch.no_grad():
        m.decoder.fc[1].weight.mul_(4.0)
        m.decoder.fc[1].bias[EOS_IDX] += 0.8
    P = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    assert odec.n_dec_layers(P) == 2
    rs = np.random.RandomState(Z)
    N = 48
    z = rs.randn(N, Z).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    zt, ct = cu(z), cu(c)
    ids, _, _ = m.generate_sentences(N, zt, ct, sample_mode='greedy')
    ref = odec.greedy(P, z, c, T, cell=cell)
    assert np.array_equal(ids.cpu().numpy(), ref)
    got, _, _ = m.generate_sentences(N, zt, ct, sample_mode='beam', beam_size=5, n_best=3)
    hyps, _, margins = odec.beam(P, z, c, T, beam_size=5, n_best=3, cell=cell, return_margins=True)
    bad = [i for i in range(N) if [list(map(int, h)) for h in got[i]] != hyps[i]]
    assert all(margins[i] < 2e-5 for i in bad) and len(bad) <= 1, (bad, margins[bad] if bad else None)
    (sids, soft), _, _ = m.generate_sentences(N, zt, ct, sample_mode='greedy_softmax', temp=0.8)
    ref_ids, ref_soft = odec.soft_sample(P, z, c, T, 'greedy_softmax', temp=0.8, cell=cell)
    assert np.array_equal(sids.cpu().numpy(), ref_ids)
    np.testing.assert_allclose(soft.cpu().numpy(), ref_soft, atol=2e-5)
    # forward_sample: one step from the initial state
    m.eval()
    zc = np.concatenate([z, c], 1)
    h0 = m.decoder.init_hidden(zt, ct).unsqueeze(0).repeat(2, 1, 1)
    tok = torch.full((N,), START_IDX, device=dev, dtype=torch.long)
    if cell == "lstm":
        logits, (h1, c1) = m.decoder.forward_sample(None, tok, zt, ct, (h0, torch.zeros_like(h0)))
        rl, rh, rc = odec.lstm_decoder_step(P, np.full(N, START_IDX), zc, odec.init_state(P, zc), np.zeros((2, N, Z + 2), np.float32))
        np.testing.assert_allclose(c1.cpu().numpy(), rc, atol=5e-6)
    else:
        logits, h1 = m.decoder.forward_sample(None, tok, zt, ct, h0)
        rl, rh = odec.decoder_step(P, np.full(N, START_IDX), zc, odec.init_state(P, zc))
    assert tuple(h1.shape) == (2, N, Z + 2)
    np.testing.assert_allclose(logits.cpu().numpy(), rl, atol=2e-5)
    np.testing.assert_allclose(h1.cpu().numpy(), rh, atol=5e-6)
    m.train()


def test_two_layer_decoder_state_dict_and_checkpoint_roundtrip(tmp_path):
    """State-dict keys of the extension are torch.nn.GRU(num_layers=2)'s (decoder.rnn.*_l1), a checkpoint round-trips through
    api.load_trained_model-style loading, and the default (layers=1) key set is unchanged."""
    from bench import model_kwargs
    from models.model import RNN_VAE
    m1 = RNN_VAE(n_vocab=24, max_seq_len=25, **model_kwargs(30, 16))
    m2 = RNN_VAE(n_vocab=24, max_seq_len=25, **model_kwargs(30, 16, dec_layers=2))
    k1, k2 = set(m1.state_dict()), set(m2.state_dict())
    assert k2 - k1 == {f"decoder.rnn.{n}_l1" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")} and k1 <= k2
    assert tuple(m2.state_dict()["decoder.rnn.weight_ih_l1"].shape) == (3 * 32, 32)
    m3 = RNN_VAE(n_vocab=24, max_seq_len=25, **model_kwargs(30, 16, dec_layers=2))
    m3.load_state_dict(m2.state_dict())
    for k, v in m2.state_dict().items():
        assert torch.equal(v, m3.state_dict()[k])


@pytest.mark.parametrize("dec_layers", [1, 2])
def test_configs4_bf16_deviation_record(dec_layers):
    """BASELINE.json configs[4] "fp32-ref vs bf16": the bf16 compute mode against torch.nn.LSTM in f32 on the CPU at configs[4]
    dimensions (2-layer biLSTM encoder h = 1024, LSTM decoder h = 1024 with 1 and 2 layers, T = 50, B = 256).  Agreement bars of the
    mode (tests/test_gpu_bf16.py): loss terms 2e-3 (absolute, relative to max(1, |value|)), worst gradient 3 % relative L2.  The
    deviations are recorded in gpurun_out/r05_configs4_bf16_report.json (copied to profiles/)."""
    from cpg import ops
    set_losses_cfg()
    m, P, ids, rnd = _case(256, 50, 24, 1022, 1024, 2, dec_layers, "lstm", seed=4400 + dec_layers)
    terms, aux, G = _ref_step(P, ids, rnd, "lstm", 1.5, 0.0, 1e-3)
    ops.set_compute_mode('bf16')
    try:
        m, out, mu, logits = _hip_step(m, ids, rnd, 1.5, 0.0, 1e-3)
    finally:
        ops.set_compute_mode('f32')
    rec = dict(config="configs[4]: 2-layer biLSTM enc h=1024, %d-layer LSTM dec h=1024, T=50, B=256" % dec_layers, mode="bf16")
    for name in ("recon", "mmdrf", "klmu", "total"):
        want = float(terms[name].detach())
        rec["abs_dev_" + name] = abs(out[name].item() - want)
        assert rec["abs_dev_" + name] < 2e-3 * max(1.0, abs(want)), (name, out[name].item(), want)
    rec["logits_max_abs_dev"] = float(np.abs(logits.detach().cpu().numpy() - aux["logits"].detach().numpy()).max())
    worst, worst_k = 0.0, None
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        want = G[k].astype(np.float64)
        r = float(np.linalg.norm(prm.grad.cpu().numpy() - want) / max(np.linalg.norm(want), 1e-30))
        if r > worst:
            worst, worst_k = r, k
    rec["worst_gradient_rel_l2"], rec["worst_gradient"] = worst, worst_k
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05_configs4_bf16_report.json")
    rows = json.load(open(path)) if os.path.exists(path) else []
    rows = [r for r in rows if r.get("config") != rec["config"]] + [rec]
    _write_report("r05_configs4_bf16_report.json", rows)
    assert 1e-6 < worst < 3e-2, (worst, worst_k)


def test_lstm_bf16_mode_at_configs1_size():
    """BASELINE.json configs[1] AS NAMED - "hidden=512 1-layer LSTM, batch=2048, seq_len<=25, bf16" - the shape `extra.lstm_bf16` is
    timed on: the bf16 mode's LSTM step against torch.nn.LSTM (f32, CPU) at B = 2048, h = 512, T = 25 (round-4 verdict: that leg was
    only agreement-tested at B = 256, H = 128).  The mode's bars: loss terms 2e-3, worst gradient 3 % relative L2."""
    from cpg import ops
    set_losses_cfg()
    m, P, ids, rnd = _case(2048, 25, 24, 510, 512, 1, 1, "lstm", seed=2048512)
    terms, aux, G = _ref_step(P, ids, rnd, "lstm", 1.5, 0.0, 1e-3)
    ops.set_compute_mode('bf16')
    try:
        m, out, mu, logits = _hip_step(m, ids, rnd, 1.5, 0.0, 1e-3)
    finally:
        ops.set_compute_mode('f32')
    for name in ("recon", "mmdrf", "klmu", "total"):
        want = float(terms[name].detach())
        assert abs(out[name].item() - want) < 2e-3 * max(1.0, abs(want)), (name, out[name].item(), want)
    worst = 0.0
    for k, prm in m.named_parameters():
        if k.startswith("classifier") or k == "decoder.emb.weight":
            continue
        want = G[k].astype(np.float64)
        worst = max(worst, float(np.linalg.norm(prm.grad.cpu().numpy() - want) / max(np.linalg.norm(want), 1e-30)))
    assert 1e-6 < worst < 3e-2, worst


# ------------------------------------------------------------------------------------------------ long beams on the per-step chain
LONG_BEAM_REPORT = []


@pytest.mark.parametrize("Z,T,tag", [(510, 25, "config B width"), (1022, 50, "config C width")])
def test_beam_long_hypotheses_per_step_chain(Z, T, tag):
    """Beam-5 / n-best-3 through the per-step chain (decoder h = 512 / 1024) with hypotheses of MANY lengths (round-4 verdict: the
    eos-biased cases ended every beam after 1-3 steps): runs with different <eos> biases and min_length (models/Beam.py:66-67 masks
    <eos> below it) make the back-pointer walks (Beam.py:119-132), the "EOS-ended beam has no children" rule (:78-80) and the
    hidden-state re-gather (_update_hidden, models/model.py:378-385) run over 2 ... >= 15 steps.  Hypotheses EXACT for these seeds."""
    from bench import model_kwargs
    from models.model import RNN_VAE
    from models.mutils import EOS_IDX
    from oracle import decode as odec
    dev = torch.device("cuda")
    torch.manual_seed(Z)
    m = RNN_VAE(n_vocab=24, max_seq_len=T, **model_kwargs(Z, 32)).to(dev)
    m.device = dev
    P0 = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    b0 = m.decoder.fc[1].bias.detach().clone()
    N = 32
    rs = np.random.RandomState(Z + 1)
    z = rs.randn(N, Z).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    all_lens = []
    for eos_bias, min_length in ((1.5, 1), (0.4, 1), (1.5, 8), (2.5, 15), (0.0, 20 if T >= 25 else 10)):
        with torch.no_grad():
            m.decoder.fc[1].bias.copy_(b0)
            m.decoder.fc[1].bias[EOS_IDX] += eos_bias
        P = dict(P0)
        P["decoder.fc.1.bias"] = P0["decoder.fc.1.bias"].copy()
        P["decoder.fc.1.bias"][EOS_IDX] += np.float32(eos_bias)
        ref, _, margins = odec.beam(P, z, c, T, beam_size=5, n_best=3, min_length=min_length, return_margins=True)
        got, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='beam', beam_size=5, n_best=3, min_length=min_length)
        bad = [i for i in range(N) if [list(map(int, h)) for h in got[i]] != ref[i]]
        lens = [len(h) for s in ref for h in s]
        all_lens += lens
        LONG_BEAM_REPORT.append(dict(test=tag, eos_bias=eos_bias, min_length=min_length, sentences=N, hyp_len_min=int(min(lens)),
                                     hyp_len_max=int(max(lens)), differing=len(bad), smallest_margin=float(margins.min())))
        _write_report("r05_long_beam_report.json", LONG_BEAM_REPORT)
        assert not bad, (tag, eos_bias, min_length, bad, [margins[i] for i in bad])
    assert min(all_lens) <= 3 and max(all_lens) >= 16, (min(all_lens), max(all_lens))
    assert len(set(all_lens)) >= 8, sorted(set(all_lens))


@pytest.mark.parametrize("K,n_best", [(15, 3), (9, 9), (24, 5)])
def test_beam_wider_than_eight_vs_oracle(golden, K, n_best):
    """Beam widths above 8 (the reference's static_eval.py:130,152 decodes with beam_size = 15): the per-step chain's
    beam_select_kernel<., 32> / beam_hyp_kernel<32> against oracle.decode.beam on the golden model trained for 200 reference
    iterations - hypotheses exact, scores 1e-4."""
    from oracle import decode as odec
    g = golden("model_A_200")
    P = weights_of(g)
    m = build_model(P)
    m.eval()
    rs = np.random.RandomState(K)
    N = 24
    z = rs.randn(N, g["greedy_z"].shape[1]).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    ref, ref_sc, margins = odec.beam(P, z, c, 25, beam_size=K, n_best=n_best, return_margins=True)
    got, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='beam', beam_size=K, n_best=n_best)
    bad = [i for i in range(N) if [list(map(int, h)) for h in got[i]] != ref[i]]
    assert all(margins[i] < 2e-5 for i in bad) and len(bad) <= 1, (bad, [margins[i] for i in bad])
    from cpg import decode as cdecode
    m.eval()      # generate_sentences returned the model to TRAIN mode (models/model.py:221-222): out-dropout would be live
    _, _, sc = cdecode.decode_beam_arrays(m.decoder, cu(z), cu(c), 25, K, n_best)
    ok = [i for i in range(N) if i not in bad]
    np.testing.assert_allclose(sc[ok], np.asarray(ref_sc, np.float32)[ok], atol=1e-4)
    with pytest.raises(Exception):
        m.generate_sentences(N, cu(z), cu(c), sample_mode='beam', beam_size=33, n_best=3)


@pytest.mark.parametrize("Z,T,N,NB", [(510, 25, 2048, 256), (1022, 50, 1024, 256)], ids=["config-B-width", "config-C-width"])
def test_plane_step_decode_chain_vs_oracle(Z, T, N, NB):
    """The GRU decode step on plane images (cpg_gru_step_fwd_planes, csrc/planes.hip: the per-step chain of sample_G for decoders too wide
    for the whole-loop kernels, taken from 1024 rows up - CLaSS at config-B / C width) against the oracle: greedy ids of N z BIT-EXACT,
    beam-5 / n-best-3 hypotheses of NB z (5 NB = 1280 rows through the plane step, images re-gathered by back-pointer) EXACT, with
    hypotheses of several lengths.  models/model.py:295-363, models/decoder.py:86-99."""
    from bench import model_kwargs
    from cpg import decode as cdecode
    from models.model import RNN_VAE
    from models.mutils import EOS_IDX
    from oracle import decode as odec
    dev = torch.device("cuda")
    torch.manual_seed(Z + 3)
    m = RNN_VAE(n_vocab=24, max_seq_len=T, **model_kwargs(Z, 32)).to(dev)
    m.device = dev
    with torch.no_grad():
        m.decoder.fc[1].bias[EOS_IDX] += 0.5
    P = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if not k.startswith("classifier")}
    assert cdecode.PlaneStep(m.decoder, N, Z + 2, False).ok and cdecode.PlaneStep(m.decoder, 5 * NB, Z + 2, False).ok
    rs = np.random.RandomState(Z)
    z = rs.randn(N, Z).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1
    ids, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='greedy')
    ref = odec.greedy(P, z, c, T)
    assert np.array_equal(ids.cpu().numpy(), ref)
    got, _, _ = m.generate_sentences(NB, cu(z[:NB]), cu(c[:NB]), sample_mode='beam', beam_size=5, n_best=3)
    hyps, _, margins = odec.beam(P, z[:NB], c[:NB], T, beam_size=5, n_best=3, return_margins=True)
    bad = [i for i in range(NB) if [list(map(int, h)) for h in got[i]] != hyps[i]]
    assert not bad, (bad, [margins[i] for i in bad])
    assert len({len(h) for s_ in hyps for h in s_}) >= 4
    # and the same decode with the form switched off (the round-4 step kernel) gives the same ids
    import os
    os.environ["CPG_NO_STEP_PLANES"] = "1"
    try:
        ids2, _, _ = m.generate_sentences(N, cu(z), cu(c), sample_mode='greedy')
    finally:
        del os.environ["CPG_NO_STEP_PLANES"]
    assert torch.equal(ids, ids2)


# ------------------------------------------------------------------------------------------------ CLaSS at 10^6 rows
def _clf(coef, icpt):
    from sklearn.linear_model import LogisticRegression
    clf = LogisticRegression()
    clf.coef_, clf.intercept_, clf.classes_ = np.asarray(coef, np.float64), np.asarray(icpt, np.float64), np.array([0, 1])
    return clf


@pytest.mark.parametrize("mode", ["greedy", "beam"])
def test_class_round_of_one_million_rows(golden, mode):
    """BASELINE.json configs[3] at its own size: ONE round of 2^20 proposals through sample_pipeline.sample_round_arrays (device GMM
    draw, LR scoring + accept test, decode of every proposal by the whole-loop kernels, residue rows) at the reference's default
    dimensions (golden model A after 200 reference iterations).  (i) 1/256 shards of the same round - rows [r n/256, (r+1) n/256),
    drawn, scored and DECODED ALONE - equal those rows of the big round, at the start, in the middle and at the very end (grid-size /
    32-bit index errors at 10^6 rows would show here); (ii) accept mask identical and LR probabilities 1e-9 against the oracle on a
    10 k slice (/root/reference/density_modeling.py:50-60); (iii) decoded residues of 192 rows spread over the round equal
    oracle.decode's greedy / beam of the same z (/root/reference/sample_pipeline.py:129-139, models/model.py:225-385)."""
    import sample_pipeline as sp
    from cpg.synth import SyntheticPeptideLoader
    from density_modeling import mogQ
    from oracle import class_sampler as ocs, decode as odec
    gm = golden("model_A_200")
    P = weights_of(gm)
    m = build_model(P)
    m.eval()
    D = gm["greedy_z"].shape[1]
    rs = np.random.RandomState(3)
    Q = mogQ.from_params(np.ones(4) / 4, 0.3 * rs.randn(4, D), np.full((4, D), 0.9))
    Q.init_attr_classifiers({'amp': _clf(0.3 * rs.randn(1, D), np.zeros(1)), 'tox': _clf(0.3 * rs.randn(1, D), np.zeros(1))},
                            clf_targets={'amp': 1, 'tox': 0})
    Q.rng = 'device'
    ds = SyntheticPeptideLoader(4, 25, 'cuda', size=16)
    n = 1 << 20
    Q._philox = [2025, 0]
    frame, st = sp.sample_round_arrays(m, ds, Q, n, sample_mode=mode)
    assert st['proposed'] == n and st['decoded'] == n
    keys = list(frame)
    W = 256
    for r in (0, 1, 127, 254, 255):
        Q._philox = [2025, 0]
        f, _ = sp.sample_round_arrays(m, ds, Q, n, sample_mode=mode, shard=(r, W))
        lo, hi = r * n // W, (r + 1) * n // W
        for k in keys:
            a, b = f[k], frame[k][lo:hi]
            if k == 'letters':
                w = min(a.shape[1], b.shape[1])
                assert bool((a[:, w:] == 0).all()) and bool((b[:, w:] == 0).all())   # zero-filled beyond the residues
                a, b = a[:, :w], b[:, :w]
            assert torch.equal(a, b), (k, r)
    fr = {k: v[:10000].cpu().numpy() for k, v in frame.items()}
    z = fr['z']
    coef, icpt, tgt = (t.cpu().numpy() for t in Q._dev_clf)
    probs = np.stack([ocs.lr_prob(z, coef[i:i + 1], icpt[i:i + 1], int(tgt[i])) for i in range(coef.shape[0])])
    np.testing.assert_allclose(fr['clfZ_prob_accum'], probs.prod(0), rtol=1e-9)
    assert 0.02 < frame['accept_z'].float().mean().item() < 0.98
    # decoded residues against the oracle on rows spread over the whole round (incl. the last rows)
    rows = np.concatenate([np.arange(64), n // 2 + np.arange(64), n - 64 + np.arange(64)])
    zz = frame['z'][torch.from_numpy(rows).cuda()].cpu().numpy()
    cbit = frame['c'][torch.from_numpy(rows).cuda()].cpu().numpy()
    c = np.zeros((len(rows), 2), np.float32)
    c[np.arange(len(rows)), cbit] = 1
    if mode == "greedy":
        ids = odec.greedy(P, zz, c, 25)
    else:
        hyps, _ = odec.beam(P, zz, c, 25, beam_size=5, n_best=3)
        L = max(len(h[0]) for h in hyps)
        ids = np.full((len(rows), L), -1, np.int64)
        for i, h in enumerate(hyps):
            ids[i, :len(h[0])] = h[0]
    ref_letters, ref_n = sp.residue_rows(torch.from_numpy(ids), ds.n_vocab)
    got_letters = frame['letters'][torch.from_numpy(rows).cuda()].cpu().numpy()
    got_n = frame['n_res'][torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.array_equal(ref_n.numpy(), got_n)
    w = min(ref_letters.shape[1], got_letters.shape[1])
    assert np.array_equal(ref_letters.numpy()[:, :w], got_letters[:, :w])


# ------------------------------------------------------------------------------------------------ vocabulary projection, backward
@pytest.mark.parametrize("R,H,V,masked", [(51200, 512, 24, True), (8192, 1024, 24, False), (4099, 256, 20, True), (6000, 2048, 32, True),
                                          (5000, 768, 24, True)])   # H / 256 = 3: not a streaming shape (round-5 advisor finding)
def test_vocab_fc_backward_streaming_form_vs_f64(R, H, V, masked):
    """nn.Dropout(p_out) + nn.Linear(h_dim, n_vocab) backward (models/decoder.py:43-45,83) in its small-vocabulary streaming form
    (csrc/decode.hip: vocab_bwd_*_kernel) against f64 products of the same f32 inputs, with and without the keep-mask, writing and
    accumulating; rows not a multiple of anything, V below its padded width."""
    from cpg import ops
    g = torch.Generator().manual_seed(R + H + V)
    dev = torch.device('cuda:0')
    hs = torch.randn(R, H, generator=g)
    dl = torch.randn(R, V, generator=g) * torch.logspace(-6, 0, R).unsqueeze(1)      # row magnitudes over six orders
    w = torch.randn(V, H, generator=g) * 0.1
    keep = (torch.rand(R, H, generator=g) >= 0.3).to(torch.uint8) if masked else None
    scale = 1.0 / 0.7 if masked else 1.0
    m64 = keep.double() * scale if masked else torch.ones(1, dtype=torch.float64)
    dh_ref = (dl.double() @ w.double()) * m64
    dw_ref = dl.double().t() @ (hs.double() * m64)
    db_ref = dl.double().sum(0)
    dw_bound = dl.abs().double().t() @ (hs.abs().double() * m64)                 # sum of |terms| per output
    dh_bound = dl.abs().double() @ w.abs().double()
    hs_d, dl_d, w_d = hs.to(dev), dl.to(dev), w.to(dev)
    keep_d = keep.to(dev) if masked else None
    for accumulate in (0, 1):
        dhs = torch.full((R, H), float("nan"), device=dev)
        dw0 = torch.randn(V, H, generator=g)
        db0 = torch.randn(V, generator=g)
        dw, db = dw0.to(dev), db0.to(dev)
        ws = ops.workspace(ops.query("cpg_vocab_fc_bwd_workspace", R, H, V), dev)
        ops.call("cpg_vocab_fc_bwd", ops._p(dl_d), ops._p(hs_d), ops._p(keep_d), float(scale), ops._p(w_d), ops._p(dhs), ops._p(dw), ops._p(db),
                 R, H, V, accumulate, None, None, ops._p(ws), ws.numel(), ops._stream())
        torch.cuda.synchronize()
        dhs_c = dhs.cpu().double()
        assert torch.isfinite(dhs_c).all()
        assert ((dhs_c - dh_ref).abs() <= 4e-7 * dh_bound * scale + 1e-30).all()
        if masked:
            assert (dhs_c[keep == 0] == 0).all()
        dw_c = dw.cpu().double() - (dw0.double() if accumulate else 0)
        db_c = db.cpu().double() - (db0.double() if accumulate else 0)
        # f32 accumulation over R rows in a fixed order: a few ulp of the running sums + (accumulate) one rounding of the old value
        tol_w = 3e-6 * dw_bound + (2e-7 * dw0.abs().double() if accumulate else 0)
        assert ((dw_c - dw_ref).abs() <= tol_w).all(), float(((dw_c - dw_ref).abs() / dw_bound).max())
        assert ((db_c - db_ref).abs() <= 3e-6 * dl.abs().double().sum(0) + (2e-7 * db0.abs().double() if accumulate else 0)).all()


# ------------------------------------------------------------------------------------------------ small recurrences: whole-sequence launches
@pytest.mark.parametrize("B,H,T,reverse,with_rowc", [(32, 80, 25, False, False), (32, 80, 25, True, False), (32, 102, 25, False, True),
                                                     (45, 102, 7, False, True), (100, 20, 5, True, False), (64, 128, 9, False, True),
                                                     (2048, 102, 25, False, True), (7, 100, 3, True, True)])
def test_small_recurrence_whole_sequence_launch_vs_per_step(B, H, T, reverse, with_rowc):
    """The training recurrence of small GRUs as ONE launch per sequence (csrc/decode_fused.hip: gru_seq_small_fwd_kernel /
    gru_seq_small_bwd_kernel; the reference's default sizes h = 80 / 102, batch 32) against the per-step launches it replaces (option
    gru_small_seq = 0: csrc/gru.hip's step kernels, the path every golden fixture has pinned): states, saved gates, gate gradients and
    the initial-state gradient - ragged row counts, widths that are no multiple of anything, both directions, with and without the
    per-row constant input term, the final-state gradient and external gradients on every state."""
    from cpg import ops
    from cpg.ops import _p, _stream, call
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 7 + H)
    V = 24
    w_hh = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(dev)
    b_hh = (torch.randn(3 * H, generator=g) * 0.1).to(dev)
    tab = (torch.randn(V, 3 * H, generator=g) * 0.4).to(dev)
    rowc = (torch.randn(B, 3 * H, generator=g) * 0.4).to(dev) if with_rowc else None
    tok = torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev)
    h0 = (torch.randn(B, H, generator=g) * 0.5).to(dev)
    dhs = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    dlast = (torch.randn(B, H, generator=g) * 0.1).to(dev)
    out = {}
    try:
        for small in (1, 0):
            ops.set_option("gru_small_seq", small)
            hs = torch.full((T + 1, B, H), float("nan"), device=dev)
            hs[T if reverse else 0] = h0
            gates = torch.full((T, 4, B, H), float("nan"), device=dev)
            call("cpg_gru_seq_fwd", T, B, H, int(reverse), _p(w_hh), _p(b_hh), _p(tok), _p(tab), _p(rowc), None, _p(hs), _p(gates), 0, B, None,
                 _p(ops.weight_exp(w_hh)), _stream())
            dG = torch.full((T, B, 4 * H), float("nan"), device=dev)
            scr = torch.empty(2, B, H, device=dev)
            dh0 = torch.full((B, H), float("nan"), device=dev)
            call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(w_hh), _p(hs), _p(gates), _p(dhs), _p(dlast), _p(dG), _p(scr), _p(dh0), 0, B, None,
                 None, None, 0, _stream())
            torch.cuda.synchronize()
            out[small] = (hs.clone(), gates.clone(), dG.clone(), dh0.clone())
    finally:
        ops.set_option("gru_small_seq", None)
    (hs1, g1, dG1, d01), (hs0, g0, dG0, d00) = out[1], out[0]
    for a in out[1]:
        assert torch.isfinite(a).all()
    # forward: f16-pair products + hardware exp / rcp cell forms against the per-step kernel's engine: a few 1e-6 on O(1) values
    assert (hs1 - hs0).abs().max().item() <= 2e-5
    assert (g1 - g0).abs().max().item() <= 4e-5
    # backward on each path's OWN saved forward values: compare through the gradient scale
    sc = dG0.abs().max().item()
    assert (dG1 - dG0).abs().max().item() <= 3e-4 * sc
    assert (d01 - d00).abs().max().item() <= 3e-4 * max(d00.abs().max().item(), 1e-6)
    # the backward kernels alone, on identical inputs (the per-step path's forward values): exact-f32 products on both sides
    ops.set_option("gru_small_seq", 1)
    try:
        dG = torch.full((T, B, 4 * H), float("nan"), device=dev)
        dh0 = torch.full((B, H), float("nan"), device=dev)
        call("cpg_gru_seq_bwd", T, B, H, int(reverse), _p(w_hh), _p(hs0), _p(g0), _p(dhs), _p(dlast), _p(dG), _p(torch.empty(2, B, H, device=dev)),
             _p(dh0), 0, B, None, None, None, 0, _stream())
        torch.cuda.synchronize()
    finally:
        ops.set_option("gru_small_seq", None)
    assert (dG - dG0).abs().max().item() <= 2e-5 * sc
    assert (dh0 - d00).abs().max().item() <= 2e-5 * max(d00.abs().max().item(), 1e-6)
