"""Pin the numpy oracle to vectors produced by the real reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import weights_of
from oracle import wae, decode, optim, class_sampler

# A_200: config A after 200 reference train_vae iterations; skip: a model built with decoder skip connections
MODELS = ["A", "micro", "enc2", "A_200", "skip"]


def rnd_of(g):
    return {k: g[k] for k in ("eps", "c", "wd_mask", "out_mask", "z_prior_full", "z_prior_rf", "rf_w", "rf_b")}


@pytest.mark.parametrize("name", MODELS)
def test_encoder(golden, name):
    g = golden("model_" + name)
    mu, lv, _ = wae.encoder_fwd(weights_of(g), g["ids"])
    np.testing.assert_allclose(mu, g["enc_mu"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(lv, g["enc_logvar"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("name", MODELS)
def test_decoder_teacher_forced(golden, name):
    g = golden("model_" + name)
    P = weights_of(g)
    lg, _ = wae.decoder_fwd(P, g["ids"], g["z"], g["c"], g["wd_mask"], g["out_mask"], 0.3)
    np.testing.assert_allclose(lg, g["logits_train"], atol=5e-6, rtol=1e-5)
    ones = np.ones_like(g["out_mask"])
    lg, _ = wae.decoder_fwd(P, g["ids"], g["z"], g["c"], g["wd_mask_eval"], ones, 0.0)
    np.testing.assert_allclose(lg, g["logits_eval"], atol=5e-6, rtol=1e-5)
    lg, _ = wae.decoder_fwd(P, g["ids"], g["enc_mu"], g["c_lab"], g["wd_mask_max"], ones, 0.0)
    np.testing.assert_allclose(lg, g["logits_max"], atol=5e-6, rtol=1e-5)


@pytest.mark.parametrize("name", MODELS)
def test_losses_and_grads(golden, name):
    g = golden("model_" + name)
    P = weights_of(g)
    variants = [""] + [p for p in ("v1.", "v2.") if p + "regu" in g]
    for p in variants:
        regu = str(g[p + "regu"])
        terms, G, aux = wae.train_loss_and_grads(P, g["ids"], rnd_of(g), float(g["beta"]), float(g["lam_l1"]),
                                                 float(g["lam_kl"]), regu)
        np.testing.assert_allclose(aux["z"], g["z"], atol=2e-6)
        assert abs(terms["recon"] - g["loss_recon"]) < 1e-5
        assert abs(terms["kl"] - g["loss_kl"]) < 1e-5
        assert abs(terms["klmu"] - g["loss_klmu"]) < 1e-5
        assert abs(terms["l1"] - g["loss_l1"]) < 1e-4
        assert abs(terms["mmd"] - g["loss_mmd_full"]) < 1e-5
        assert abs(terms["mmdrf"] - g["loss_mmd_rf"]) < 1e-5
        assert abs(terms["total"] - g[p + "loss_total"]) < 1e-5
        np.testing.assert_allclose(aux["dlogits"], g[p + "g.logits"], atol=1e-7, rtol=1e-4)
        np.testing.assert_allclose(aux["dz"], g[p + "g.z"], atol=2e-7, rtol=2e-4)
        for k, v in G.items():
            ref = g[p + "g." + k]
            np.testing.assert_allclose(v, ref, atol=5e-7 + 2e-5 * np.abs(ref).max(), rtol=0, err_msg=f"{p}{k}")


@pytest.mark.parametrize("name", MODELS)
def test_greedy_bit_exact(golden, name):
    g = golden("model_" + name)
    P = weights_of(g)
    ids, logits = decode.greedy(P, g["greedy_z"], g["greedy_c"], 25, return_logits=True)
    assert np.array_equal(ids, g["greedy_ids"])
    np.testing.assert_allclose(logits, g["greedy_logits"], atol=1e-5)
    ids = decode.greedy(P, g["greedy_z"], g["greedy_c"], 25, prevent_empty=True)
    assert np.array_equal(ids, g["greedy_ids_prevent_empty"])


@pytest.mark.parametrize("name", MODELS)
def test_beam(golden, name):
    g = golden("model_" + name)
    P = weights_of(g)
    ref = g["beam_hyps"]
    n = ref.shape[0]
    hyps, _ = decode.beam(P, g["greedy_z"][:n], g["greedy_c"][:n], 25, beam_size=5, n_best=3)
    for i in range(n):
        for j in range(3):
            want = [int(t) for t in ref[i, j] if t >= 0]
            assert hyps[i][j] == want, (i, j)


@pytest.mark.parametrize("name", ["micro_clip", "micro_noclip", "A_clip"])
def test_train_trajectory(golden, name):
    """params after k reference train_vae iterations incl. F6 duplicate-embedding semantics."""
    g = golden("train_" + name)
    P = {k: v.copy() for k, v in weights_of(g, "w0.").items() if not k.startswith("classifier")}
    opt = optim.AdamClip(P, lr=1e-3, max_norm=float(g["clip"]))
    n_total = g["batches"].shape[0]
    end_it = int(g["beta_end_iter"])
    regu = str(g["z_regu"])
    for it in range(n_total):
        beta = 1.0 if it <= 0 else (2.0 if it >= end_it else 1.0 + (it / end_it))
        rnd = {k: g[k][it] for k in ("eps", "c", "wd_mask", "out_mask", "z_prior_full", "z_prior_rf")}
        rnd["rf_w"], rnd["rf_b"] = g["rf_w"], g["rf_b"]
        terms, G, aux = wae.train_loss_and_grads(P, g["batches"][it], rnd, beta, 0.0, 1e-3, regu)
        if it == 0:
            assert abs(terms["total"] - g["log0.train_L_vae"]) < 1e-5
            assert abs(terms["recon"] - g["log0.train_L_vae_recon"]) < 1e-5
            assert abs(terms["kl"] - g["log0.train_L_vae_kl"]) < 1e-5
            assert abs(terms["mmd"] - g["log0.train_L_wae_mmd"]) < 1e-5
            assert abs(terms["mmdrf"] - g["log0.train_L_wae_mmdrf"]) < 1e-5
            assert abs(terms["klmu"] - g["log0.train_z_logvar_KL_penalty"]) < 1e-5
            assert abs(np.abs(aux["mu"]).mean() - g["log0.train_z_mu_L1"]) < 1e-6
        opt.step(P, G)
        snap = f"w{it + 1}."
        if snap + "word_emb.weight" in g:
            for k in P:
                # Adam's update is lr*m/(sqrt(v)+eps): elements with |g| ~ eps are ill-conditioned, so allow
                # 2% of one lr step; the F6 double update of the embedding is a ~1e-3 effect (checked below)
                np.testing.assert_allclose(P[k], g[snap + k], atol=2e-5, rtol=0, err_msg=f"{snap}{k}")
        if it == 0:
            w0, w1 = g["w0.word_emb.weight"], g["w1.word_emb.weight"]
            moved = np.abs(w1 - w0)[np.abs(G["word_emb.weight"]) > 1e-4]
            # two sequential Adam updates with the same gradient move a weight by ~2*lr, not lr
            assert moved.size and np.median(moved) > 1.7e-3


def test_class_rejection(golden):
    g = golden("class_small")
    z = class_sampler.gmm_sample(g["gmm_means"], g["gmm_covars"], g["counts"], g["normals"])
    assert np.array_equal(z, g["z"])
    clfs = [(g["amp_coef"], g["amp_intercept"], 1), (g["tox_coef"], g["tox_intercept"], 0)]
    probs, accum, acc = class_sampler.rejection_mask(z, clfs, g["uniforms"])
    np.testing.assert_allclose(probs[0], g["prob_amp"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(probs[1], g["prob_tox"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(accum, g["prob_accum"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(acc, g["accepted"])


@pytest.mark.parametrize("mode,temp", [("none_softmax", 1.0), ("greedy_softmax", 1.0), ("greedy_softmax", 0.7),
                                       ("categorical_softmax", 0.9)])
def test_soft_sampling_modes_golden(golden, mode, temp):
    """sample_G soft modes (models/model.py:337-359) vs the reference's outputs, incl. its quirks (ids frozen at <start> for
    none_softmax, soft row zeroed at the <eos> step); the categorical draws are replayed from the fixture."""
    from oracle import decode as odecode
    g = golden("soft_A")
    P = weights_of(g)
    tag = f"{mode}_t{temp}"
    ids, soft = odecode.soft_sample(P, g["z"], g["c"], 25, mode, temp, sampled=g[tag + ".ids"])
    assert np.array_equal(ids, g[tag + ".ids"])
    np.testing.assert_allclose(soft, g[tag + ".soft"], atol=2e-6)


def test_sqdist_forms_agree():
    """The Gram form of |x-y|^2 the oracle switches to at config-B size equals the reference's broadcast form."""
    rs = np.random.RandomState(0)
    x, y = rs.randn(64, 100).astype(np.float32), rs.randn(64, 100).astype(np.float32)
    a, b = wae._sqdist(x, y, "broadcast"), wae._sqdist(x, y, "gram")
    np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(wae._sqdist(x, x, "gram").diagonal(), 0.0, atol=1e-10)
    l1, g1 = wae.mmd_full_kernel(x, y, 7.0)
    old = wae.SQDIST_BROADCAST_LIMIT
    try:
        wae.SQDIST_BROADCAST_LIMIT = 0
        l2, g2 = wae.mmd_full_kernel(x, y, 7.0)
    finally:
        wae.SQDIST_BROADCAST_LIMIT = old
    assert abs(float(l1) - float(l2)) < 1e-7
    np.testing.assert_allclose(g1, g2, atol=1e-9)


@pytest.mark.parametrize("kernel", ["gaussian", "laplace", "energy"])
def test_mmd_kernels(golden, kernel):
    """compute_mmd_kernel's three kernels (losses.py:96-108): loss and d loss / d z1 against the reference's autograd."""
    import torch
    from oracle import torch_ref
    g = golden("mmd_kernels")
    loss, dz = wae.mmd_full_kernel(g["z1"], g["z2"], float(g["sigma"]), kernel)
    ref_l, ref_g = float(g[kernel + ".loss"]), g[kernel + ".dz1"]
    assert abs(float(loss) - ref_l) < 2e-6 * max(1.0, abs(ref_l))
    assert np.abs(dz - ref_g).max() < 1e-5 * np.abs(ref_g).max()
    z1 = torch.tensor(g["z1"], requires_grad=True)
    lt = torch_ref.mmd_full_kernel(z1, torch.tensor(g["z2"]), float(g["sigma"]), kernel)
    lt.backward()
    assert abs(lt.item() - ref_l) < 1e-6 * max(1.0, abs(ref_l))
    assert np.abs(z1.grad.numpy() - ref_g).max() < 1e-5 * np.abs(ref_g).max()


# ---- the torch-CPU restatement that bench.py times as `cpu_baseline` (oracle/torch_ref.py), pinned to the same vectors
@pytest.mark.parametrize("name", MODELS)
def test_torch_ref_losses_and_grads(golden, name):
    import torch
    from oracle import torch_ref
    g = golden("model_" + name)
    P = weights_of(g)
    rnd = {k: torch.from_numpy(np.ascontiguousarray(g[k])) for k in
           ("eps", "c", "wd_mask", "out_mask", "z_prior_full", "z_prior_rf", "rf_w", "rf_b")}
    m = torch_ref.RefWAE.from_state(P)
    terms, aux = torch_ref.train_loss(m, torch.from_numpy(g["ids"]), rnd, float(g["beta"]), float(g["lam_l1"]),
                                      float(g["lam_kl"]), str(g["regu"]))
    terms["total"].backward()
    for a, b in (("recon", "loss_recon"), ("kl", "loss_kl"), ("klmu", "loss_klmu"), ("mmd", "loss_mmd_full"),
                 ("mmdrf", "loss_mmd_rf"), ("total", "loss_total")):
        assert abs(terms[a].item() - float(g[b])) < 1e-5, a
    np.testing.assert_allclose(aux["logits"].detach().numpy(), g["logits_train"], atol=5e-6, rtol=1e-5)
    for k, p in m.named_parameters():
        ref = g["g." + m.ref_name(k)]
        np.testing.assert_allclose(p.grad.numpy(), ref, atol=5e-7 + 2e-5 * np.abs(ref).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize("name", ["micro_clip", "A_clip"])
def test_torch_ref_trajectory(golden, name):
    """k iterations of clip + Adam with the duplicate embedding entry (F6) reproduce the reference's parameters."""
    import torch
    from oracle import torch_ref
    g = golden("train_" + name)
    m = torch_ref.RefWAE.from_state(weights_of(g, "w0."))
    tr = torch_ref.Trainer(m, lr=1e-3, clip=float(g["clip"]))
    end_it = int(g["beta_end_iter"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for it in range(g["batches"].shape[0]):
        beta = 1.0 if it <= 0 else (2.0 if it >= end_it else 1.0 + it / end_it)
        rnd = {k: t(g[k][it]) for k in ("eps", "c", "wd_mask", "out_mask", "z_prior_full", "z_prior_rf")}
        rnd.update(rf_w=t(g["rf_w"]), rf_b=t(g["rf_b"]))
        tr.step(t(g["batches"][it]), rnd, beta=beta, lam_l1=0.0, lam_kl=1e-3, z_regu=str(g["z_regu"]))
        snap = f"w{it + 1}."
        if snap + "word_emb.weight" in g:
            for k, p in m.named_parameters():
                np.testing.assert_allclose(p.detach().numpy(), g[snap + m.ref_name(k)], atol=3e-5, rtol=0, err_msg=snap + k)  # Adam is ill-conditioned where |g| ~ eps


@pytest.mark.parametrize("tag,kw", [("t1.0", dict(temp=1.0)), ("t0.7", dict(temp=0.7)), ("t1.0_pe", dict(temp=1.0, prevent_empty=True))])
def test_categorical_replay(golden, tag, kw):
    """sample_G 'categorical' (models/model.py:308-309): with the uniforms that reproduce the reference's captured
    torch.multinomial draws (tests/golden/make_golden.py:categorical_vectors) the inverse-CDF restatement gives its ids."""
    g = golden("categorical_A")
    ids = decode.categorical(weights_of(g), g["z"], g["c"], 25, g[tag + ".u"], **kw)
    ref = g[tag + ".ids"]
    assert np.array_equal(ids, ref[:, :ids.shape[1]]) and (ref[:, ids.shape[1]:] == 1).all()
    assert (ref == 3).any(1).mean() > 0.3   # the fixture does exercise <eos> / finished rows


def test_encoder_interlayer_dropout(golden):
    """nn.GRU(dropout=p_dropout) between the layers of a 2-layer encoder in train mode (models/encoder.py:25-30), mask replayed from
    the reference's own draw: (mu, logvar) and the reference's autograd gradients of sum(mu*gmu + logvar*glv); eval mode = no mask."""
    g = golden("encdrop")
    P = weights_of(g)
    mu, lv, ec = wae.encoder_fwd(P, g["ids"], g["enc_keep"], float(g["p"]))
    np.testing.assert_allclose(mu, g["mu_train"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(lv, g["logvar_train"], atol=2e-6, rtol=1e-5)
    G = {}
    demb = wae.encoder_bwd(P, g["gmu"], g["glv"], ec, G)
    for k, v in G.items():
        ref = g["g." + k]
        np.testing.assert_allclose(v, ref, atol=5e-7 + 2e-5 * np.abs(ref).max(), rtol=0, err_msg=k)
    np.testing.assert_allclose(demb, g["g.word_emb.weight"], atol=5e-7 + 2e-5 * np.abs(g["g.word_emb.weight"]).max(), rtol=0)
    mu, lv, _ = wae.encoder_fwd(P, g["ids"])
    np.testing.assert_allclose(mu, g["mu_eval"], atol=2e-6, rtol=1e-5)
    assert np.abs(g["mu_eval"] - g["mu_train"]).max() > 1e-3      # the mask matters


def test_sampling_in_train_mode(golden):
    """generate_sentences(eval_mode=False) (models/model.py:216-221): out-dropout live in every decode step - greedy ids bit-exact
    and beam-5 hypotheses exact with the reference's captured masks; without them the ids are the eval-mode ids."""
    g = golden("sample_train")
    P = weights_of(g)
    ids = decode.greedy(P, g["z"], g["c"], 25, out_keep=g["greedy_keep"])
    assert np.array_equal(ids, g["greedy_ids"])
    assert np.array_equal(decode.greedy(P, g["z"], g["c"], 25), g["greedy_ids_eval_mode"])
    n = g["beam_hyps"].shape[0]
    hyps, _ = decode.beam(P, g["z"][:n], g["c"][:n], 25, beam_size=5, n_best=3, out_keep=g["beam_keep"])
    for i in range(n):
        for j in range(3):
            assert hyps[i][j] == [int(t) for t in g["beam_hyps"][i, j] if t >= 0], (i, j)


# ------------------------------------------------------------------------------------------------ multi-layer decoder extension
def _torch_dec(cell, V, E, H, L, seed):
    """A torch.nn.GRU / nn.LSTM(num_layers=L) decoder with the reference's state-dict names (decoder.rnn.*_l{l}); -> (rnn, fc, emb, P)."""
    import torch
    import torch.nn as nn
    torch.manual_seed(seed)
    emb = nn.Embedding(V, E, 1)
    rnn = (nn.GRU if cell == "gru" else nn.LSTM)(E + H, H, num_layers=L, batch_first=True)
    fc = nn.Linear(H, V)
    with torch.no_grad():
        fc.weight.mul_(5.0)
        fc.bias[3] += 0.7
    P = {"word_emb.weight": emb.weight.detach().numpy().copy(), "decoder.fc.1.weight": fc.weight.detach().numpy().copy(),
         "decoder.fc.1.bias": fc.bias.detach().numpy().copy()}
    P.update({"decoder.rnn." + k: v.detach().numpy().copy() for k, v in rnn.state_dict().items()})
    return rnn, fc, emb, P


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_multilayer_decoder_oracle_pinned_to_torch(cell):
    """The multi-layer decoder EXTENSION of oracle/decode.py (BASELINE.json configs[4] "2-layer dec"; the reference has one layer,
    models/decoder.py:40-41 - parity unpinned against it) against torch.nn.GRU / nn.LSTM(num_layers=2) driven step by step with
    h0 = [z;c] in every layer (c0 = 0): greedy ids exact + per-step logits 1e-5, and a beam search that uses the same Beam
    bookkeeping but torch for the step and `h[:, idx]` for the reorder of EVERY layer's state (models/model.py:378-385)."""
    import torch
    from oracle import decode as odec
    V, E, Z, L, T, N, K = 24, 10, 14, 2, 12, 6, 4
    H = Z + 2
    rnn, fc, emb, P = _torch_dec(cell, V, E, H, L, seed=3)
    assert odec.n_dec_layers(P) == L
    rs = np.random.RandomState(1)
    z = rs.randn(N, Z).astype(np.float32)
    c = np.zeros((N, 2), np.float32)
    c[np.arange(N), rs.randint(0, 2, N)] = 1

    def tstep(tok, zc, st):
        x = torch.cat([emb(torch.from_numpy(tok)), zc], 1).unsqueeze(1)
        out, st = rnn(x, st)
        return fc(out[:, 0]), st

    def init(zc):
        h0 = zc.unsqueeze(0).repeat(L, 1, 1).contiguous()
        return h0 if cell == "gru" else (h0, torch.zeros_like(h0))

    with torch.no_grad():
        zc = torch.from_numpy(np.concatenate([z, c], 1))
        st = init(zc)
        tok = np.full(N, 2, np.int64)
        fin = np.zeros(N, bool)
        cols, lgs = [tok], []
        for i in range(T):
            lg, st = tstep(tok, zc, st)
            lgs.append(lg.numpy())
            tok = lg.numpy().argmax(1).astype(np.int64)
            tok[fin] = 1
            fin |= tok == 3
            cols.append(tok)
            if fin.all():
                break
        ids, logits = odec.greedy(P, z, c, T, return_logits=True, cell=cell)
        assert np.array_equal(ids, np.stack(cols, 1))
        np.testing.assert_allclose(logits, np.stack(lgs, 1), atol=1e-5)
        # beam: torch step + reorder of all layers
        zcK = zc.repeat(K, 1)
        st = init(zcK)
        beams = [odec._Beam(K, 2, 1) for _ in range(N)]
        tok = np.stack([b.next_ys[-1] for b in beams]).T.reshape(-1)
        for step in range(T):
            lg, st = tstep(tok, zcK, st)
            lg = lg.numpy().reshape(K, N, -1)
            parts = [st] if cell == "gru" else list(st)
            views = [p.view(L, K, N, H) for p in parts]
            for j, b in enumerate(beams):
                if not b.done():
                    b.advance(odec._log_softmax(lg[:, j]))
                idx = torch.from_numpy(np.asarray(b.prev_ks[-1]))
                for v in views:
                    v[:, :, j] = v[:, idx, j].clone()
            tok = np.stack([b.next_ys[-1] for b in beams]).T.reshape(-1)
            if all(b.done() for b in beams):
                break
        want = [b.best()[0] for b in beams]
        got, _ = odec.beam(P, z, c, T, beam_size=K, n_best=2, cell=cell)
        assert got == want
        lens = {len(h) for s in want for h in s}
        assert len(lens) > 1
