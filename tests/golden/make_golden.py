#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference (read-only at
/root/reference) in the build container and running its own functions on CPU.

Only DATA leaves this script (npz files with inputs, weights by state-dict key,
captured random draws and the reference's outputs); no reference source travels.
Run:  python tests/golden/make_golden.py            (needs /root/reference)

What is captured:
  model_<name>.npz   weights, ids, encoder mu/logvar, captured randomness
                     (eps, c, word-dropout mask, out-dropout mask, z_prior, rf_w, rf_b),
                     teacher-forced logits (train + eval mode), every loss term,
                     gradients of the train_vae loss wrt every parameter, greedy decode ids
                     (+ per-step logits), beam-5 hypotheses + scores.
  train_<name>.npz   parameters after k reference train_vae iterations (F6 duplicate-param
                     semantics of clip_grad_norm_ + Adam included) with all per-iteration
                     randomness, and the metrics the reference logged at it=0.
  class_small.npz    sklearn GMM/LR parameters, replayed numpy draws, z, probs, accept mask
                     from density_modeling.mogQ.rejection_sample.
"""
import os
import sys
import types
import contextlib

import numpy as np
import torch

REF = os.environ.get("CPG_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

# ---- stubs for packages the image lacks (pure logging / plotting, not arithmetic) ----
_tbl = types.ModuleType("tensorboard_logger")
_tbl_inner = types.ModuleType("tensorboard_logger.tensorboard_logger")


class _NullLogger:
    def __init__(self, *a, **k):
        pass

    def log_value(self, *a, **k):
        pass

    def log_histogram(self, *a, **k):
        pass

    def log_images(self, *a, **k):
        pass


for _n in ("configure", "log_value", "log_histogram", "log_images"):
    setattr(_tbl_inner, _n, lambda *a, **k: None)
_tbl_inner.Logger = _NullLogger
_tbl.tensorboard_logger = _tbl_inner
sys.modules["tensorboard_logger"] = _tbl
sys.modules["tensorboard_logger.tensorboard_logger"] = _tbl_inner
for _n in ("h5py", "seaborn"):
    sys.modules.setdefault(_n, types.ModuleType(_n))
# matplotlib is present; vis.scripts.covar imports it
os.environ.setdefault("MPLBACKEND", "Agg")

import cfg as rcfg  # noqa: E402  (reference cfg)
import losses as rlosses  # noqa: E402
import utils as rutils  # noqa: E402
from models.model import RNN_VAE  # noqa: E402
import train_vae as rtrain  # noqa: E402
import tb_json_logger as rtbj  # noqa: E402

CPU = torch.device("cpu")
T = 25
V = 24


# ------------------------------------------------------------------ helpers
def synth_ids(B, T, V, gen):
    """SURVEY 8d synthetic peptides: <start> aa{L} <eos> <pad>*, L~U{5..T-2}."""
    ids = torch.full((B, T), 1, dtype=torch.long)
    L = torch.randint(5, T - 1, (B,), generator=gen)
    for b in range(B):
        l = int(L[b])
        ids[b, 0] = 2
        ids[b, 1:1 + l] = torch.randint(4, V, (l,), generator=gen)
        ids[b, 1 + l] = 3
    return ids


def model_kwargs(z_dim, enc_h, enc_layers=1, emb_dim=150, p_word=0.3, p_out=0.3, skip=False, enc_dropout=0.0):
    return dict(
        z_dim=z_dim, c_dim=2, emb_dim=emb_dim, pretrained_emb=None, freeze_embeddings=False,
        flow=0, flow_type='',
        E_args=dict(h_dim=enc_h, biGRU=True, layers=enc_layers, p_dropout=enc_dropout),
        G_args=dict(G_class='gru',
                    GRU_args=dict(p_word_dropout=p_word, p_out_dropout=p_out, skip_connetions=skip),
                    deconv_args=dict(max_seq_len=T, num_filters=100, kernel_size=4, num_deconv_layers=3,
                                     useRNN=False, temperature=1.0, use_batch_norm=True,
                                     num_conv_layers=2, add_final_conv_layer=True)),
        C_args=dict(min_filter_width=3, max_filter_width=5, num_filters=100, dropout=0.5))


class Capture:
    """Records the random draws the reference makes (torch + numpy), in call order."""

    def __init__(self):
        self.log = []

    @contextlib.contextmanager
    def on(self):
        o_randn, o_randn_like, o_rand = torch.randn, torch.randn_like, torch.rand
        o_multi, o_binom, o_unif = np.random.multinomial, np.random.binomial, np.random.uniform

        def wrap(name, fn):
            def inner(*a, **k):
                r = fn(*a, **k)
                self.log.append((name, r.clone() if torch.is_tensor(r) else np.array(r)))
                return r
            return inner
        torch.randn, torch.randn_like, torch.rand = wrap('randn', o_randn), wrap('randn_like', o_randn_like), wrap('rand', o_rand)
        np.random.multinomial, np.random.binomial, np.random.uniform = \
            wrap('multinomial', o_multi), wrap('binomial', o_binom), wrap('uniform', o_unif)
        try:
            yield self
        finally:
            torch.randn, torch.randn_like, torch.rand = o_randn, o_randn_like, o_rand
            np.random.multinomial, np.random.binomial, np.random.uniform = o_multi, o_binom, o_unif

    def take(self, name):
        for i, (n, v) in enumerate(self.log):
            if n == name:
                del self.log[i]
                return v
        raise KeyError(name)


class DropHook:
    """Forward hook on the decoder's nn.Dropout: recovers the keep-mask (0/1)."""

    def __init__(self, module):
        self.masks = []
        self.h = module.register_forward_hook(self._hook)

    def _hook(self, mod, inp, out):
        x = inp[0]
        if mod.training and mod.p > 0:
            keep = (out != 0) | (x == 0)
            self.masks.append(keep.to(torch.uint8).clone())
        else:
            self.masks.append(torch.ones_like(x, dtype=torch.uint8))

    def remove(self):
        self.h.remove()


def np_state(model):
    return {"w." + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def build(seed, **kw):
    torch.manual_seed(seed)
    m = RNN_VAE(n_vocab=V, max_seq_len=T, **kw)
    m.device = CPU  # reference hard-codes cuda (models/model.py:41); api.py:96 patches it the same way
    return m


def reset_rf():
    rlosses.rf.clear()


# ------------------------------------------------------------------ forward / loss / grad vectors
def model_vectors(name, seed, B, N_greedy, N_beam, z_regu_variants, model=None, **kw):
    out = {}
    if model is None:
        model = build(seed, **kw)
    out.update(np_state(model))
    gen = torch.Generator().manual_seed(seed)
    ids = synth_ids(B, T, V, gen)
    out["ids"] = ids.numpy()
    Z = model.z_dim

    # encoder (deterministic)
    with torch.no_grad():
        mu, logvar = model.forward_encoder(ids)
    out["enc_mu"], out["enc_logvar"] = mu.numpy(), logvar.numpy()

    # ---- one reference training-loss evaluation per z_regu variant, capturing randomness
    beta = 1.25
    for vi, regu in enumerate(z_regu_variants):
        model.zero_grad()
        model.train()
        reset_rf()
        torch.manual_seed(seed + 17)
        np.random.seed(seed + 17)
        cap = Capture()
        hook = DropHook(model.decoder.fc[0])
        with cap.on():
            (z_mu, z_logvar), (z, c), logits = model(ids, q_c='prior', sample_z=1)
            recon = rlosses.recon_dec(ids, logits)
            kl = rlosses.kl_gaussianprior(z_mu, z_logvar)
            mmd = rlosses.wae_mmd_gaussianprior(z, method='full_kernel')
            mmdrf = rlosses.wae_mmd_gaussianprior(z, method='rf')
        hook.remove()
        l1 = z_logvar.abs().sum(1).mean(0)
        klmu = rlosses.kl_gaussian_sharedmu(z_mu, z_logvar)
        regu_val = {'kl': kl, 'mmd': mmd, 'mmdrf': mmdrf}[regu]
        lam_l1, lam_kl = 0.05, 1e-3
        loss = recon + beta * regu_val + lam_l1 * l1 + lam_kl * klmu
        z.retain_grad()
        logits.retain_grad()
        loss.backward()
        p = "" if vi == 0 else f"v{vi}."
        if vi == 0:
            out["eps"] = cap.take('randn').numpy()
            out["c"] = cap.take('multinomial').astype(np.float32)
            out["wd_mask"] = cap.take('binomial').astype(np.uint8)
            out["out_mask"] = hook.masks[0].numpy()
            out["z_prior_full"] = cap.take('randn_like').numpy()
            out["z_prior_rf"] = cap.take('randn_like').numpy()
            out["rf_w"] = cap.take('randn').numpy()
            out["rf_b"] = cap.take('rand').numpy() * (2 * np.pi)
            # reference scales rand by 2*pi (losses.py:74); store the scaled basis actually used
            out["rf_b"] = rlosses.rf['gaussian'][1].numpy().copy()
            assert np.allclose(out["rf_w"], rlosses.rf['gaussian'][0].numpy())
            out["z"] = z.detach().numpy()
            out["logits_train"] = logits.detach().numpy()
            out["loss_recon"] = np.float32(recon.item())
            out["loss_kl"] = np.float32(kl.item())
            out["loss_klmu"] = np.float32(klmu.item())
            out["loss_l1"] = np.float32(l1.item())
            out["loss_mmd_full"] = np.float32(mmd.item())
            out["loss_mmd_rf"] = np.float32(mmdrf.item())
            out["beta"] = np.float32(beta)
            out["lam_l1"] = np.float32(lam_l1)
            out["lam_kl"] = np.float32(lam_kl)
        out[p + "regu"] = np.array(regu)
        out[p + "loss_total"] = np.float32(loss.item())
        out[p + "g.z"] = z.grad.numpy().copy()
        out[p + "g.logits"] = logits.grad.numpy().copy()
        for k, prm in model.named_parameters():
            if k.startswith("classifier"):
                continue
            out[p + "g." + k] = (prm.grad.numpy().copy() if prm.grad is not None
                                 else np.zeros(tuple(prm.shape), np.float32))

    # eval-mode teacher forcing (out-dropout off; word dropout still applied: decoder.py:117-133)
    model.eval()
    np.random.seed(seed + 29)
    cap = Capture()
    with cap.on(), torch.no_grad():
        zz = torch.from_numpy(out["z"])
        cc = torch.from_numpy(out["c"])
        logits_eval = model.forward_decoder(ids, zz, cc)
    out["wd_mask_eval"] = cap.take('binomial').astype(np.uint8)
    out["logits_eval"] = logits_eval.numpy()

    # sample_z='max', labels as q_c (model.py:179-183)
    np.random.seed(seed + 31)
    labels = (torch.arange(B) % 2).long()
    cap = Capture()
    with cap.on(), torch.no_grad():
        (_, _), (z_max, c_lab), logits_max = model(ids, q_c=labels, sample_z='max')
    out["labels"] = labels.numpy()
    out["wd_mask_max"] = cap.take('binomial').astype(np.uint8)
    out["logits_max"] = logits_max.numpy()
    out["c_lab"] = c_lab.numpy()

    # ---- greedy decode + per-step logits (G6)
    g = torch.Generator().manual_seed(seed + 3)
    zs = torch.randn(N_greedy, Z, generator=g)
    cs = torch.zeros(N_greedy, 2)
    cs[torch.arange(N_greedy), torch.randint(0, 2, (N_greedy,), generator=g)] = 1.0
    with torch.no_grad():
        sent, _, _ = model.generate_sentences(N_greedy, zs, cs, sample_mode='greedy')
        # per-step logits for margin analysis: replay with forward_sample
        model.eval()
        h = model.decoder.init_hidden(zs, cs).unsqueeze(0)
        tok = torch.full((N_greedy,), 2, dtype=torch.long)
        step_logits = []
        for i in range(sent.size(1) - 1):
            lg, h = model.decoder.forward_sample(None, tok, zs, cs, h)
            step_logits.append(lg.clone())
            tok = sent[:, i + 1]
        model.train()
    out["greedy_z"], out["greedy_c"] = zs.numpy(), cs.numpy()
    out["greedy_ids"] = sent.numpy()
    out["greedy_logits"] = torch.stack(step_logits, 1).numpy()
    with torch.no_grad():
        sent_pe, _, _ = model.generate_sentences(N_greedy, zs, cs, sample_mode='greedy', prevent_empty=True)
    out["greedy_ids_prevent_empty"] = sent_pe.numpy()

    # ---- beam-5 / n_best 3 (G7)
    zb, cb = zs[:N_beam], cs[:N_beam]
    with torch.no_grad():
        model.eval()
        hyps = model.sample_G(N_beam, zb, cb, sample_mode='beam', beam_size=5, n_best=3)
        model.train()
    arr = np.full((N_beam, 3, T + 2), -1, dtype=np.int64)
    for i, hs in enumerate(hyps):
        for j, hyp in enumerate(hs):
            ids_ = [int(t) for t in hyp]
            arr[i, j, :len(ids_)] = ids_
    out["beam_hyps"] = arr

    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **out)
    print(f"model_{name}.npz: {len(out)} arrays")
    return model


# ------------------------------------------------------------------ train_vae trajectory (G8/G9)
class FakeBatch:
    def __init__(self, text):
        self.text = text


class FakeDataset:
    def __init__(self, batches):
        self.batches, self.i = batches, 0

    def next_batch(self, name):
        b = self.batches[self.i]
        self.i += 1
        return FakeBatch(b)

    def idx2sentence(self, s):
        return ""


def train_vectors(name, seed, B, n_iter, clip, z_regu, **kw):
    out = {}
    model = build(seed, **kw)
    out.update({"w0." + k[2:]: v for k, v in np_state(model).items()})
    gen = torch.Generator().manual_seed(seed + 5)
    batches = [synth_ids(B, T, V, gen) for _ in range(n_iter + 1)]
    out["batches"] = torch.stack(batches).numpy()
    cfgv = rcfg.Bunch(
        lr=1e-3, s_iter=0, n_iter=n_iter,
        beta=rcfg.Bunch(start=rcfg.Bunch(val=1.0, iter=0), end=rcfg.Bunch(val=2.0, iter=4)),
        lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3, z_regu_loss=z_regu,
        cheaplog_every=10 ** 9, expsvlog_every=10 ** 9, clip_grad=clip, chkpt_path="/tmp/x_{}.pt")
    # it=0 satisfies it % every == 0 -> the reference logs and samples one sentence at it 0
    logged = {}
    rtrain.log_value = lambda k, v, it: logged.setdefault(it, {}).__setitem__(k, float(v))
    reset_rf()
    torch.manual_seed(seed + 7)
    np.random.seed(seed + 7)
    cap = Capture()
    hook = DropHook(model.decoder.fc[0])
    snaps = {}

    # snapshot params after iteration k by wrapping dataset.next_batch (called at the top of each iter)
    ds = FakeDataset(batches)
    orig_next = ds.next_batch

    def next_batch(nm):
        if ds.i in (1, n_iter):
            snaps[ds.i] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return orig_next(nm)
    ds.next_batch = next_batch
    with cap.on():
        rtrain.train_vae(cfgv, model, ds)
    hook.remove()
    final = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # per-iteration randomness in consumption order (train_vae.py:24-30; it=0 also draws a logging sample)
    n_total = n_iter + 1
    eps, cs, wds, zpf, zpr = [], [], [], [], []
    rfw = rfb = None
    outmasks = []
    hi = 0
    for it in range(n_total):
        eps.append(cap.take('randn').numpy())
        cs.append(cap.take('multinomial').astype(np.float32))
        wds.append(cap.take('binomial').astype(np.uint8))
        outmasks.append(hook.masks[hi].numpy()); hi += 1
        zpf.append(cap.take('randn_like').numpy())
        zpr.append(cap.take('randn_like').numpy())
        if it == 0:
            rfw = cap.take('randn').numpy()
            cap.take('rand')
            rfb = rlosses.rf['gaussian'][1].numpy().copy()
            # logging sample at it=0: sample_z_prior(1) -> randn ; sample_c_prior -> multinomial;
            # forward_sample passes the eval-mode dropout (hook fires, mask of ones)
            cap.take('randn')
            cap.take('multinomial')
            while hi < len(hook.masks) and hook.masks[hi].dim() == 2:
                hi += 1
    out["eps"], out["c"], out["wd_mask"] = np.stack(eps), np.stack(cs), np.stack(wds)
    out["out_mask"] = np.stack(outmasks)
    out["z_prior_full"], out["z_prior_rf"] = np.stack(zpf), np.stack(zpr)
    out["rf_w"], out["rf_b"] = rfw, rfb
    for it, snap in snaps.items():
        for k, v in snap.items():
            if not k.startswith("classifier"):
                out[f"w{it}." + k] = v.numpy()
    for k, v in final.items():
        if not k.startswith("classifier"):
            out[f"w{n_total}." + k] = v.numpy()
    for k, v in logged[0].items():
        out["log0." + k] = np.float64(v)
    out["clip"] = np.float32(clip)
    out["z_regu"] = np.array(z_regu)
    out["beta_end_iter"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, f"train_{name}.npz"), **out)
    print(f"train_{name}.npz: {len(out)} arrays; logged it0 = {logged[0]}")


def trained_vectors(name, seed, B_train, n_iter, **kw):
    """The same vectors as model_vectors, for a model that the REFERENCE's own train_vae.train_vae has trained for n_iter
    iterations first (Adam lr 1e-3, clip 5.0, z_regu 'mmdrf', beta 1 -> 2: cfg.py defaults) on synthetic peptides: gates move
    away from their default-init regime (|W_hh| grows, r / z saturate on the <pad> tail), logit margins open up."""
    model = build(seed, **kw)
    gen = torch.Generator().manual_seed(seed + 5)
    pool = synth_ids(64 * B_train, T, V, gen)

    class PoolDataset:
        def next_batch(self, nm):
            return FakeBatch(pool[torch.randint(0, pool.shape[0], (B_train,), generator=gen)])

        def idx2sentence(self, s):
            return ""
    cfgv = rcfg.Bunch(
        lr=1e-3, s_iter=0, n_iter=n_iter,
        beta=rcfg.Bunch(start=rcfg.Bunch(val=1.0, iter=0), end=rcfg.Bunch(val=2.0, iter=n_iter)),
        lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3, z_regu_loss='mmdrf',
        cheaplog_every=10 ** 9, expsvlog_every=10 ** 9, clip_grad=5.0, chkpt_path="/tmp/x_{}.pt")
    logged = {}
    rtrain.log_value = lambda k, v, it: logged.setdefault(it, {}).__setitem__(k, float(v))
    reset_rf()
    torch.manual_seed(seed + 7)
    np.random.seed(seed + 7)
    w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rtrain.train_vae(cfgv, model, PoolDataset())
    moved = {k: float((model.state_dict()[k] - w0[k]).abs().max()) for k in ("decoder.rnn.weight_hh_l0", "encoder.rnn.weight_hh_l0")}
    print(f"trained {n_iter} reference iterations; it0 recon {logged[0]['train_L_vae_recon']:.4f}; max |dW_hh| {moved}")
    model_vectors(name, seed, B=16, N_greedy=256, N_beam=16, z_regu_variants=["mmdrf"], model=model)


if __name__ == "__main__" and os.environ.get("CPG_GOLDEN_ONLY") == "trained":
    torch.set_num_threads(4)
    trained_vectors("A_200", 1238, B_train=32, n_iter=200, **model_kwargs(z_dim=100, enc_h=80))


# ------------------------------------------------------------------ CLaSS rejection sampling (G10)
def class_vectors(seed=1238, D=14, K=5, n=4000):
    import density_modeling as rdm
    from sklearn.linear_model import LogisticRegression
    rng = np.random.RandomState(seed)
    N = 600
    mu = torch.from_numpy(rng.randn(N, D) * 0.8)
    logvar = torch.from_numpy(rng.randn(N, D) * 0.1 - 2.0)
    torch.manual_seed(seed)
    np.random.seed(seed)
    Q = rdm.mogQ(mu, logvar, n_components=K, z_num_samples=10, covariance_type='diag')
    clfs = {}
    for a, s in (("amp", 1), ("tox", 2)):
        r2 = np.random.RandomState(seed + s)
        X = mu.numpy()
        w = r2.randn(D)
        y = (X @ w + 0.3 * r2.randn(N) > 0).astype(np.float64)
        clf = LogisticRegression(solver='lbfgs', max_iter=200)
        clf.fit(X, y)
        clfs[a] = clf
    Q.init_attr_classifiers(clfs, clf_targets={'amp': 1, 'tox': 0})
    np.random.seed(seed + 11)
    samples_z, scores_z, accepted = Q.rejection_sample(n)
    # replay sklearn's GaussianMixture.sample draw order on the same numpy global stream
    np.random.seed(seed + 11)
    rs = np.random.mtrand._rand
    counts = rs.multinomial(n, Q.mog.weights_)
    normals = np.concatenate([rs.standard_normal(size=(int(cn), D)) for cn in counts], 0)
    uniforms = np.random.uniform(size=n)
    comp = np.repeat(np.arange(K), counts)
    z64 = Q.mog.means_[comp] + normals * np.sqrt(Q.mog.covariances_[comp])
    assert np.array_equal(z64.astype(np.float32), samples_z.numpy()), "GMM replay mismatch"
    out = dict(
        gmm_weights=Q.mog.weights_, gmm_means=Q.mog.means_, gmm_covars=Q.mog.covariances_,
        counts=counts.astype(np.int64), normals=normals, uniforms=uniforms,
        z=samples_z.numpy(), accepted=accepted,
        prob_accum=scores_z['clfZ_prob_accum'], prob_amp=scores_z['clfZ_amp=1'], prob_tox=scores_z['clfZ_tox=0'],
        amp_coef=clfs['amp'].coef_, amp_intercept=clfs['amp'].intercept_,
        tox_coef=clfs['tox'].coef_, tox_intercept=clfs['tox'].intercept_)
    np.savez_compressed(os.path.join(OUT, "class_small.npz"), **out)
    print("class_small.npz: accept rate", accepted.mean())


if __name__ == "__main__" and not os.environ.get("CPG_GOLDEN_ONLY"):
    torch.set_num_threads(4)
    # config A: reference defaults (cfg.py:262-274): z=100, enc h=80 biGRU 1 layer, emb 150
    model_vectors("A", 1238, B=16, N_greedy=256, N_beam=64, z_regu_variants=["mmdrf"],
                  **model_kwargs(z_dim=100, enc_h=80))
    # micro: He=16, Z=14 (Hd=16), small emb; all three regularisers' gradients
    model_vectors("micro", 77, B=7, N_greedy=64, N_beam=32, z_regu_variants=["mmdrf", "kl", "mmd"],
                  **model_kwargs(z_dim=14, enc_h=16, emb_dim=12))
    # 2-layer encoder (encoder.py:27,46-47: uses top layer's fwd/bwd final states)
    model_vectors("enc2", 99, B=5, N_greedy=16, N_beam=8, z_regu_variants=["mmdrf"],
                  **model_kwargs(z_dim=22, enc_h=24, enc_layers=2, emb_dim=20))
    train_vectors("micro_clip", 77, B=6, n_iter=4, clip=0.1, z_regu="mmdrf",
                  **model_kwargs(z_dim=14, enc_h=16, emb_dim=12))
    train_vectors("micro_noclip", 78, B=6, n_iter=4, clip=1e9, z_regu="kl",
                  **model_kwargs(z_dim=14, enc_h=16, emb_dim=12))
    train_vectors("A_clip", 1238, B=8, n_iter=1, clip=0.25, z_regu="mmdrf",
                  **model_kwargs(z_dim=100, enc_h=80))
    class_vectors()


# ------------------------------------------------------------------ CNN classifier logits (G11) - separate small fixture
def classifier_vectors(name="A", seed=1238, B=16, **kw):
    model = build(seed, **kw)
    model.eval()
    gen = torch.Generator().manual_seed(seed)
    ids = synth_ids(B, T, V, gen)
    with torch.no_grad():
        logits = model.forward_classifier(ids)
        np.random.seed(seed + 41)
        cap = Capture()
        with cap.on():
            (mu, logvar), (z, c), dec = model(ids, q_c='classifier', sample_z='max')
    out = {"w." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()
           if k.startswith("classifier") or k == "word_emb.weight"}
    out.update(ids=ids.numpy(), logits=logits.numpy(), c_softmax=c.numpy())
    # ---- gradients (the reference never trains the classifier, SURVEY F11, but its autograd defines them):
    # (i) of sum(logits * gl) wrt the classifier's parameters and the embedding table, eval mode (no dropout);
    gen2 = torch.Generator().manual_seed(seed + 43)
    gl = torch.randn(B, 2, generator=gen2)
    model.zero_grad()
    (model.forward_classifier(ids) * gl).sum().backward()
    out["gl"] = gl.numpy()
    for k, prm in model.named_parameters():
        if (k.startswith("classifier") or k == "word_emb.weight") and prm.grad is not None:
            out["gcls." + k] = prm.grad.numpy().copy()
    # (ii) of the reconstruction loss through c = softmax(classifier(x)) (models/model.py:186-188, q_c='classifier'), z = mu,
    # eval mode, the word-dropout mask captured: what reaches the classifier's parameters comes through c alone
    model.zero_grad()
    np.random.seed(seed + 47)
    cap = Capture()
    with cap.on():
        (mu, logvar), (z, c), dec = model(ids, q_c='classifier', sample_z='max')
    loss = rlosses.recon_dec(ids, dec)
    loss.backward()
    out["qc.wd_mask"] = cap.take('binomial').astype(np.uint8)
    out["qc.loss"] = np.float32(loss.item())
    for k, prm in model.named_parameters():
        if k.startswith("classifier") and prm.grad is not None:
            out["gqc." + k] = prm.grad.numpy().copy()
    out["gqc.word_emb.weight"] = model.word_emb.weight.grad.numpy().copy()
    out.update({"wfull." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, f"classifier_{name}.npz"), **out)
    print(f"classifier_{name}.npz written")


if __name__ == "__main__" and os.environ.get("CPG_GOLDEN_ONLY") == "classifier":
    classifier_vectors("A", 1238, **model_kwargs(z_dim=100, enc_h=80))


def soft_vectors(name="A", seed=1238, N=48, **kw):
    """RNN_VAE.sample_G soft modes (models/model.py:337-359): none_softmax and greedy_softmax are deterministic given (z,c);
    categorical_softmax is captured with the sampled indices so the soft outputs can be replayed."""
    model = build(seed, **kw)
    with torch.no_grad():
        model.decoder.fc[1].weight.mul_(6.0)   # spread the logits so rows emit <eos> at varied steps
        model.decoder.fc[1].bias[3] += 1.0
    gen = torch.Generator().manual_seed(seed + 5)
    z = torch.randn(N, model.z_dim, generator=gen)
    c = torch.zeros(N, 2)
    c[torch.arange(N), torch.randint(0, 2, (N,), generator=gen)] = 1
    out = {"w." + k: v.detach().numpy().copy() for k, v in model.state_dict().items() if not k.startswith("classifier")}
    out.update(z=z.numpy(), c=c.numpy())
    for mode, temp in (("none_softmax", 1.0), ("greedy_softmax", 1.0), ("greedy_softmax", 0.7)):
        (ids, soft), _, _ = model.generate_sentences(N, z, c, sample_mode=mode, temp=temp)
        tag = f"{mode}_t{temp}"
        out[tag + ".ids"] = ids.numpy()
        out[tag + ".soft"] = soft.detach().numpy()
    torch.manual_seed(seed + 9)
    (ids, soft), _, _ = model.generate_sentences(N, z, c, sample_mode="categorical_softmax", temp=0.9)
    out["categorical_softmax_t0.9.ids"] = ids.numpy()
    out["categorical_softmax_t0.9.soft"] = soft.detach().numpy()
    np.savez_compressed(os.path.join(OUT, f"soft_{name}.npz"), **out)
    print(f"soft_{name}.npz written", {k: v.shape for k, v in out.items() if not k.startswith("w.")})


if __name__ == "__main__" and os.environ.get("CPG_GOLDEN_ONLY") == "soft":
    soft_vectors("A", 1238, **model_kwargs(z_dim=100, enc_h=80))


def categorical_vectors(name="A", seed=1238, N=96, **kw):
    """RNN_VAE.sample_G 'categorical' (models/model.py:308-309): the reference's draws are captured through torch.multinomial
    (what Categorical.sample calls) and turned into the uniforms an inverse-CDF sampler needs to reproduce them - the midpoint
    of the drawn token's cumulative-probability interval - so the ids below are the REFERENCE's ids for replayable draws."""
    model = build(seed, **kw)
    with torch.no_grad():
        model.decoder.fc[1].weight.mul_(4.0)   # spread the logits a little: varied lengths
        model.decoder.fc[1].bias[3] += 0.5
    gen = torch.Generator().manual_seed(seed + 21)
    z = torch.randn(N, model.z_dim, generator=gen)
    c = torch.zeros(N, 2)
    c[torch.arange(N), torch.randint(0, 2, (N,), generator=gen)] = 1
    out = {"w." + k: v.detach().numpy().copy() for k, v in model.state_dict().items() if not k.startswith("classifier")}
    out.update(z=z.numpy(), c=c.numpy())
    orig = torch.multinomial
    for tag, kwargs in (("t1.0", dict(temp=1.0)), ("t0.7", dict(temp=0.7)), ("t1.0_pe", dict(temp=1.0, prevent_empty=True))):
        rec = []

        def hooked(probs, num, replacement=False, **k2):
            r = orig(probs, num, replacement, **k2)
            rec.append((probs.detach().double().numpy().copy(), r.numpy().reshape(-1).copy()))
            return r
        torch.multinomial = hooked
        try:
            torch.manual_seed(seed + 31)
            ids, _, _ = model.generate_sentences(N, z, c, sample_mode="categorical", **kwargs)
        finally:
            torch.multinomial = orig
        u = np.zeros((25, N))
        u[:] = 0.5
        for t, (p, k) in enumerate(rec):
            cum = np.cumsum(p, 1) / p.sum(1, keepdims=True)
            lo = np.where(k > 0, cum[np.arange(N), np.maximum(k - 1, 0)], 0.0)
            u[t] = 0.5 * (lo + cum[np.arange(N), k])
        out[tag + ".ids"] = ids.numpy()
        out[tag + ".u"] = u
    np.savez_compressed(os.path.join(OUT, f"categorical_{name}.npz"), **out)
    print(f"categorical_{name}.npz written", {k: v.shape for k, v in out.items() if not k.startswith("w.")})


if __name__ == "__main__" and os.environ.get("CPG_GOLDEN_ONLY") == "categorical":
    categorical_vectors("A", 1238, **model_kwargs(z_dim=100, enc_h=80))


def mmd_kernel_vectors(seed=1238, N=48, D=100, sigma=7.0):
    """losses.mmd_full_kernel with each of compute_mmd_kernel's three kernels (losses.py:47-56,96-108): loss and d loss/d z1
    from the reference's autograd, on an encoder-like z1 (shifted, narrower) against a N(0,I) prior sample."""
    gen = torch.Generator().manual_seed(seed + 41)
    z1 = (0.6 * torch.randn(N, D, generator=gen) + 0.3).requires_grad_(True)
    z2 = torch.randn(N, D, generator=gen)
    out = dict(z1=z1.detach().numpy().copy(), z2=z2.numpy().copy(), sigma=np.float32(sigma))
    for kernel in ("gaussian", "laplace", "energy"):
        z1.grad = None
        loss = rlosses.mmd_full_kernel(z1, z2, sigma=sigma, kernel=kernel)
        loss.backward()
        out[kernel + ".loss"] = np.float32(loss.item())
        out[kernel + ".dz1"] = z1.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "mmd_kernels.npz"), **out)
    print("mmd_kernels.npz written", {k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})


if __name__ == "__main__" and os.environ.get("CPG_GOLDEN_ONLY") == "mmd_kernels":
    mmd_kernel_vectors()


# ------------------------------------------------------------------ option branches inside hot-path functions (round 4)
def encoder_dropout_vectors(name="encdrop", seed=313, B=6, p=0.25, **kw):
    """GRUEncoder with layers > 1 and p_dropout > 0 (models/encoder.py:25-30: nn.GRU(dropout=p_dropout) drops the output of every
    layer but the last, in train mode).  The mask is drawn INSIDE ATen (at::dropout on the time-major [T,B,2*He] layer output, one
    bernoulli_(1-p) fill from torch's CPU generator) - it is recovered here by replaying that draw from the same seed, and the
    replay is VERIFIED: the reference's (mu, logvar) must equal those of the same weights run layer by layer with the replayed
    mask applied.  Stored: weights, ids, the keep mask [B,T,2*He], (mu, logvar) in train mode, the reference's gradients of
    sum(mu * gmu + logvar * glv) wrt every encoder parameter and the embedding; and the eval-mode (mu, logvar) (no dropout)."""
    model = build(seed, **kw)
    enc = model.encoder
    assert enc.rnn.num_layers == 2 and enc.rnn.dropout == p
    gen = torch.Generator().manual_seed(seed)
    ids = synth_ids(B, T, V, gen)
    He = enc.rnn.hidden_size
    out = {k: v for k, v in np_state(model).items() if not k.startswith("w.classifier")}
    model.train()
    model.zero_grad()
    torch.manual_seed(seed + 3)
    mu, logvar = model.forward_encoder(ids)                      # first torch-generator draw of the call = the dropout noise
    torch.manual_seed(seed + 3)
    keep_tm = torch.empty(T, B, 2 * He).bernoulli_(1 - p)        # the replay (time-major, as ATen sees the layer output)
    # verify the replay: same weights, layer by layer
    sd = enc.rnn.state_dict()
    l0 = torch.nn.GRU(model.emb_dim, He, bidirectional=True, batch_first=True)
    l1 = torch.nn.GRU(2 * He, He, bidirectional=True, batch_first=True)
    l0.load_state_dict({k: v for k, v in sd.items() if k.endswith("_l0") or k.endswith("_l0_reverse")})
    l1.load_state_dict({k.replace("_l1", "_l0"): v for k, v in sd.items() if "_l1" in k})
    with torch.no_grad():
        o0, _ = l0(model.word_emb(ids))
        _, h1 = l1(o0 * keep_tm.transpose(0, 1) / (1 - p))
        hcat = torch.cat((h1[-2], h1[-1]), 1)
        assert float((enc.q_mu(hcat) - mu).abs().max()) < 1e-6, "dropout-mask replay does not reproduce the reference's encoder"
    g2 = torch.Generator().manual_seed(seed + 5)
    gmu, glv = torch.randn(mu.shape, generator=g2), torch.randn(mu.shape, generator=g2)
    ((mu * gmu).sum() + (logvar * glv).sum()).backward()
    out.update(ids=ids.numpy(), enc_keep=keep_tm.transpose(0, 1).contiguous().to(torch.uint8).numpy(), p=np.float32(p),
               mu_train=mu.detach().numpy(), logvar_train=logvar.detach().numpy(), gmu=gmu.numpy(), glv=glv.numpy())
    for k, prm in model.named_parameters():
        if (k.startswith("encoder") or k == "word_emb.weight") and prm.grad is not None:
            out["g." + k] = prm.grad.numpy().copy()
    model.eval()
    with torch.no_grad():
        mu_e, lv_e = model.forward_encoder(ids)
    out.update(mu_eval=mu_e.numpy(), logvar_eval=lv_e.numpy())
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}.npz written; kept fraction {float(keep_tm.mean()):.3f}")


def sample_train_mode_vectors(name="sample_train", seed=515, N=48, N_beam=12, **kw):
    """RNN_VAE.generate_sentences(..., eval_mode=False) (models/model.py:216-221): the model STAYS in train mode, so the decoder's
    nn.Dropout(p_out) (models/decoder.py:43-45) is live in every forward_sample step of sample_G (:86-109) - greedy and beam.  The
    keep masks are captured with a forward hook, step by step."""
    model = build(seed, **kw)
    with torch.no_grad():
        model.decoder.fc[1].weight.mul_(5.0)   # spread the logits: rows end at varied steps, dropout flips some argmaxes
        model.decoder.fc[1].bias[3] += 1.0
    gen = torch.Generator().manual_seed(seed + 5)
    z = torch.randn(N, model.z_dim, generator=gen)
    c = torch.zeros(N, 2)
    c[torch.arange(N), torch.randint(0, 2, (N,), generator=gen)] = 1
    out = {k: v for k, v in np_state(model).items() if not k.startswith("w.classifier")}
    out.update(z=z.numpy(), c=c.numpy())
    model.train()
    hook = DropHook(model.decoder.fc[0])
    torch.manual_seed(seed + 7)
    with torch.no_grad():
        ids, _, _ = model.generate_sentences(N, z, c, eval_mode=False, sample_mode='greedy')
    assert model.training
    out["greedy_ids"] = ids.numpy()
    out["greedy_keep"] = torch.stack(hook.masks).numpy()                 # [steps, N, H]
    assert out["greedy_keep"].mean() < 0.9, "dropout was not live"
    with torch.no_grad():
        model.eval()
        ids_eval, _, _ = model.generate_sentences(N, z, c, eval_mode=True, sample_mode='greedy')
        model.train()
    out["greedy_ids_eval_mode"] = ids_eval.numpy()
    hook.masks.clear()
    torch.manual_seed(seed + 9)
    with torch.no_grad():
        hyps = model.sample_G(N_beam, z[:N_beam], c[:N_beam], sample_mode='beam', beam_size=5, n_best=3)
    hook.remove()
    out["beam_keep"] = torch.stack(hook.masks).numpy()                   # [steps, 5*N_beam, H], rows beam-major
    arr = np.full((N_beam, 3, T + 2), -1, dtype=np.int64)
    for i, hs in enumerate(hyps):
        for j, hyp in enumerate(hs):
            t_ = [int(t) for t in hyp]
            arr[i, j, :len(t_)] = t_
    out["beam_hyps"] = arr
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}.npz written: greedy {ids.shape}, rows differing from eval mode "
          f"{int((ids.numpy()[:, :min(ids.shape[1], ids_eval.shape[1])] != ids_eval.numpy()[:, :min(ids.shape[1], ids_eval.shape[1])]).any(1).sum())}"
          f"/{N}; beam steps {out['beam_keep'].shape[0]}")


if __name__ == "__main__" and os.environ.get("CPG_GOLDEN_ONLY") == "options":
    torch.set_num_threads(4)
    # decoder skip connections (models/decoder.py:48-51,80-81,103-105): every vector of model_vectors for a model built with them
    model_vectors("skip", 211, B=6, N_greedy=48, N_beam=12, z_regu_variants=["mmdrf"],
                  **model_kwargs(z_dim=30, enc_h=24, emb_dim=20, skip=True))
    encoder_dropout_vectors("encdrop", 313, **model_kwargs(z_dim=22, enc_h=24, enc_layers=2, emb_dim=20, enc_dropout=0.25))
    sample_train_mode_vectors("sample_train", 515, **model_kwargs(z_dim=30, enc_h=24, emb_dim=20))
