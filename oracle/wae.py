"""WAE/VAE forward, losses and hand-derived backward in numpy float32 (oracle; test infrastructure only).

Follows the reference's training step (train_vae.py:24-42):
  RNN_VAE.forward            models/model.py:146-195
  GRUEncoder.forward         models/encoder.py:38-52
  GRUDecoder.forward         models/decoder.py:56-84   (WordDropout :117-133)
  losses.recon_dec           losses.py:18-31
  losses.kl_gaussianprior    losses.py:8-10 ; kl_gaussian_sharedmu losses.py:13-15
  losses.mmd_full_kernel     losses.py:47-56,96-108    (incl. the `H - diag(H)` broadcast quirk)
  losses.mmd_rf              losses.py:59-93
All randomness (eps, c, word-dropout mask, out-dropout keep mask, z_prior, rf_w, rf_b) is an INPUT.
Parameters are a dict keyed by the reference's state-dict names.
"""
import numpy as np

from .gru import gru_seq_fwd, gru_seq_bwd, F32

UNK, PAD, START, EOS = 0, 1, 2, 3  # models/mutils.py:5-8


# ----------------------------------------------------------------------------- encoder
def _enc_layers(P):
    n = 0
    while f"encoder.rnn.weight_ih_l{n}" in P:
        n += 1
    return n


def encoder_fwd(P, ids, enc_keep=None, p_drop=0.0):
    """ids [B,T] int -> mu, logvar [B,Z]; cache for backward.  Steps through ALL T positions incl. pads (F4).
    enc_keep (optional, 0/1 [B,T,2*He]) with p_drop: nn.GRU(dropout=p_dropout) in train mode (models/encoder.py:25-30) - the
    concatenated output of every layer but the last is multiplied by keep / (1 - p) before it feeds the next layer."""
    x = P["word_emb.weight"][ids]  # [B,T,E]
    B, T, _ = x.shape
    L = _enc_layers(P)
    cache = {"ids": ids, "layers": [], "drop": None}
    if enc_keep is not None and p_drop > 0 and L > 1:
        cache["drop"] = (enc_keep.astype(F32) * F32(1.0 / (1.0 - p_drop))).astype(F32)
    for l in range(L):
        outs, lc = [], {"x": x}
        for sfx, rev in (("", False), ("_reverse", True)):
            w_ih, w_hh = P[f"encoder.rnn.weight_ih_l{l}{sfx}"], P[f"encoder.rnn.weight_hh_l{l}{sfx}"]
            b_ih, b_hh = P[f"encoder.rnn.bias_ih_l{l}{sfx}"], P[f"encoder.rnn.bias_hh_l{l}{sfx}"]
            H = w_hh.shape[1]
            gi = (x.reshape(B * T, -1) @ w_ih.T + b_ih).reshape(B, T, 3 * H).astype(F32)
            hs, h_last, cs = gru_seq_fwd(gi, np.zeros((B, H), F32), w_hh, b_hh, reverse=rev)
            outs.append((hs, h_last))
            lc[sfx] = cs
        cache["layers"].append(lc)
        x = np.concatenate([outs[0][0], outs[1][0]], 2)
        if cache["drop"] is not None and l < L - 1:
            x = (x * cache["drop"]).astype(F32)
    h = np.concatenate([outs[0][1], outs[1][1]], 1)  # top layer fwd/bwd final states (encoder.py:46-47)
    mu = (h @ P["encoder.q_mu.weight"].T + P["encoder.q_mu.bias"]).astype(F32)
    logvar = (h @ P["encoder.q_logvar.weight"].T + P["encoder.q_logvar.bias"]).astype(F32)
    cache["h"] = h
    return mu, logvar, cache


def encoder_bwd(P, dmu, dlogvar, cache, G):
    h = cache["h"]
    G["encoder.q_mu.weight"] = (dmu.T @ h).astype(F32)
    G["encoder.q_mu.bias"] = dmu.sum(0).astype(F32)
    G["encoder.q_logvar.weight"] = (dlogvar.T @ h).astype(F32)
    G["encoder.q_logvar.bias"] = dlogvar.sum(0).astype(F32)
    dh = (dmu @ P["encoder.q_mu.weight"] + dlogvar @ P["encoder.q_logvar.weight"]).astype(F32)
    L = len(cache["layers"])
    He = dh.shape[1] // 2
    ids = cache["ids"]
    B, T = ids.shape
    dlast = {"": dh[:, :He], "_reverse": dh[:, He:]}
    dout = np.zeros((B, T, 2 * He), F32)  # gradient wrt the layer's concatenated output sequence
    for l in range(L - 1, -1, -1):
        lc = cache["layers"][l]
        x = lc["x"]
        dx = np.zeros_like(x)
        for di, (sfx, rev) in enumerate((("", False), ("_reverse", True))):
            w_ih, w_hh = P[f"encoder.rnn.weight_ih_l{l}{sfx}"], P[f"encoder.rnn.weight_hh_l{l}{sfx}"]
            dhs = dout[:, :, di * He:(di + 1) * He]
            dgi, _, dW_hh, db_hh = gru_seq_bwd(dhs, dlast[sfx] if l == L - 1 else None, lc[sfx], w_hh, reverse=rev)
            G[f"encoder.rnn.weight_hh_l{l}{sfx}"] = dW_hh
            G[f"encoder.rnn.bias_hh_l{l}{sfx}"] = db_hh
            flat = dgi.reshape(B * T, -1)
            G[f"encoder.rnn.weight_ih_l{l}{sfx}"] = (flat.T @ x.reshape(B * T, -1)).astype(F32)
            G[f"encoder.rnn.bias_ih_l{l}{sfx}"] = flat.sum(0).astype(F32)
            dx += (flat @ w_ih).reshape(x.shape)
        dout = dx
        if cache.get("drop") is not None and l > 0:
            dout = (dout * cache["drop"]).astype(F32)   # through the dropout in front of layer l
    # embedding gradient (padding_idx row gets none: nn.Embedding(..., PAD_IDX), models/model.py:47)
    demb = np.zeros_like(P["word_emb.weight"])
    np.add.at(demb, ids.reshape(-1), dout.reshape(B * T, -1))
    demb[PAD] = 0
    return demb


# ----------------------------------------------------------------------------- decoder
def word_dropout(ids, wd_mask):
    """WordDropout.forward (decoder.py:117-133): masked positions become <unk>; no exemption for START/PAD."""
    out = ids.copy()
    out[wd_mask.astype(bool)] = UNK
    return out


def decoder_fwd(P, ids, z, c, wd_mask, out_keep, p_out):
    """Teacher-forced GRU decoder -> logits [B,T,V]."""
    B, T = ids.shape
    tok = word_dropout(ids, wd_mask)
    zc = np.concatenate([z, c], 1).astype(F32)  # init_hidden (decoder.py:53-54) and per-step input tail
    emb = P["word_emb.weight"][tok]
    x = np.concatenate([emb, np.broadcast_to(zc[:, None, :], (B, T, zc.shape[1]))], 2).astype(F32)
    w_ih, w_hh = P["decoder.rnn.weight_ih_l0"], P["decoder.rnn.weight_hh_l0"]
    b_ih, b_hh = P["decoder.rnn.bias_ih_l0"], P["decoder.rnn.bias_hh_l0"]
    H = w_hh.shape[1]
    gi = (x.reshape(B * T, -1) @ w_ih.T + b_ih).reshape(B, T, 3 * H).astype(F32)
    hs, _, cs = gru_seq_fwd(gi, zc, w_hh, b_hh)
    rnn_out = hs
    if "decoder.skip_weight_x.weight" in P:
        # skip connections (models/decoder.py:48-51,80-81): rnn_out := skip_weight_x(rnn_out) + skip_weight_z([z;c]), bias-free
        sx = (hs.reshape(B * T, H) @ P["decoder.skip_weight_x.weight"].T).reshape(B, T, H).astype(F32)
        sz = (zc @ P["decoder.skip_weight_z.weight"].T).astype(F32)
        rnn_out = (sx + sz[:, None, :]).astype(F32)
    scale = F32(1.0 / (1.0 - p_out)) if p_out > 0 else F32(1.0)
    hd = (rnn_out * (out_keep.astype(F32) * scale)).astype(F32)
    logits = (hd.reshape(B * T, H) @ P["decoder.fc.1.weight"].T + P["decoder.fc.1.bias"]).reshape(B, T, -1)
    cache = dict(tok=tok, x=x, cs=cs, hd=hd, keep=out_keep.astype(F32) * scale, zc=zc, hs=hs)
    return logits.astype(F32), cache


def decoder_bwd(P, dlogits, cache, G, E):
    B, T, V = dlogits.shape
    hd, x = cache["hd"], cache["x"]
    H = hd.shape[2]
    dl = dlogits.reshape(B * T, V)
    G["decoder.fc.1.weight"] = (dl.T @ hd.reshape(B * T, H)).astype(F32)
    G["decoder.fc.1.bias"] = dl.sum(0).astype(F32)
    dhs = ((dl @ P["decoder.fc.1.weight"]).reshape(B, T, H) * cache["keep"]).astype(F32)
    dzc_skip = 0.0
    if "decoder.skip_weight_x.weight" in P:
        d2 = dhs.reshape(B * T, H)
        G["decoder.skip_weight_x.weight"] = (d2.T @ cache["hs"].reshape(B * T, H)).astype(F32)
        dsz = dhs.sum(1).astype(F32)
        G["decoder.skip_weight_z.weight"] = (dsz.T @ cache["zc"]).astype(F32)
        dzc_skip = (dsz @ P["decoder.skip_weight_z.weight"]).astype(F32)
        dhs = (d2 @ P["decoder.skip_weight_x.weight"]).reshape(B, T, H).astype(F32)
    w_ih, w_hh = P["decoder.rnn.weight_ih_l0"], P["decoder.rnn.weight_hh_l0"]
    dgi, dh0, dW_hh, db_hh = gru_seq_bwd(dhs, None, cache["cs"], w_hh)
    G["decoder.rnn.weight_hh_l0"], G["decoder.rnn.bias_hh_l0"] = dW_hh, db_hh
    flat = dgi.reshape(B * T, -1)
    G["decoder.rnn.weight_ih_l0"] = (flat.T @ x.reshape(B * T, -1)).astype(F32)
    G["decoder.rnn.bias_ih_l0"] = flat.sum(0).astype(F32)
    dx = (flat @ w_ih).reshape(B, T, -1)
    dzc = (dx[:, :, E:].sum(1) + dh0 + dzc_skip).astype(F32)
    demb = np.zeros_like(P["word_emb.weight"])
    np.add.at(demb, cache["tok"].reshape(-1), dx[:, :, :E].reshape(B * T, E))
    demb[PAD] = 0
    return dzc, demb


# ----------------------------------------------------------------------------- losses
def recon_dec(ids, logits, dp=None):
    """Mean NLL over non-PAD next-token targets of the whole batch.  Returns loss, dlogits.
    dp = (allreduce_sum, world): data-parallel form - the count is the GLOBAL count / world, so that the later
    SUM-all-reduce / world of the gradients equals the single-device gradient."""
    B, T, V = logits.shape
    tgt = np.concatenate([ids[:, 1:], np.full((B, 1), PAD, ids.dtype)], 1).reshape(-1)
    lg = logits.reshape(B * T, V).astype(F32)
    m = lg.max(1, keepdims=True)
    lse = m + np.log(np.exp(lg - m).sum(1, keepdims=True))
    logp = lg - lse
    valid = tgt != PAD
    cnt = max(int(valid.sum()), 1)
    if dp is not None:
        cnt = max(float(dp[0](np.array([float(valid.sum())]))[0]) / dp[1], 1.0)
    nll = -logp[np.arange(B * T), tgt]
    loss = F32(nll[valid].sum() / cnt)
    d = np.exp(logp)
    d[np.arange(B * T), tgt] -= 1.0
    d[~valid] = 0.0
    return loss, (d / cnt).reshape(B, T, V).astype(F32)


def kl_gaussianprior(mu, lv):
    B = mu.shape[0]
    loss = F32(np.mean(0.5 * np.sum(np.exp(lv) + mu * mu - 1.0 - lv, 1)))
    return loss, (mu / B).astype(F32), (0.5 * (np.exp(lv) - 1.0) / B).astype(F32)


def kl_gaussian_sharedmu(mu, lv):
    B = mu.shape[0]
    loss = F32(np.mean(0.5 * np.sum(np.exp(lv) - 1.0 - lv, 1)))
    return loss, (0.5 * (np.exp(lv) - 1.0) / B).astype(F32)


def logvar_l1(lv):
    B = lv.shape[0]
    return F32(np.abs(lv).sum(1).mean(0)), (np.sign(lv) / B).astype(F32)


SQDIST_BROADCAST_LIMIT = 1 << 26  # elements of the [N,M,D] float64 difference tensor the reference's form may take


def _sqdist(x, y, form=None):
    """|x_i - y_j|^2 in float64.  The reference builds the [N,M,D] difference tensor (losses.py:99-103); that form is kept
    for every size the golden vectors use.  Above SQDIST_BROADCAST_LIMIT elements (config B: 2048 x 2048 x 510 doubles =
    17 GB) the Gram form |x|^2 + |y|^2 - 2 x.y^T is evaluated instead, also in float64: the two agree to ~1e-12 relative
    (tests/test_oracle_golden.py::test_sqdist_forms_agree), far below the 1e-4 bars."""
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    if form is None:
        form = "broadcast" if xd.shape[0] * yd.shape[0] * xd.shape[1] <= SQDIST_BROADCAST_LIMIT else "gram"
    if form == "broadcast":
        return ((xd[:, None, :] - yd[None, :, :]) ** 2).sum(2)
    d = (xd * xd).sum(1)[:, None] + (yd * yd).sum(1)[None, :] - 2.0 * (xd @ yd.T)
    return np.maximum(d, 0.0)


def _mmd_kernel(d, sigma, kernel):
    """compute_mmd_kernel (losses.py:96-108) on squared distances d -> K, dK/dd."""
    s2 = float(sigma) ** 2
    if kernel == "gaussian":
        K = np.exp(-d / s2)
        return K, -K / s2
    if kernel == "laplace":
        r = np.sqrt(d + s2)
        K = np.exp(-r)
        return K, -K / (2.0 * r)
    if kernel == "energy":
        return (d + s2) ** -0.25, -0.25 * (d + s2) ** -1.25
    raise ValueError(kernel)


def mmd_full_kernel(z1, z2, sigma, kernel="gaussian"):
    """H = K11+K22-2K12 with K from compute_mmd_kernel; then `H - diag(H)` BROADCASTS the diagonal vector over rows
    (every column j loses H_jj in every row): loss = (sum H - N*sum_j H_jj)/(N(N-1)).  -> loss, d loss / d z1."""
    N = z1.shape[0]
    (K11, W11), (K22, _), (K12, W12) = (_mmd_kernel(_sqdist(a, b), sigma, kernel) for a, b in ((z1, z1), (z2, z2), (z1, z2)))
    Hm = K11 + K22 - 2.0 * K12
    loss = (Hm.sum() - N * np.trace(Hm)) / (N * (N - 1))
    # d loss / d z1.  Coefficient on each H_ij: (1 - N*[i==j]) / (N(N-1)); d d_ij / d x_i = 2 (x_i - y_j)
    coef = (np.ones((N, N)) - N * np.eye(N)) / (N * (N - 1))
    z1d, z2d = z1.astype(np.float64), z2.astype(np.float64)
    # K11_ij depends on z1_i and z1_j ; K12_ij on z1_i only
    A = coef * W11
    A = A + A.T
    g = 2.0 * (A.sum(1)[:, None] * z1d - A @ z1d)
    Bm = -2.0 * coef * W12
    g += 2.0 * (Bm.sum(1)[:, None] * z1d - Bm @ z2d)
    return F32(loss), g.astype(F32)


def gaussian_rf(z, rf_w, rf_b, sigma):
    R = rf_w.shape[1]
    pre = (z @ rf_w) / F32(sigma) + rf_b
    return (np.cos(pre) * F32((2.0 / R) ** 0.5)).astype(F32), pre


def mmd_rf(z1, z2, rf_w, rf_b, sigma, dp=None):
    R = rf_w.shape[1]
    f1, pre1 = gaussian_rf(z1, rf_w, rf_b, sigma)
    f2, _ = gaussian_rf(z2, rf_w, rf_b, sigma)
    B = z1.shape[0]
    if dp is None:
        diff = f1.mean(0) - f2.mean(0)
    else:  # global feature means; every rank differentiates the global loss wrt its own rows (pre-scaled by world)
        s1, s2 = dp[0](f1.sum(0).astype(np.float64)), dp[0](f2.sum(0).astype(np.float64))
        diff = ((s1 - s2) / (B * dp[1])).astype(F32)
    loss = F32((diff * diff).sum())
    dpre = (-np.sin(pre1) * F32((2.0 / R) ** 0.5)) * (2.0 * diff / B)[None, :]
    dz1 = (dpre @ rf_w.T) / F32(sigma)
    return loss, dz1.astype(F32)


# ----------------------------------------------------------------------------- full training-loss evaluation
def train_loss_and_grads(P, ids, rnd, beta, lam_l1, lam_kl, z_regu, sigma=7.0, p_out=0.3, dp=None):
    """One train_vae loss evaluation + backward (train_vae.py:26-40).
    rnd: dict with eps, c, wd_mask, out_mask, z_prior_full, z_prior_rf, rf_w, rf_b.
    Returns (terms dict, grads dict keyed like the state dict, aux dict with z/logits/mu/logvar)."""
    E = P["word_emb.weight"].shape[1]
    mu, lv, ec = encoder_fwd(P, ids, rnd.get("enc_keep"), float(rnd.get("enc_p", 0.0)))
    std = np.exp(lv / 2).astype(F32)
    z = (mu + std * rnd["eps"]).astype(F32)
    c = rnd["c"].astype(F32)
    logits, dc = decoder_fwd(P, ids, z, c, rnd["wd_mask"], rnd["out_mask"], p_out)
    recon, dlogits = recon_dec(ids, logits, dp)
    kl, dmu_kl, dlv_kl = kl_gaussianprior(mu, lv)
    mmd, dz_mmd = mmd_full_kernel(z, rnd["z_prior_full"], sigma)
    mmdrf, dz_rf = mmd_rf(z, rnd["z_prior_rf"], rnd["rf_w"], rnd["rf_b"], sigma, dp)
    l1, dlv_l1 = logvar_l1(lv)
    klmu, dlv_klmu = kl_gaussian_sharedmu(mu, lv)
    regu = {"kl": kl, "mmd": mmd, "mmdrf": mmdrf}[z_regu]
    total = F32(recon + beta * regu + lam_l1 * l1 + lam_kl * klmu)
    # backward
    G = {}
    dzc, demb_dec = decoder_bwd(P, dlogits, dc, G, E)
    Z = z.shape[1]
    dz = dzc[:, :Z].copy()
    dmu = np.zeros_like(mu)
    dlv = (lam_l1 * dlv_l1 + lam_kl * dlv_klmu).astype(F32)
    if z_regu == "kl":
        dmu += beta * dmu_kl
        dlv += beta * dlv_kl
    elif z_regu == "mmd":
        dz += beta * dz_mmd
    else:
        dz += beta * dz_rf
    dmu += dz
    dlv += dz * rnd["eps"] * 0.5 * std
    demb_enc = encoder_bwd(P, dmu.astype(F32), dlv.astype(F32), ec, G)
    G["word_emb.weight"] = (demb_enc + demb_dec).astype(F32)
    terms = dict(total=total, recon=recon, kl=kl, mmd=mmd, mmdrf=mmdrf, l1=l1, klmu=klmu)
    aux = dict(mu=mu, logvar=lv, z=z, logits=logits, dz=dz.astype(F32), dlogits=dlogits, enc_h=ec["h"])
    return terms, G, aux
