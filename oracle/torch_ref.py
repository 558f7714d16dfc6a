"""torch-CPU restatement of the reference's WAE training step (oracle; test infrastructure only).

Why a second restatement next to oracle/wae.py (numpy): SURVEY.md 8(d) asks for the CPU baseline to be timed on the
backend the reference itself runs on - PyTorch's ATen CPU kernels (`nn.GRU`, `F.cross_entropy`, autograd, `optim.Adam`) -
because the reference cannot travel to the GPU box.  This module drives exactly those library calls in the order the
reference does, so its step time on N host threads stands in for `python main.py --phase 1` on CPU; the numpy oracle is
the bit-level checker, this one is the *timed* baseline (bench.py `cpu_baseline`, kind "port").

Follows, with all randomness injectable so it can be pinned to the golden vectors (tests/test_oracle_golden.py):
  RNN_VAE.forward                    models/model.py:146-195    (encoder :96-105, sample_z :107-112)
  GRUEncoder.forward                 models/encoder.py:38-52    (nn.GRU over ALL T positions, cat(h[-2], h[-1]))
  GRUDecoder.forward / WordDropout   models/decoder.py:56-84, 117-133
  losses.recon_dec / kl_* / mmd_*    losses.py:8-108            (incl. the `H - diag(H)` broadcast quirk)
  train_vae loss + optimiser         train_vae.py:15,26-42      (Adam over vae_params() with word_emb listed twice, F6)
Only `tests/`, `__graft_entry__.smoke()` and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

UNK, PAD, START, EOS = 0, 1, 2, 3  # models/mutils.py:5-8


class RefWAE(nn.Module):
    """Same parameter names / shapes as the reference's RNN_VAE (classifier omitted: not part of the training step)."""

    def __init__(self, n_vocab, emb_dim, enc_h, enc_layers, z_dim, c_dim=2, p_out=0.3, cell="gru", skip=False, dec_layers=1):
        """cell='lstm': the build's LSTM EXTENSION (BASELINE.json configs[1] names an LSTM; the reference has none):
        nn.LSTM in place of nn.GRU, encoder read for its final HIDDEN states, decoder h0 = [z;c], c0 = 0.  Parity of that mode is
        pinned to torch.nn.LSTM only - "parity unpinned" against the reference.
        dec_layers > 1: the build's multi-layer decoder EXTENSION (BASELINE.json configs[4] names a "2-layer dec"; the reference
        hard-wires one layer, models/decoder.py:40-41): nn.GRU / nn.LSTM(num_layers=dec_layers), every layer's h0 = [z;c] (the
        reference's `init_h.unsqueeze(0)` repeated per layer), c0 = 0.  Parity unpinned against the reference as well."""
        super().__init__()
        self.cell = cell
        rnn = {"gru": nn.GRU, "lstm": nn.LSTM}[cell]
        self.word_emb = nn.Embedding(n_vocab, emb_dim, PAD)
        self.enc_rnn = rnn(emb_dim, enc_h, num_layers=enc_layers, bidirectional=True, batch_first=True)
        self.q_mu = nn.Linear(2 * enc_h, z_dim)
        self.q_logvar = nn.Linear(2 * enc_h, z_dim)
        hd = z_dim + c_dim
        self.dec_layers = dec_layers
        self.dec_rnn = rnn(emb_dim + hd, hd, num_layers=dec_layers, batch_first=True)
        self.fc = nn.Linear(hd, n_vocab)
        self.skip = skip
        if skip:   # models/decoder.py:48-51
            self.skip_weight_x = nn.Linear(hd, hd, bias=False)
            self.skip_weight_z = nn.Linear(hd, hd, bias=False)
        self.p_out = p_out
        self.z_dim = z_dim

    _MAP = (("encoder.rnn.", "enc_rnn."), ("encoder.q_mu.", "q_mu."), ("encoder.q_logvar.", "q_logvar."),
            ("decoder.rnn.", "dec_rnn."), ("decoder.fc.1.", "fc."), ("decoder.skip_weight_x.", "skip_weight_x."),
            ("decoder.skip_weight_z.", "skip_weight_z."))

    @classmethod
    def from_state(cls, P, p_out=0.3, cell="gru"):
        """P: dict of numpy arrays / tensors keyed by the reference's state-dict names."""
        V, E = P["word_emb.weight"].shape
        He = P["encoder.rnn.weight_hh_l0"].shape[1]
        L = 0
        while f"encoder.rnn.weight_ih_l{L}" in P:
            L += 1
        Z = P["encoder.q_mu.weight"].shape[0]
        Ld = 1
        while f"decoder.rnn.weight_hh_l{Ld}" in P:
            Ld += 1
        m = cls(V, E, He, L, Z, P["decoder.rnn.weight_hh_l0"].shape[1] - Z, p_out, cell, "decoder.skip_weight_x.weight" in P, Ld)
        sd = {}
        for k, v in P.items():
            if k.startswith("classifier") or k == "decoder.emb.weight":
                continue
            for a, b in cls._MAP:
                if k.startswith(a):
                    k = b + k[len(a):]
                    break
            sd[k] = torch.as_tensor(v).clone()
        m.load_state_dict(sd)
        return m

    def ref_name(self, k):
        for a, b in self._MAP:
            if k.startswith(b):
                return a + k[len(b):]
        return k

    def vae_params(self):
        """Order and multiplicity of RNN_VAE.vae_params() (models/model.py:88-94): the shared embedding comes twice."""
        enc = list(self.enc_rnn.parameters()) + list(self.q_mu.parameters()) + list(self.q_logvar.parameters())
        dec = [self.word_emb.weight] + list(self.dec_rnn.parameters()) + list(self.fc.parameters())
        if self.skip:   # decoder.parameters() order: emb, rnn, fc, skip_weight_x, skip_weight_z
            dec += [self.skip_weight_x.weight, self.skip_weight_z.weight]
        return [self.word_emb.weight] + enc + dec

    def forward(self, ids, rnd=None):
        """-> mu, logvar, z, logits.  rnd (optional dict of tensors): eps, c, wd_mask, out_mask; drawn here otherwise, with
        the reference's distributions (numpy's binomial / multinomial are replaced by torch draws: same laws)."""
        rnd = rnd or {}
        B, T = ids.shape
        _, h = self.enc_rnn(self.word_emb(ids))                       # encoder.py:41-42
        if self.cell == "lstm":
            h = h[0]                                                  # (h_n, c_n): the hidden states take nn.GRU's place
        h = torch.cat((h[-2], h[-1]), dim=1)                          # encoder.py:46-47
        mu, logvar = self.q_mu(h), self.q_logvar(h)
        eps = rnd["eps"] if "eps" in rnd else torch.randn(B, self.z_dim)
        z = mu + torch.exp(logvar / 2) * eps                          # model.py:107-112
        c = rnd["c"] if "c" in rnd else F.one_hot(torch.randint(0, 2, (B,)), 2).float()
        wd = rnd["wd_mask"] if "wd_mask" in rnd else (torch.rand(B, T) < 0.3)
        tok = ids.clone()
        tok[wd.bool()] = UNK                                          # decoder.py:117-133 (no exemptions, also in eval)
        zc = torch.cat([z, c], 1)
        x = torch.cat([self.word_emb(tok), zc.unsqueeze(1).expand(-1, T, -1)], 2)   # decoder.py:67-74
        h0 = zc.unsqueeze(0).repeat(self.dec_layers, 1, 1).contiguous()   # decoder.py:77 (h0 = [z;c]; every layer of the extension)
        out, _ = self.dec_rnn(x, h0 if self.cell == "gru" else (h0, torch.zeros_like(h0)))
        if self.skip:                                                 # decoder.py:80-81
            out = self.skip_weight_x(out) + self.skip_weight_z(zc.unsqueeze(1).expand(-1, T, -1))
        if "out_mask" in rnd:
            out = out * rnd["out_mask"].float() / (1.0 - self.p_out)
        else:
            out = F.dropout(out, self.p_out, True)                    # decoder.py:43-45,83
        return mu, logvar, z, self.fc(out)


def recon_dec(ids, logits):
    """losses.py:18-31: mean NLL over the non-PAD next-token targets of the whole batch."""
    B, T, V = logits.shape
    tgt = torch.cat([ids[:, 1:], torch.full((B, 1), PAD, dtype=ids.dtype)], 1)
    return F.cross_entropy(logits.reshape(-1, V), tgt.reshape(-1), reduction="mean", ignore_index=PAD)


def kl_gaussianprior(mu, logvar):
    return torch.mean(0.5 * torch.sum(logvar.exp() + mu ** 2 - 1 - logvar, 1))


def kl_gaussian_sharedmu(mu, logvar):
    return torch.mean(0.5 * torch.sum(logvar.exp() - 1 - logvar, 1))


def _mmd_kernel(x, y, sigma, kernel="gaussian"):
    """losses.py:96-108 with the reference's [N,M,D] broadcast (that tensor is why the step is memory-hungry on CPU)."""
    d = (x.unsqueeze(1) - y.unsqueeze(0)).pow(2).sum(2)
    if kernel == "gaussian":
        return torch.exp(-d / sigma ** 2)
    if kernel == "laplace":
        return torch.exp(-torch.sqrt(d + sigma ** 2))
    if kernel == "energy":
        return torch.pow(d + sigma ** 2, -0.25)
    raise ValueError(kernel)


def mmd_full_kernel(z1, z2, sigma=7.0, kernel="gaussian"):
    """losses.py:47-56: `H - diag(H)` broadcasts the diagonal VECTOR over rows."""
    N = z1.size(0)
    H = _mmd_kernel(z1, z1, sigma, kernel) + _mmd_kernel(z2, z2, sigma, kernel) - 2 * _mmd_kernel(z1, z2, sigma, kernel)
    H = H - torch.diag(H)
    return H.sum() / (N * (N - 1))


def mmd_rf(z1, z2, rf_w, rf_b, sigma=7.0):
    """losses.py:59-93 with a given random-feature basis."""
    R = rf_w.shape[1]

    def feat(z):
        return torch.cos(z @ rf_w / sigma + rf_b) * math.sqrt(2.0 / R)
    return ((feat(z1).mean(0) - feat(z2).mean(0)) ** 2).sum()


def train_loss(m, ids, rnd, beta, lam_l1, lam_kl, z_regu, full_mmd=True, sigma=7.0):
    """train_vae.py:26-37.  full_mmd=False skips the logged-only full-kernel term (it is 91 % of the reference's CPU step at
    B=2048 and does not fit host memory at z=510); it is required when z_regu == 'mmd'."""
    mu, logvar, z, logits = m(ids, rnd)
    terms = {"recon": recon_dec(ids, logits), "kl": kl_gaussianprior(mu, logvar)}
    if full_mmd or z_regu == "mmd":
        zp = rnd["z_prior_full"] if "z_prior_full" in rnd else torch.randn_like(z)
        terms["mmd"] = mmd_full_kernel(z, zp, sigma)
    zp = rnd["z_prior_rf"] if "z_prior_rf" in rnd else torch.randn_like(z)
    terms["mmdrf"] = mmd_rf(z, zp, rnd["rf_w"], rnd["rf_b"], sigma)
    terms["l1"] = logvar.abs().sum(1).mean(0)
    terms["klmu"] = kl_gaussian_sharedmu(mu, logvar)
    terms["total"] = terms["recon"] + beta * terms[z_regu] + lam_l1 * terms["l1"] + lam_kl * terms["klmu"]
    return terms, dict(mu=mu, logvar=logvar, z=z, logits=logits)


class Trainer:
    """Adam(vae_params(), lr) + clip_grad_norm_(vae_params(), clip) exactly as train_vae.py:15,39-42 calls them - the
    duplicate embedding entry goes to both, as in the reference (F6)."""

    def __init__(self, m, lr=1e-3, clip=5.0):
        import warnings
        self.m, self.clip = m, clip
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # "duplicate parameters" - that is the behaviour being restated
            self.opt = torch.optim.Adam(m.vae_params(), lr=lr)

    def step(self, ids, rnd, beta=1.0, lam_l1=0.0, lam_kl=1e-3, z_regu="mmdrf", full_mmd=True):
        terms, _ = train_loss(self.m, ids, rnd, beta, lam_l1, lam_kl, z_regu, full_mmd)
        self.opt.zero_grad()
        terms["total"].backward()
        nn.utils.clip_grad_norm_(self.m.vae_params(), self.clip)
        self.opt.step()
        return terms
