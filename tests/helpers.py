"""Shared helpers for the GPU parity tests (model construction from golden fixtures)."""
import numpy as np
import torch

from conftest import weights_of


def model_kwargs_from_weights(P):
    V, E = P["word_emb.weight"].shape
    He = P["encoder.rnn.weight_hh_l0"].shape[1]
    Z = P["encoder.q_mu.weight"].shape[0]
    layers = 0
    while f"encoder.rnn.weight_ih_l{layers}" in P:
        layers += 1
    return V, dict(
        z_dim=Z, c_dim=2, emb_dim=E, pretrained_emb=None, freeze_embeddings=False, flow=0, flow_type='',
        E_args=dict(h_dim=He, biGRU=True, layers=layers, p_dropout=0.0),
        G_args=dict(G_class='gru', GRU_args=dict(p_word_dropout=0.3, p_out_dropout=0.3, skip_connetions=False),
                    deconv_args=dict()),
        C_args=dict(min_filter_width=3, max_filter_width=5, num_filters=100, dropout=0.5))


def build_model(P, device="cuda", T=25):
    from models.model import RNN_VAE
    V, kw = model_kwargs_from_weights(P)
    m = RNN_VAE(n_vocab=V, max_seq_len=T, **kw)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in P.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("classifier") for k in missing), missing
    m = m.to(device)
    m.device = torch.device(device)
    return m


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rnd_cuda(g, idx=None):
    pick = (lambda a: a) if idx is None else (lambda a: a[idx])
    return dict(eps=cu(pick(g["eps"])), c=cu(pick(g["c"])), wd_mask=cu(pick(g["wd_mask"])), out_mask=cu(pick(g["out_mask"])))
