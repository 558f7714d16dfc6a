"""Shared helpers for the GPU parity tests (model construction from golden fixtures)."""
import numpy as np
import torch

from conftest import weights_of


def model_kwargs_from_weights(P, enc_dropout=0.0):
    V, E = P["word_emb.weight"].shape
    He = P["encoder.rnn.weight_hh_l0"].shape[1]
    Z = P["encoder.q_mu.weight"].shape[0]
    layers = 0
    while f"encoder.rnn.weight_ih_l{layers}" in P:
        layers += 1
    return V, dict(
        z_dim=Z, c_dim=2, emb_dim=E, pretrained_emb=None, freeze_embeddings=False, flow=0, flow_type='',
        E_args=dict(h_dim=He, biGRU=True, layers=layers, p_dropout=enc_dropout),
        G_args=dict(G_class='gru', GRU_args=dict(p_word_dropout=0.3, p_out_dropout=0.3, skip_connetions="decoder.skip_weight_x.weight" in P),
                    deconv_args=dict()),
        C_args=dict(min_filter_width=3, max_filter_width=5, num_filters=100, dropout=0.5))


def build_model(P, device="cuda", T=25, enc_dropout=0.0):
    from models.model import RNN_VAE
    V, kw = model_kwargs_from_weights(P, enc_dropout)
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


# --------------------------------------------------------------------------------------------- golden checks
# Bodies of the golden-vector parity tests, shared by tests/test_gpu_parity.py (launcher's own tile choice) and
# tests/test_gpu_tiles.py (every compiled tile instantiation forced through the CPG_* knobs).
def set_losses_cfg(sigma=7.0, rf_dim=500):
    import cfg
    cfg.losses.wae_mmd.sigma, cfg.losses.wae_mmd.rf_dim = sigma, rf_dim


def check_encoder_golden(g):
    m = build_model(weights_of(g))
    with torch.no_grad():
        mu, lv = m.forward_encoder(cu(g["ids"]))
    np.testing.assert_allclose(mu.cpu().numpy(), g["enc_mu"], atol=1e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), g["enc_logvar"], atol=1e-5)


def check_decoder_teacher_forced_golden(g):
    m = build_model(weights_of(g))
    ids = cu(g["ids"])
    with torch.no_grad():
        lg = m.forward_decoder(ids, cu(g["z"]), cu(g["c"]), wd_mask=cu(g["wd_mask"]), out_keep=cu(g["out_mask"]))
        np.testing.assert_allclose(lg.cpu().numpy(), g["logits_train"], atol=2e-5)
        m.eval()
        lg = m.forward_decoder(ids, cu(g["z"]), cu(g["c"]), wd_mask=cu(g["wd_mask_eval"]))
        np.testing.assert_allclose(lg.cpu().numpy(), g["logits_eval"], atol=2e-5)
        (mu, lv), (z, c), lg = m(ids, q_c=cu(g["labels"]), sample_z='max', rnd=dict(wd_mask=cu(g["wd_mask_max"])))
        np.testing.assert_allclose(lg.cpu().numpy(), g["logits_max"], atol=2e-5)
        np.testing.assert_allclose(c.cpu().numpy(), g["c_lab"])


def train_loss(m, losses, ids, rnd, g, regu, beta, lam_l1, lam_kl, idx=None):
    pick = (lambda a: a) if idx is None else (lambda a: a[idx])
    (mu, lv), (z, c), logits = m(ids, q_c='prior', sample_z=1, rnd=rnd)
    recon = losses.recon_dec(ids, logits)
    kl = losses.kl_gaussianprior(mu, lv)
    mmd = losses.wae_mmd_gaussianprior(z, method='full_kernel', z_prior=cu(pick(g["z_prior_full"])))
    mmdrf = losses.wae_mmd_gaussianprior(z, method='rf', z_prior=cu(pick(g["z_prior_rf"])))
    l1 = losses.logvar_l1(lv)
    klmu = losses.kl_gaussian_sharedmu(mu, lv)
    regu_v = {'kl': kl, 'mmd': mmd, 'mmdrf': mmdrf}[regu]
    loss = recon + beta * regu_v + lam_l1 * l1 + lam_kl * klmu
    return loss, dict(recon=recon, kl=kl, mmd=mmd, mmdrf=mmdrf, l1=l1, klmu=klmu, z=z, logits=logits, mu=mu, lv=lv)


def check_losses_and_grads_golden(g, ragged=False):
    import losses
    set_losses_cfg()
    variants = [""] + [p for p in ("v1.", "v2.") if p + "regu" in g]
    for p in variants:
        m = build_model(weights_of(g))
        m.decoder.ragged = ragged
        losses.rf.clear()
        losses.rf['gaussian'] = (cu(g["rf_w"]), cu(g["rf_b"]))
        ids = cu(g["ids"])
        loss, t = train_loss(m, losses, ids, rnd_cuda(g), g, str(g[p + "regu"]), float(g["beta"]), float(g["lam_l1"]),
                             float(g["lam_kl"]))
        t["z"].retain_grad()
        t["logits"].retain_grad()
        loss.backward()
        torch.cuda.synchronize()
        assert abs(t["recon"].item() - g["loss_recon"]) < 1e-4
        assert abs(t["kl"].item() - g["loss_kl"]) < 1e-4
        assert abs(t["klmu"].item() - g["loss_klmu"]) < 1e-4
        assert abs(t["l1"].item() - g["loss_l1"]) < 1e-4
        assert abs(t["mmd"].item() - g["loss_mmd_full"]) < 1e-4
        assert abs(t["mmdrf"].item() - g["loss_mmd_rf"]) < 1e-4
        assert abs(loss.item() - g[p + "loss_total"]) < 1e-4
        np.testing.assert_allclose(t["z"].detach().cpu().numpy(), g["z"], atol=1e-5)
        np.testing.assert_allclose(t["logits"].grad.cpu().numpy(), g[p + "g.logits"], atol=1e-6, rtol=1e-3)
        np.testing.assert_allclose(t["z"].grad.cpu().numpy(), g[p + "g.z"], atol=2e-6, rtol=1e-3)
        for k, prm in m.named_parameters():
            if k.startswith("classifier") or k == "decoder.emb.weight":
                continue
            ref = g[p + "g." + k]
            got = prm.grad.cpu().numpy()
            np.testing.assert_allclose(got, ref, atol=2e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=f"{p}{k}")


def check_train_trajectory_golden(g, ragged=False):
    import losses
    from cpg.optim import FusedAdamClip
    set_losses_cfg()
    P0 = {k: v for k, v in weights_of(g, "w0.").items()}
    m = build_model(P0)
    m.decoder.ragged = ragged
    losses.rf.clear()
    losses.rf['gaussian'] = (cu(g["rf_w"]), cu(g["rf_b"]))
    opt = FusedAdamClip(m.vae_params(), lr=1e-3, max_norm=float(g["clip"]))
    n_total = g["batches"].shape[0]
    end_it = int(g["beta_end_iter"])
    regu = str(g["z_regu"])
    for it in range(n_total):
        beta = 1.0 if it <= 0 else (2.0 if it >= end_it else 1.0 + it / end_it)
        ids = cu(g["batches"][it])
        loss, t = train_loss(m, losses, ids, rnd_cuda(g, it), g, regu, beta, 0.0, 1e-3, idx=it)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it == 0:
            assert abs(loss.item() - g["log0.train_L_vae"]) < 1e-4
            assert abs(t["mmd"].item() - g["log0.train_L_wae_mmd"]) < 1e-4
        snap = f"w{it + 1}."
        if snap + "word_emb.weight" in g:
            sd = m.state_dict()
            for k in sd:
                if k.startswith("classifier"):
                    continue
                np.testing.assert_allclose(sd[k].cpu().numpy(), g[snap + k], atol=3e-5, rtol=0, err_msg=f"{snap}{k}")
