"""The training step replayed from one captured hipGraph (train_vae.GraphedTrainStep) against the same steps run eagerly: the
device-side Philox base, Adam iteration counter and loss weights make every replay a new step - parameters after k steps and the
logged scalars are bit-identical to the eager run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed, H, B, T=25):
    import cfg
    import losses
    import train_vae as tv
    from bench import model_kwargs
    from cpg.synth import synth_ids
    from models.model import RNN_VAE
    dev = torch.device("cuda")
    torch.manual_seed(seed)
    m = RNN_VAE(n_vocab=24, max_seq_len=T, **model_kwargs(H - 2, H)).to(dev)
    m.device = dev
    m.use_device_rng(1234)
    losses.rf.clear()
    torch.manual_seed(99)
    losses._rf_basis(torch.zeros(1, H - 2, device=dev), 500, False)
    losses.set_prior_sampler(lambda z: m._randn(z.shape[0], z.shape[1]))
    losses.set_distributed(None, 1)
    cfgv = cfg.Bunch(lr=1e-3, clip_grad=5.0, z_regu_loss='mmdrf', lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
                     beta=cfg.Bunch(start=cfg.Bunch(val=1.0, iter=0), end=cfg.Bunch(val=2.0, iter=5)))
    tr = tv.make_optimizer(cfgv, m, None, 1)
    g = torch.Generator().manual_seed(seed + 1)
    batches = [synth_ids(B, T, 24, g).to(dev) for _ in range(7)]
    return cfgv, m, tr, batches


@pytest.mark.parametrize("H,B", [(64, 256), (102, 32)])   # persistent forward + direct-to-LDS backward / the per-step kernels
def test_graph_replays_equal_eager_steps(H, B):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    import losses
    import train_vae as tv
    from cpg import ops
    try:
        cfgv, m_e, tr_e, batches = _setup(5, H, B)
        outs_e = [tv.train_step(cfgv, m_e, tr_e, x, it) for it, x in enumerate(batches)]
        loss_e = [float(o["L_vae"].item()) for o in outs_e[-1:]]
        p_e = {k: v.detach().clone() for k, v in m_e.named_parameters()}
        cfgv, m_g, tr_g, batches = _setup(5, H, B)
        step = tv.GraphedTrainStep(cfgv, m_g, tr_g, warmup=2)
        vals = []
        for it, x in enumerate(batches):
            o = step(x, it)
            vals.append(float(o["L_vae"].item()))       # read before the next replay overwrites the static outputs
        assert step.graph is not None and step.calls == len(batches)
        torch.cuda.synchronize()
        ops.check_persistent()
        assert vals[-1] == loss_e[-1], (vals, loss_e)
        assert len(set(vals)) == len(vals)              # every replay saw a new batch / new draws / a new beta
        for k, v in m_g.named_parameters():
            assert torch.equal(v, p_e[k]), k
        assert int(tr_g.iter_dev.item()) == len(batches) == int(tr_e.iter_dev.item())
    finally:
        losses.rf.clear()
        losses.set_prior_sampler(None)


def test_device_rng_base_advances():
    """Two steps' worth of draws with end_step() in between equal one uninterrupted stream: offset + device base = the counter."""
    from cpg import ops
    dev = torch.device("cuda")
    a = ops.DeviceRng(7)
    x1 = a.normal((1000,), dev)
    m1 = a.bernoulli((333,), 0.3, dev)
    a.end_step()
    assert a.offset == 0 and int(a.base.item()) > 0
    x2 = a.normal((1000,), dev)
    b = ops.DeviceRng(7)
    y1, n1, y2 = b.normal((1000,), dev), b.bernoulli((333,), 0.3, dev), b.normal((1000,), dev)
    assert torch.equal(x1, y1) and torch.equal(m1, n1) and torch.equal(x2, y2)
    assert not torch.equal(x1, x2)
