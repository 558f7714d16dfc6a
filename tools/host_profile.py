#!/usr/bin/env python3
"""Host-side cost of enqueueing one training step (config B): cProfile over N eager steps with no host synchronisation inside -
where the ~2.5 ms of python / ctypes / autograd time per step goes.   python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import cfg  # noqa: E402
import losses  # noqa: E402
import train_vae as tv  # noqa: E402
from cpg.synth import synth_ids  # noqa: E402
from models.model import RNN_VAE  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
T, V, B, Hh = 25, 24, 2048, 512
torch.manual_seed(1238)
model = RNN_VAE(n_vocab=V, max_seq_len=T, **bench.model_kwargs(Hh - 2, Hh)).to(dev)
model.device = dev
losses.rf.clear()
losses._rf_basis(torch.zeros(1, Hh - 2, device=dev), 500, False)
model.use_device_rng(1238)
losses.set_prior_sampler(lambda z: model._randn(z.shape[0], z.shape[1]))
cfgv = cfg.Bunch(lr=1e-3, clip_grad=5.0, z_regu_loss='mmdrf', lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
                 beta=cfg.Bunch(start=cfg.Bunch(val=1.0, iter=0), end=cfg.Bunch(val=2.0, iter=40000)))
trainer = tv.make_optimizer(cfgv, model, None, 1)
ids = synth_ids(B, T, V, torch.Generator().manual_seed(1)).to(dev)
for it in range(5):
    tv.train_step(cfgv, model, trainer, ids, it)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(N):
    tv.train_step(cfgv, model, trainer, ids, 5 + it)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
