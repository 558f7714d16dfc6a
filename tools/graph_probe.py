#!/usr/bin/env python3
"""Probe: one training step captured as a HIP graph (torch.cuda.CUDAGraph) and replayed, against eager launches.  The captured
graph freezes the host-side RNG offsets (same dropout masks every replay) - this measures the launch floor only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import cfg  # noqa: E402
import losses  # noqa: E402
import train_vae as tv  # noqa: E402
from cpg import ops  # noqa: E402
from cpg.synth import synth_ids  # noqa: E402
from models.model import RNN_VAE  # noqa: E402

dev = torch.device("cuda")
T, V, B, Hh = 25, 24, 2048, 512
Z = Hh - 2
torch.manual_seed(1238)
model = RNN_VAE(n_vocab=V, max_seq_len=T, **bench.model_kwargs(Z, Hh)).to(dev)
model.device = dev
losses.rf.clear()
losses._rf_basis(torch.zeros(1, Z, device=dev), 500, False)
model.use_device_rng(1238)
losses.set_prior_sampler(lambda z: model._randn(z.shape[0], z.shape[1]))
cfgv = cfg.Bunch(lr=1e-3, clip_grad=5.0, z_regu_loss='mmdrf', lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
                 beta=cfg.Bunch(start=cfg.Bunch(val=1.0, iter=0), end=cfg.Bunch(val=2.0, iter=40000)))
trainer = tv.make_optimizer(cfgv, model, None, 1)
ids = synth_ids(B, T, V, torch.Generator().manual_seed(1)).to(dev)


def step():
    return tv.train_step(cfgv, model, trainer, ids, 10)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("eager: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph replay: %.3f ms/step   loss %.4f" % ((time.perf_counter() - t0) / 20 * 1e3, out["L_vae"].item()))
