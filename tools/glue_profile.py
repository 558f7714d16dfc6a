#!/usr/bin/env python3
"""Which torch (ATen) ops one training step dispatches beside the library's own kernels, with the repo source line that issued
them (TorchDispatchMode + traceback; ops issued from the autograd engine have no Python frame and are listed by their shapes)."""
import collections
import os
import sys
import traceback

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
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

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
for _ in range(3):
    tv.train_step(cfgv, model, trainer, ids, 10)
torch.cuda.synchronize()

SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.transpose", "aten.slice", "aten.select", "aten.as_strided", "aten.expand",
        "aten.unsqueeze", "aten.squeeze", "aten._unsafe_view", "aten.alias", "aten.permute", "aten.empty", "aten.narrow",
        "aten.reshape", "aten.unbind", "aten.split", "aten.chunk", "aten.is_", "aten.sym_", "aten.stride", "aten.size",
        "aten._local_scalar_dense", "aten.lift_fresh", "aten.new_empty", "aten.resize_", "aten.set_")
log = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            frame = "autograd engine"
            for fr in reversed(traceback.extract_stack()):
                if "controlled-peptide-generation_amd" in fr.filename and "tools/" not in fr.filename:
                    frame = "%s:%d %s" % (fr.filename.split("controlled-peptide-generation_amd/")[-1], fr.lineno, fr.name)
                    break
            shapes = ",".join(str(tuple(a.shape)) for a in args if isinstance(a, torch.Tensor))
            log[(name, frame, shapes)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    tv.train_step(cfgv, model, trainer, ids, 10)
torch.cuda.synchronize()
for (name, frame, shapes), n in sorted(log.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%2d x %-34s %-60s %s" % (n, name, frame, shapes))
print("total dispatched ops:", sum(log.values()))
