"""Which torch ops (not cpg kernels) launch device work in one headline training step, and from where: torch.profiler with stacks
over 3 steps after warm-up; prints op name, launches per step, and the innermost repo frame.  Used to pick glue to fuse away."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))


def main():
    import bench
    import cfg
    import losses
    import train_vae as tv
    from cpg.synth import synth_ids
    from models.model import RNN_VAE
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda")
    Hh, B, T, V = 512, 2048, 25, 24
    torch.manual_seed(1238)
    model = RNN_VAE(n_vocab=V, max_seq_len=T, **bench.model_kwargs(Hh - 2, Hh)).to(dev)
    model.device = dev
    losses.rf.clear()
    losses._rf_basis(torch.zeros(1, Hh - 2, device=dev), 500, False)
    model.use_device_rng(1238)
    losses.set_prior_sampler(lambda z: model._randn(z.shape[0], z.shape[1]))
    losses.set_distributed(None, 1)
    cfgv = cfg.Bunch(lr=1e-3, clip_grad=5.0, z_regu_loss='mmdrf', lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
                     beta=cfg.Bunch(start=cfg.Bunch(val=1.0, iter=0), end=cfg.Bunch(val=2.0, iter=40000)))
    trainer = tv.make_optimizer(cfgv, model, None, 1)
    g = torch.Generator().manual_seed(1238)
    batches = [synth_ids(B, T, V, g).to(dev) for _ in range(4)]

    def step(x, it):
        return tv.train_step(cfgv, model, trainer, x, it)

    for i in range(5):
        step(batches[i % len(batches)], i)
    torch.cuda.synchronize()
    n = 1
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(batches[0], 5)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
    nodes = [e for e in evs if e.name.startswith("autograd::engine::evaluate_function")]
    out = []
    for ev in evs:
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        t = ev.time_range.start
        node = next((nd.name.split(": ")[-1] for nd in nodes if nd.time_range.start <= t <= nd.time_range.end), "forward")
        out.append((t, ev.name, [tuple(x) for x in (ev.input_shapes or []) if x], node, len(ev.kernels)))
    t_prev_node = None
    for t, name, shapes, node, k in sorted(out):
        print(f"{name:18s} x{k} {str(shapes)[:70]:70s} in {node}")


if __name__ == "__main__":
    main()
