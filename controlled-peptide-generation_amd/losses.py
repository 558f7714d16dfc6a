"""Loss terms of the WAE/VAE training step, computed by the HIP kernels of libcpg_hip.so.

Same names, arguments and return values (differentiable 0-dim tensors) as the reference's losses.py:
  kl_gaussianprior :8-10, kl_gaussian_sharedmu :13-15, recon_dec :18-31, wae_mmd_gaussianprior :34-44,
  mmd_full_kernel :47-56 (with the `H - diag(H)` broadcast quirk), mmd_rf :59-63,
  the process-global random-feature basis `rf` :66-82.
Extras: `z_prior=` can be injected (the reference draws it with torch.randn_like inside), and `logvar_l1` exposes the
penalty train_vae.py:33 computes inline.  Under data parallelism `set_distributed(reduce_fn, world)` makes the batch-global
statistics (non-PAD token count, RFF feature means) equal to the single-device values.
"""
import math

import torch

import cfg  # cfg.losses.wae_mmd is read at call time, like the reference (losses.py:38)
from cpg import ops

rf = {}
_dist = {"reduce": None, "world": 1, "gather": None, "rank": 0}
_prior_sampler = {"fn": None}


def set_distributed(reduce_fn, world, gather_fn=None, rank=0):
    """reduce_fn(tensor) must SUM-all-reduce in place across ranks (e.g. torch.distributed.all_reduce).  gather_fn(tensor
    [b,D]) -> [world*b, D] (rank order, equal shards) is needed only for the full-kernel MMD AS THE REGULARISER under data
    parallelism: that term couples every pair of rows of the global batch, so z is all-gathered (mmd_full_kernel_global)."""
    _dist["reduce"], _dist["world"], _dist["gather"], _dist["rank"] = reduce_fn, int(world), gather_fn, int(rank)


def set_prior_sampler(fn):
    """fn(like_tensor) -> N(0,1) tensor; lets the model's device RNG stream replace torch.randn_like."""
    _prior_sampler["fn"] = fn


def _randn_like(z):
    return _prior_sampler["fn"](z) if _prior_sampler["fn"] is not None else torch.randn_like(z)


def kl_gaussianprior(mu, logvar):
    return ops.LatentTermFn.apply(mu, logvar, 0, mu.size(0))


def kl_gaussian_sharedmu(mu, logvar):
    return ops.LatentTermFn.apply(mu, logvar, 1, mu.size(0))


def logvar_l1(logvar):
    """z_logvar.abs().sum(1).mean(0) (train_vae.py:33)."""
    return ops.LatentTermFn.apply(torch.zeros_like(logvar), logvar, 2, logvar.size(0))


def latent_terms(mu, logvar):
    """(kl_gaussianprior, kl_gaussian_sharedmu, logvar_l1) of one (mu, logvar) pair from a single pass - what train_step uses;
    each equals the stand-alone function's value and gradient."""
    fused = getattr(mu, '_cpg_latent', None)
    if fused is not None and fused[0] is logvar:
        return fused[1], fused[2], fused[3]      # the model's fused latent node (ops.LatentFn) has them: same sums, one backward launch
    return ops.LatentTermsFn.apply(mu, logvar, mu.size(0))


def global_target_count(sequences):
    """Data parallel: the GLOBAL number of non-PAD targets / world as a device scalar (the mean is over the global batch; the gradient
    all-reduce is a SUM / world, so the local term is pre-scaled by world); None on a single rank (the kernel's own count)."""
    if _dist["reduce"] is None:
        return None
    with torch.no_grad():
        B, T = sequences.shape
        tgt = torch.cat([sequences[:, 1:], torch.full((B, 1), ops.PAD_IDX, device=sequences.device)], 1)
        count = (tgt != ops.PAD_IDX).sum().float().reshape(1)
        _dist["reduce"](count)
        return count / _dist["world"]


def recon_dec(sequences, logits):
    fused = getattr(logits, '_cpg_recon', None)
    if fused is not None and fused[0] is sequences:
        return fused[1]      # the decoder ended in ops.VocabReconFn for exactly these targets (train_vae.train_step): its loss
    return ops.ReconLossFn.apply(logits, sequences, global_target_count(sequences))


def wae_mmd_gaussianprior(z, method='full_kernel', z_prior=None, global_batch=False):
    """global_batch (data parallel, full kernel only): evaluate the term on the all-gathered global batch - exact, at the
    price of world^2 times the rank-local Gram work - instead of on this rank's shard (what a logged-only value gets)."""
    if z_prior is None:
        z_prior = _randn_like(z)
    cfgm = cfg.losses.wae_mmd
    if method == 'full_kernel':
        if global_batch and _dist["world"] > 1:
            if _dist["gather"] is None:
                raise RuntimeError("full-kernel MMD over the global batch needs losses.set_distributed(..., gather_fn=, rank=)")
            zg = ops.AllGatherRowsFn.apply(z, _dist["gather"], _dist["rank"], _dist["world"])
            with torch.no_grad():
                zpg = _dist["gather"](z_prior.contiguous())
            return mmd_full_kernel(zg, zpg, sigma=cfgm.sigma, kernel=cfgm.kernel)
        return mmd_full_kernel(z, z_prior, sigma=cfgm.sigma, kernel=cfgm.kernel)
    return mmd_rf(z, z_prior, **cfgm)


def mmd_full_kernel(z1, z2, sigma, kernel='gaussian'):
    if kernel not in ops.MMD_KERNELS:  # the reference falls through with K unbound here (losses.py:102-108)
        raise ValueError('unknown mmd kernel ' + str(kernel))
    assert z1.size(0) == z2.size(0), 'expected matching sizes z1 z2'
    return ops.MmdFullFn.apply(z1, z2, float(sigma), ops.MMD_KERNELS[kernel])


def _rf_basis(z, rf_dim, rf_resample):
    if 'gaussian' not in rf or rf_resample:
        rf_w = torch.randn((z.shape[1], rf_dim), device=z.device)
        rf_b = math.pi * 2 * torch.rand((rf_dim,), device=z.device)
        rf['gaussian'] = (rf_w, rf_b)
    rf_w, rf_b = rf['gaussian']
    assert rf_w.shape == (z.shape[1], rf_dim), 'not expecting z dim or rf_dim to change'
    return rf_w, rf_b


def mmd_rf(z1, z2, sigma, kernel='gaussian', rf_dim=500, rf_resample=False):
    if kernel != 'gaussian':
        raise ValueError('todo implement rf for kernel ' + kernel)
    rf_w, rf_b = _rf_basis(z1, rf_dim, rf_resample)
    b_global = z1.size(0) * _dist["world"]
    return ops.MmdRfFn.apply(z1, z2, rf_w, rf_b, float(sigma), b_global, _dist["reduce"], _dist["world"])
