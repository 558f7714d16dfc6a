"""Device kernels of the CLaSS sampler (density_modeling.py:43-60,79-80): proposal draw and z-space rejection."""
import torch

from .ops import _p, _stream, call


def gmm_sample(means, covars, comp, normals):
    """means/covars f64 [K,D], comp int32 [n] (component of each row, in component order), normals f64 [n,D] -> z f32."""
    n, D = normals.shape
    z = torch.empty(n, D, device=normals.device, dtype=torch.float32)
    call("cpg_gmm_sample", _p(means.contiguous()), _p(covars.contiguous()), _p(comp.contiguous()), _p(normals.contiguous()),
         n, D, _p(z), _stream())
    return z


def lr_score_accept(z, coef, intercept, target, uniforms):
    """z f32 [n,D]; coef f64 [A,D]; intercept f64 [A]; target int32 [A]; uniforms f64 [n]
    -> probs f64 [A,n], accum f64 [n], accepted uint8 [n]."""
    n, D = z.shape
    A = coef.shape[0]
    probs = torch.empty(A, n, device=z.device, dtype=torch.float64)
    accum = torch.empty(n, device=z.device, dtype=torch.float64)
    acc = torch.empty(n, device=z.device, dtype=torch.uint8)
    call("cpg_lr_score_accept", _p(z.contiguous()), n, D, _p(coef.contiguous()), _p(intercept.contiguous()),
         _p(target.contiguous()), A, _p(uniforms.contiguous()), _p(probs), _p(accum), _p(acc), _stream())
    return probs, accum, acc
