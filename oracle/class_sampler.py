"""CLaSS proposal sampling + z-space classifier rejection (oracle; test infrastructure only).

  mogQ.sample                 density_modeling.py:79-80  -> sklearn GaussianMixture.sample (diag covariance)
  RejSampleBase.score_clf     density_modeling.py:43-48  -> sklearn LogisticRegression.predict_proba[:, target]
  RejSampleBase.rejection_sample  density_modeling.py:50-60  accept = U < prod_attr p_attr(z)
Third-party arithmetic restated (scikit-learn, unpinned in amp_gen.yml:18; fixtures from 1.7.2):
  GaussianMixture.sample (diag): counts ~ multinomial(n, weights); rows for component k (in component order) are
      mean_k + standard_normal * sqrt(cov_k)      computed in float64, then cast to float32 by mogQ.sample.
  Binary LogisticRegression.predict_proba: p1 = expit(x . coef + intercept), p0 = 1 - p1 (float64; x is float32).
All random draws (counts, normals, uniforms) are INPUTS.
"""
import numpy as np


def gmm_sample(means, covars, counts, normals):
    """means/covars [K,D] f64, counts [K] int, normals [n,D] f64 (in component order) -> z [n,D] float32."""
    comp = np.repeat(np.arange(len(counts)), counts)
    z = means[comp] + normals * np.sqrt(covars[comp])
    return z.astype(np.float32)


def lr_prob(z, coef, intercept, target):
    """z [n,D] float32; coef [1,D] f64, intercept [1] f64 -> P(class == target) float64 [n]."""
    d = z.astype(np.float64) @ coef.reshape(-1).astype(np.float64) + float(np.asarray(intercept).reshape(-1)[0])
    p1 = 1.0 / (1.0 + np.exp(-d))
    return p1 if target == 1 else 1.0 - p1


def rejection_mask(z, clfs, uniforms):
    """clfs: list of (coef, intercept, target).  Returns probs [A,n], accum [n], accepted [n] bool."""
    probs = np.stack([lr_prob(z, w, b, t) for (w, b, t) in clfs])
    accum = np.ones(z.shape[0], np.float64)
    for p in probs:
        accum = accum * p
    return probs, accum, uniforms < accum
