"""CLaSS proposal density + z-space rejection sampling on the MI355X path.

Counterpart of the reference's density_modeling.py: RejSampleBase :38-60 and mogQ :63-80 (the alternatives fullQ /
gaussianQ / evaluate_nll are diagnostics outside the hot path and not provided).
  * the proposal Q_xi(z) is a diagonal Gaussian mixture fitted ONCE on the host with scikit-learn, as in the reference;
  * `rejection_sample(n)` keeps the reference's return triple (samples_z torch f32 [n,D], scores dict of float64 arrays,
    accepted bool array); the arithmetic - mixture draw, logistic-regression scoring of every attribute, product and the
    accept test - runs in two HIP kernels (cpg_gmm_sample, cpg_lr_score_accept);
  * rng='numpy' (default) consumes numpy's global generator in exactly scikit-learn's / the reference's order
    (multinomial counts, per-component standard normals, then np.random.uniform), so a seeded run reproduces the
    reference's samples and accept mask; rng='device' draws with the on-device Philox streams instead (throughput).
"""
import numpy as np
import torch

from cpg import class_sampler, ops


def prior_logpdf(z):
    """log N(z; 0, I) of one point (reference density_modeling.py:11-14)."""
    import math
    return -0.5 * z.shape[0] * math.log(math.tau) - 0.5 * float((z.double() ** 2).sum())


def evaluate_nll(q, points):
    """(NLL under Q, NLL under the prior) per point, each point perturbed by ONE shared normal draw along its posterior
    standard deviations as the reference does (density_modeling.py:118-128: `torch.randn(1).item()`); host-side diagnostic."""
    mu, lv = points
    N = mu.shape[0]
    llq = llp = 0.0
    for s in range(N):
        z = mu[s] + (0.5 * lv[s]).exp() * torch.randn(1).item()
        llq += q.logpdf(z)
        llp += prior_logpdf(z)
    return -llq / N, -llp / N


class RejSampleBase:
    rng = 'numpy'
    device = torch.device('cuda')
    _philox = None
    _cdf = None

    def init_attr_classifiers(self, attr_clfs, clf_targets):
        """attr_clfs: {attr: fitted binary sklearn LogisticRegression (or any object with coef_/intercept_)}."""
        self.attr_clfs = attr_clfs
        self.clf_targets = clf_targets
        names = list(attr_clfs)
        coef = np.stack([np.asarray(attr_clfs[a].coef_, np.float64).reshape(-1) for a in names])
        icpt = np.array([float(np.asarray(attr_clfs[a].intercept_).reshape(-1)[0]) for a in names], np.float64)
        tgt = np.array([int(clf_targets[a]) for a in names], np.int32)
        for a in names:
            classes = getattr(attr_clfs[a], 'classes_', np.array([0, 1]))
            assert len(classes) == 2, 'binary attribute classifiers expected'
        self._dev_clf = (torch.from_numpy(coef).to(self.device), torch.from_numpy(icpt).to(self.device),
                         torch.from_numpy(tgt).to(self.device))

    def score_clf(self, attr_name, z):
        names = list(self.attr_clfs)
        coef, icpt, tgt = self._dev_clf
        i = names.index(attr_name)
        z = z.to(self.device).float()
        u = torch.zeros(z.shape[0], device=self.device, dtype=torch.float64)
        probs, _, _ = class_sampler.lr_score_accept(z, coef[i:i + 1], icpt[i:i + 1], tgt[i:i + 1], u)
        return probs[0].cpu().numpy()

    def rejection_sample(self, n_samples, prefix='clfZ', return_device=False, shard=(0, 1)):
        """shard = (rank, world) (device rng only): this call draws rows [rank*n/world, (rank+1)*n/world) of the round's
        n_samples-row stream - the union over ranks is exactly what one rank would draw alone (counter-based streams)."""
        rank, world = shard
        n_local, row0 = n_samples, 0
        if world > 1:
            assert self.rng == 'device', "sharded rounds need rng='device' (numpy's global generator cannot be split by rows)"
            assert n_samples % (4 * world) == 0, 'n_samples must be a multiple of 4*world'
            n_local, row0 = n_samples // world, rank * (n_samples // world)
        samples_z = self.sample(n_local, to_cpu=False, _rows=(row0, n_samples))
        if self.rng == 'numpy':
            uniforms = torch.from_numpy(np.random.uniform(size=n_samples)).to(self.device)
        else:
            seed, off = self._next_philox(n_samples)
            uniforms = ops.rng_uniform((n_local,), seed, off + row0 // 2, self.device, dtype=torch.float64)  # 2 doubles per counter
        coef, icpt, tgt = self._dev_clf
        probs, accum, acc = class_sampler.lr_score_accept(samples_z, coef, icpt, tgt, uniforms)
        if return_device:
            return samples_z, probs, accum, acc
        scores_z = {prefix + '_prob_accum': accum.cpu().numpy()}
        for i, attr in enumerate(self.attr_clfs):
            scores_z['{}_{}={}'.format(prefix, attr, self.clf_targets[attr])] = probs[i].cpu().numpy()
        return samples_z.cpu(), scores_z, acc.cpu().numpy().astype(bool)

    def score_names(self, prefix='clfZ'):
        return [prefix + '_prob_accum'] + ['{}_{}={}'.format(prefix, a, self.clf_targets[a]) for a in self.attr_clfs]

    def _next_philox(self, n):
        if self._philox is None:
            self._philox = [1238, 0]
        off = self._philox[1]
        self._philox[1] += (n + 3) // 4 + 1
        return self._philox[0], off


class mogQ(RejSampleBase):
    def __init__(self, mu, logvar, n_components=10, z_num_samples=10, **mog_kwargs):
        import sklearn.mixture
        self.mu, self.logvar = mu, logvar
        self.N, self.D = mu.shape
        self.z = torch.cat([mu + (0.5 * logvar).exp() * torch.randn_like(logvar) for _ in range(z_num_samples)], dim=0)
        self.n_components = n_components
        self.mog = sklearn.mixture.GaussianMixture(n_components=n_components, **mog_kwargs)
        self.mog.fit(self.z.cpu().numpy())
        print('mog-{}. Converged: {} in {} iters, log likelihood lower bound: {:.4f}'.format(
            n_components, self.mog.converged_, self.mog.n_iter_, self.mog.lower_bound_))
        self._upload()

    @classmethod
    def from_params(cls, weights, means, covars, device=None):
        """Build from mixture parameters directly (diag covariance) - synthetic benchmarks / tests."""
        self = cls.__new__(cls)
        self.mog = None
        self._w, self._m, self._c = (np.asarray(a, np.float64) for a in (weights, means, covars))
        self.D = self._m.shape[1]
        if device is not None:
            self.device = device
        self._upload()
        return self

    def _upload(self):
        if self.mog is not None:
            assert self.mog.covariance_type == 'diag', "only covariance_type='diag' (the reference's setting) is on the GPU path"
            self._w, self._m, self._c = self.mog.weights_, self.mog.means_, self.mog.covariances_
        self._dm = torch.from_numpy(np.ascontiguousarray(self._m)).to(self.device)
        self._dc = torch.from_numpy(np.ascontiguousarray(self._c)).to(self.device)

    def logpdf(self, x):
        assert x.dim() == 1, 'expecting  single sample'
        return self.mog.score(x.view(1, -1).cpu().numpy())

    def sample(self, n_samples, to_cpu=True, _rows=None):
        """_rows = (row0, n_total): draw rows [row0, row0 + n_samples) of an n_total-row device stream (sharded rounds)."""
        K, D = self._m.shape
        if self.rng == 'numpy':
            rs = np.random.mtrand._rand  # the generator scikit-learn's sample() uses for random_state=None
            counts = rs.multinomial(n_samples, self._w)
            normals = np.concatenate([rs.standard_normal(size=(int(c), D)) for c in counts], 0)
            comp = torch.from_numpy(np.repeat(np.arange(K), counts).astype(np.int32)).to(self.device)
            normals = torch.from_numpy(normals).to(self.device)
        else:
            row0, n_total = _rows if _rows is not None else (0, n_samples)
            assert row0 % 4 == 0
            seed, off = self._next_philox(n_total)
            u = ops.rng_uniform((n_samples,), seed, off + row0 // 4, self.device)
            if getattr(self, '_cdf', None) is None:
                self._cdf = torch.from_numpy(np.cumsum(self._w)).to(self.device).float()
            # component of each row by inverse CDF (rows are iid: no need for scikit-learn's component-sorted order)
            comp = torch.searchsorted(self._cdf, u).clamp(max=K - 1).to(torch.int32)
            Dp = -(-D // 4) * 4   # rows start on a 4-element counter boundary, so any row range of the stream can be drawn alone
            seed, off = self._next_philox(n_total * Dp)
            normals = ops.rng_normal((n_samples, Dp), seed, off + row0 * (Dp // 4), self.device)[:, :D].double()
        z = class_sampler.gmm_sample(self._dm, self._dc, comp, normals)
        return z.cpu() if to_cpu else z
