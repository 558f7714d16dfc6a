"""CLaSS sampling driver on the MI355X path (counterpart of the reference's sample_pipeline.py:129-324).

Round structure and stop rule are the reference's: rounds of `n_samples_per_round` proposals -> z-space rejection ->
decode -> drop duplicates within the round and against earlier rounds -> stop once `n_samples_acc` accepted peptides
exist (:299-322).  Differences, all documented in DESIGN.md:
  * decode mode is selectable ('beam' with beam_size 5 as the reference's decode_from_z :129-139, or 'greedy');
  * `decode_accepted_only=True` classifies first and decodes only accepted z - per-z independent, so the accepted
    peptides are identical, but c must then be passed explicitly (the reference draws c per 1024-chunk);
  * encodings come from `states_<split>_<iter>.npz` (same field names as the reference's h5: src, z, mu, logvar, label,
    split) or are computed on the fly; modlamp descriptors (H, uH, charge) need modlamp, which is not installed: columns
    are filled when it is importable, skipped otherwise;
  * under data parallelism every rank runs its own rounds (rank-distinct seeds) and accepted rows are all-gathered.
"""
import argparse
import datetime
import logging
import os

import numpy as np
import torch

import cfg
from density_modeling import mogQ

LOG = logging.getLogger('GenerationAPI')
logging.basicConfig(format='%(asctime)s %(message)s', datefmt='%m/%d/%Y %I:%M:%S %p', level=logging.INFO)

Q_CLASS = mogQ
Q_KWARGS = {'n_components': None, 'z_num_samples': 10, 'covariance_type': None}


def decode_from_z(z, model, dataset, sample_mode='beam', beam_size=5, chunk=1024, c=None):
    """All z -> peptide strings; chunks of 1024 like the reference (bigger chunks only change the c draw order)."""
    out = []
    LOG.info('Decoder decoding: {}'.format(sample_mode))
    if hasattr(dataset, 'ids_to_peptides') and sample_mode in ('beam', 'greedy') and c is not None:
        # array path: same decode kernels, best hypothesis / greedy row handed over as arrays and turned into strings
        # in one vectorised pass (the nested python lists of the reference format cost more than the decoding)
        from cpg import decode as cdecode
        was_training = model.training
        model.eval()
        try:
            for i, zchunk in enumerate(torch.split(z, chunk)):
                cc = c[i * chunk:i * chunk + zchunk.size(0)].to(model.device).float()
                zz = zchunk.to(model.device).float()
                if sample_mode == 'beam':
                    hyps, lens, _ = cdecode.decode_beam_arrays(model.decoder, zz, cc, model.MAX_SEQ_LEN, beam_size, 3, 1)
                    best = hyps[:, 0, :].copy()
                    best[np.arange(best.shape[1])[None, :] >= lens[:, 0:1]] = -1
                    out += dataset.ids_to_peptides(best)
                else:
                    out += dataset.ids_to_peptides(cdecode.decode_hard(model.decoder, zz, cc, model.MAX_SEQ_LEN).cpu().numpy())
        finally:
            model.train()  # generate_sentences always leaves the model in train mode
        return out
    for i, zchunk in enumerate(torch.split(z, chunk)):
        cc = None if c is None else c[i * chunk:i * chunk + zchunk.size(0)]
        kw = dict(sample_mode=sample_mode)
        if sample_mode == 'beam':
            kw['beam_size'] = beam_size
        s, _, _ = model.generate_sentences(zchunk.size(0), zchunk.to(model.device), cc, **kw)
        out += [h[0] for h in s] if sample_mode == 'beam' else [row.tolist() for row in s.cpu()]
    return dataset.idx2sentences(out, print_special_tokens=False)


def get_encodings_from_states(query, split, attributes=None, savepath=None, n_iter=None):
    attributes = attributes if attributes is not None else cfg.attributes
    fn = os.path.join(savepath or cfg.savepath, 'states_{}_{}.npz'.format(split, n_iter if n_iter is not None else cfg.vae.n_iter))
    assert os.path.exists(fn), 'need dumped states ({}); run dump_encodings first'.format(fn)
    f = np.load(fn)
    mu, logvar, lab = torch.from_numpy(f['mu']).double(), torch.from_numpy(f['logvar']).double(), torch.from_numpy(f['label'])
    col = {k: i for i, (k, _) in enumerate(attributes)}
    sel = torch.ones(lab.shape[0], dtype=torch.bool)
    for attr, val in query.items():
        sel &= lab[:, col[attr]] == val
    return mu[sel], logvar[sel]


@torch.no_grad()
def dump_encodings(model, ids, labels, split, savepath, n_iter, batch=4096):
    """Encode-only pass (mu, logvar, z=mu) -> states_<split>_<iter>.npz (schema of vis/scripts/build_index.py:32-81)."""
    mus, lvs = [], []
    for chunk in torch.split(ids, batch):
        mu, lv = model.forward_encoder(chunk.to(model.device))
        mus.append(mu.cpu())
        lvs.append(lv.cpu())
    mu, lv = torch.cat(mus).numpy(), torch.cat(lvs).numpy()
    os.makedirs(savepath, exist_ok=True)
    np.savez_compressed(os.path.join(savepath, 'states_{}_{}.npz'.format(split, n_iter)), src=ids.cpu().numpy(),
                        z=mu.astype(np.float16), mu=mu.astype(np.float16), logvar=lv.astype(np.float16),
                        label=np.asarray(labels), split=np.zeros((ids.shape[0], 1), np.int64))


def build_clfZ(zpos_mu, zneg_mu):
    """Logistic regression between attr=1 and attr=0 encodings (reference :169-192); host-side, one-off."""
    from sklearn.linear_model import LogisticRegression
    X = torch.cat([zpos_mu, zneg_mu], 0).numpy()
    Y = np.concatenate([np.ones(zpos_mu.shape[0]), np.zeros(zneg_mu.shape[0])])
    clf = LogisticRegression(solver='lbfgs', max_iter=200)
    clf.fit(X, Y)
    LOG.info('Fitted LogReg classifier in z-space: {} pos, {} neg. train accuracy={:.5f}'.format(
        zpos_mu.shape[0], zneg_mu.shape[0], clf.score(X, Y)))
    return clf


def compute_modlamp(df):
    try:
        from modlamp.analysis import GlobalAnalysis
    except ImportError:
        return df
    ana = GlobalAnalysis(df.peptide.str.replace(' ', ''))
    ana.calc_H(); ana.calc_uH(); ana.calc_charge()
    df.loc[:, 'H'], df.loc[:, 'uH'], df.loc[:, 'charge'] = ana.H[0], ana.uH[0], ana.charge[0]
    return df


def get_new_samples(model, dataset, Q, n_samples, sample_mode='beam', decode_accepted_only=False):
    import pandas as pd
    samples_z, scores_z, accept_z = Q.rejection_sample(n_samples=n_samples)
    if decode_accepted_only:
        keep = torch.from_numpy(np.nonzero(accept_z)[0])
        samples_z = samples_z[keep]
        scores_z = {k: v[accept_z] for k, v in scores_z.items()}
        accept_z = accept_z[accept_z]
    c = torch.zeros(samples_z.shape[0], 2, device=model.device)
    c[:, 1] = 1.0
    samples = decode_from_z(samples_z, model, dataset, sample_mode=sample_mode, c=c if decode_accepted_only else None)
    return pd.DataFrame({'peptide': samples, 'z': [tuple(z.tolist()) for z in samples_z], 'accept_z': accept_z, **scores_z})


def one_sampling_round(model, dataset, Q, n_samples_per_round, **kw):
    df = compute_modlamp(get_new_samples(model, dataset, Q, n_samples_per_round, **kw))
    df['accept'] = df['accept_z']
    return df


def run_rounds(model, dataset, Q, n_samples_per_round, n_samples_acc, max_rounds=1000, **kw):
    import pandas as pd
    samples = pd.DataFrame(columns=['peptide'])
    rounds = 0

    def finished(df):
        return len(df) >= n_samples_acc and df['accept'].sum() >= n_samples_acc

    while not finished(samples) and rounds < max_rounds:
        rounds += 1
        LOG.info("Round #{}".format(rounds))
        new = one_sampling_round(model, dataset, Q, n_samples_per_round, **kw)
        new = new.loc[new.peptide.drop_duplicates().index]
        new = new[~new['peptide'].isin(samples['peptide'])]
        samples = pd.concat([samples, new], ignore_index=True, sort=False)
        LOG.info('Q_xi(z|a) rejection sampling acceptance rate: {}/{}'.format(samples['accept_z'].sum(), len(samples)))
    return samples


def save_samples(samples, basedir, fn_prefix):
    out = os.path.join(basedir, fn_prefix) + '_{}'.format(datetime.datetime.now().isoformat().split('T')[0])
    os.makedirs(basedir, exist_ok=True)
    with open(out + '.plain.txt', 'w') as fh:
        fh.write(samples['peptide'].to_string(index=False))
    samples.drop(columns='z').to_csv(out + '.csv', index_label='idx')
    samples.to_pickle(out + '.pkl')
    acc = samples[samples.accept.astype(bool)]
    acc.drop(columns='z').to_csv('{}.accepted.{}.csv'.format(out, len(acc)), index_label='idx')
    acc.to_pickle('{}.accepted.{}.pkl'.format(out, len(acc)))
    LOG.info('Sample lists written to {}.*'.format(out))


def main(args):
    from cpg.synth import SyntheticPeptideLoader
    from models.model import RNN_VAE
    device = torch.device('cuda')
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    dataset = SyntheticPeptideLoader(cfg.vae.batch_size, cfg.max_seq_len, device, size=cfg.hw.synthetic_size, seed=cfg.seed)
    model = RNN_VAE(n_vocab=dataset.n_vocab, max_seq_len=cfg.max_seq_len, **cfg.model).to(device)
    model.device = device
    ckpt = cfg.vae.chkpt_path.format(cfg.vae.n_iter)
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location=device), strict=False)
        LOG.info('Loaded model from ' + ckpt)
    model.eval()
    for k in Q_KWARGS:
        if hasattr(args, 'Q_' + k):
            Q_KWARGS[k] = getattr(args, 'Q_' + k)
    query = {'amp': 1} if args.Q_select_amppos else {}
    mu, logvar = get_encodings_from_states(query=query, split='train')
    Q = Q_CLASS(mu, logvar, **Q_KWARGS)
    z_clfs = {a: build_clfZ(get_encodings_from_states({a: 1}, 'train')[0], get_encodings_from_states({a: 0}, 'train')[0])
              for a in ['amp', 'tox']}
    Q.init_attr_classifiers(z_clfs, clf_targets={'amp': 1, 'tox': 0})
    samples = run_rounds(model, dataset, Q, args.n_samples_per_round, args.n_samples_acc, sample_mode=args.sample_mode,
                         decode_accepted_only=args.decode_accepted_only)
    save_samples(samples, cfg.savepath, args.samples_outfn_prefix)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(argument_default=argparse.SUPPRESS, description='Override config float & string values')
    cfg._cfg_import_export(parser, cfg, mode='fill_parser')
    parser.add_argument('--QClass', default='mogQ')
    parser.add_argument('--Q_n_components', type=int, default=100)
    parser.add_argument('--Q_covariance_type', default='diag')
    parser.add_argument('--n_samples_per_round', type=int, default=5000)
    parser.add_argument('--n_samples_acc', type=int, default=100)
    parser.add_argument('--samples_outfn_prefix', default='samples')
    parser.add_argument('--Q_select_amppos', type=int, default=0)
    parser.add_argument('--sample_mode', default='beam')
    parser.add_argument('--decode_accepted_only', action='store_true', default=False)
    a = parser.parse_args()
    cfg._override_config(a, cfg)
    cfg._update_cfg()
    cfg._print(cfg)
    main(a)
