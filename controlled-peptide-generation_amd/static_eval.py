"""Smoke evaluation of a trained checkpoint through api.py (counterpart of the reference's static_eval.py) - ADJACENT row (SURVEY
8f rank 3): thin callers of the same kernels, with the reference's function names and printed line formats:
  test_interpolated_peptides :32-51, test_interpolated_z :54-82, test_sampling :85-106, test_reconstruction :109-139,
  test_reconstruction_interpol :142-162, main :165-217 (run-dir discovery, result lookup, the five checks in the reference's order).
Differences: the sequences to reconstruct are an ARGUMENT of the two reconstruction checks (the reference reads a module-global
`args` that only exists under `__main__`); the model lives on the GPU; `--long` (t-SNE plots of the dumped states through vis/,
out of scope: SURVEY section 2) only makes sure the states dump exists and says so."""
import argparse
import logging
import os

import numpy as np
import torch

import cfg
from api import (Vocab, generate_interpolated_samples, get_model_and_vocab_path, get_result_for_model, interpolate_peptides,
                 load_trained_model, pretty_print_samples, recon_sequence, sample_from_model)

LOG = logging.getLogger('GenerationAPI')

DEFAULT_SEQS = ('M T G E I D T A M L I G G I E F F L K F A I Y Y F H E R A W Q L I R, '
                'M D K L I V L K M L N S K L P Y G Q R K P F S L R')
HARD_SAMPLERS = ({'sample_mode': 'greedy'}, {'sample_mode': 'categorical', 'temp': 1.0}, {'sample_mode': 'categorical', 'temp': 0.3},
                 {'sample_mode': 'beam', 'beam_size': 5, 'n_best': 3})


def _split_seqs(seqs):
    if isinstance(seqs, str):
        return [s.strip().split() for s in seqs.split(',')]
    return [list(s) for s in seqs]


def test_interpolated_peptides(model, vocab, start='M L L L L L A L A L L A L L L A L L L', end='M S S S S S L A A A L L'):
    """Greedy decodes along the path between the posterior means of two fixed peptides, for each interpolation method."""
    out = {}
    for method in ('linear', 'tanh', 'slerp'):
        LOG.info("INTERPOLATING WITH {} METHOD".format(method))
        peps = interpolate_peptides(model, vocab, start, end,
                                    interpolation_kwargs=dict(c=None, interpolation_method=method, interpolation_samples=9),
                                    mb_sample_kwargs=dict(sample_mode='greedy'))
        for w, p in zip(peps['interpolation'], peps['predictions']):
            print("{:.2f}".format(w), " ".join(p[0]))
        out[method] = peps
    return out


def test_interpolated_z(model, vocab):
    """Decodes along the tanh path between two prior draws, greedy and beam."""
    z_start, z_end = model.sample_z_prior(1), model.sample_z_prior(1)
    print('# interpolate between z1, z2 sampled from prior. vary sampling')
    out = []
    for kwargs in (HARD_SAMPLERS[0], HARD_SAMPLERS[3]):
        print('### interpolate z1 z2 from prior: ', kwargs)
        samples = generate_interpolated_samples(model, vocab, z_start, z_end, c=None, interpolation_method='tanh',
                                                interpolation_samples=11, **kwargs)
        for w, p in zip(samples['interpolation'], samples['predictions']):
            print("prior_zs - {:6s} - w={:.2f} - {}".format(kwargs['sample_mode'], w, " ".join(p[0])))
        out.append(samples)
    return out


def test_sampling(model, vocab, n_samples=4):
    """The same prior draws decoded by every hard sampling mode."""
    z_fix, c_fix = model.sample_z_prior(n_samples), model.sample_c_prior(n_samples)
    print('# sampled z from prior, varying sample_mode')
    out = []
    for kwargs in HARD_SAMPLERS:
        payload = sample_from_model(model, vocab, z=z_fix, c=c_fix, n_samples=n_samples, **kwargs)
        print('### prior: ', kwargs)
        print(pretty_print_samples(payload['predictions']))
        out.append(payload)
    return out


def test_reconstruction(model, vocab, seqs=DEFAULT_SEQS):
    """Each sequence encoded to z = mu and decoded by every hard mode, then beam-15 decodes of four posterior samples."""
    out = []
    for seq in _split_seqs(seqs):
        print('#### reco of', ' '.join(seq), '  -- z = mu = max_z q(z|x) ')
        for kwargs in HARD_SAMPLERS:
            recos = recon_sequence(model, vocab, seq, sample_q='max', c=None, **kwargs)
            print(pretty_print_samples(recos['predictions'], print_all_hypotheses=False), kwargs['sample_mode'])
            out.append(recos)
        print('#### reco  of', ' '.join(seq), '  -- beam 15, z = 4x sampled q(z|x) ')
        recos = recon_sequence(model, vocab, seq, sample_q=4, c=None, sample_mode='beam', beam_size=15, n_best=3)
        print(pretty_print_samples(recos['predictions'], print_all_hypotheses=False))
        out.append(recos)
    return out


def test_reconstruction_interpol(model, vocab, seqs=DEFAULT_SEQS):
    """Beam-15 decodes along the tanh path between the posterior means of consecutive sequences."""
    seqs = _split_seqs(seqs)
    out = []
    for seq1, seq2 in zip(seqs[:-1], seqs[1:]):
        print('#### reco interpol start source: ', ' '.join(seq1), '  -- z = mu = max_z q(z|x), beam 15')
        samples = interpolate_peptides(model, vocab, seq1, seq2,
                                       interpolation_kwargs=dict(c=None, interpolation_method='tanh', interpolation_samples=9),
                                       mb_sample_kwargs=dict(sample_mode='beam', beam_size=15, n_best=3))
        for w, p in zip(samples['interpolation'], samples['predictions']):
            print("recon interpol - w={:.2f} - {}".format(w, " ".join(p[0])))
        print('#### reco interpol end source:   ', ' '.join(seq2))
        out.append(samples)
    return out


def main(args=None):
    seqs = getattr(args, 'seqs', DEFAULT_SEQS)
    model_path, vocab_path, base = get_model_and_vocab_path()
    vocab = Vocab(vocab_path)
    model = load_trained_model(model_path, vocab.size())
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    get_result_for_model(model_path, print_results=True)
    if getattr(args, 'long', False):
        names = [os.path.join(base, "states_{}_{}".format(split, cfg.vae.n_iter)) for split in ('train', 'val', 'test')]
        have = [any(os.path.exists(n + ext) for ext in ('.h5', '.npz')) for n in names]
        LOG.info("states dumps present: %s; the t-SNE / KDE plots of vis/ are out of scope of this build (main.py --phase 1 writes "
                 "the dumps, sample_pipeline.py consumes them)", dict(zip(names, have)))
    test_interpolated_peptides(model, vocab)
    test_sampling(model, vocab, n_samples=4)
    test_interpolated_z(model, vocab)
    test_reconstruction(model, vocab, seqs)
    test_reconstruction_interpol(model, vocab, seqs)
    from cpg import ops
    ops.check_persistent()


if __name__ == "__main__":
    logging.basicConfig(format='%(asctime)s %(message)s', datefmt='%m/%d/%Y %I:%M:%S %p', level=logging.INFO)
    LOG.info("Running API test.")
    parser = argparse.ArgumentParser(argument_default=argparse.SUPPRESS, description='Override config float & string values')
    cfg._cfg_import_export(parser, cfg, mode='fill_parser')
    parser.add_argument('--seqs', default=DEFAULT_SEQS, help='comma separated list of seqs to reconstruct between')
    parser.add_argument('--long', '-long', action='store_true', default=False, help='check the states dumps (plots: out of scope)')
    a = parser.parse_args()
    cfg._override_config(a, cfg)
    cfg._update_cfg()
    main(a)
