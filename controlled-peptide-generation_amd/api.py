"""Inference helpers on the MI355X path (counterpart of the reference's api.py) - ADJACENT row: thin
callers of the same kernels, with the reference's names, arguments and return values:
  Vocab :27-75, load_trained_model :78-98, encode_sequence :101-115, sample_from_model :118-149, interpolate_z :152-205,
  generate_interpolated_samples :208-238, recon_sequence :241-255, interpolate_peptides :258-274, pretty_print_samples
  :277-287, get_model_and_vocab_path :290-305, get_result_for_model :308-334.
Differences: the model lives on the GPU (`load_trained_model(..., device=)`; the reference maps to CPU) and a checkpoint
that LACKS a parameter of the model is an error instead of being silently ignored (strict=False, :93); keys the model does not have
(the AAE discriminator the reference's comment at :94 names) are ignored exactly as the reference ignores them."""
import codecs
import json
import logging
import os

import numpy as np
import torch

import cfg
from models.model import RNN_VAE

LOG = logging.getLogger("GenerationAPI")


class Vocab:
    def __init__(self, vocab_path):
        self.fix_length = cfg.max_seq_len
        self.ix2word, self.word2ix = {}, {}
        with codecs.open(vocab_path, 'r', 'utf-8') as f:
            for line in f:
                parts = line.split()
                word, ix = " ".join(parts[:-1]), int(parts[-1])
                self.ix2word[ix], self.word2ix[word] = word, ix
        self.special_tokens = {'<unk>', '<pad>', '<start>', '<eos>'}
        self.special_tokens_ix = {self.word2ix[w] for w in self.special_tokens}

    def to_ix(self, seq, fix_length=True):
        if isinstance(seq, str):
            seq = seq.split()
        elif not isinstance(seq, list):
            raise ValueError('Only strings or lists of strings accepted.')
        seq = (["<start>"] if seq[0] != "<start>" else []) + seq
        seq = seq + (["<eos>"] if seq[-1] != "<eos>" else [])
        if fix_length:
            seq = seq + ["<pad>"] * (self.fix_length - len(seq))
        return torch.LongTensor([self.word2ix[t] for t in seq]).view(1, -1)

    def to_word(self, seq, print_special_tokens=True):
        ids = [int(s) for s in seq]
        if not print_special_tokens:
            ids = [i for i in ids if i not in self.special_tokens_ix]
        return [self.ix2word[i] for i in ids]

    def size(self):
        return len(self.ix2word)


def load_state_dict_checked(model, sd):
    """load_state_dict as the reference calls it (api.py:90-95: strict=False "only meaning to ignore AAE discriminator from AAE"):
    UNEXPECTED keys - parameters of modules this model does not have, e.g. an AAE checkpoint's discriminator - are ignored (logged
    once), so such checkpoints load as they do in the reference.  MISSING keys stay an error (the reference's own commented-out
    `assert not missing_keys`): they would leave parts of the model randomly initialised; the aliased `decoder.emb.weight` (same
    tensor as `word_emb.weight`) is the one exemption.  Shape mismatches raise in torch itself, as in the reference."""
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if k != 'decoder.emb.weight']
    if missing:
        raise RuntimeError('checkpoint does not match the model: missing {}'.format(missing))
    if unexpected:
        LOG.warning('checkpoint keys ignored (no such module in the model): %s', sorted(unexpected))
    return model


def load_trained_model(model_path, n_vocab, device=None):
    device = device or torch.device('cuda')
    model = RNN_VAE(n_vocab, max_seq_len=cfg.max_seq_len, **cfg.model)
    load_state_dict_checked(model, torch.load(model_path, map_location='cpu'))
    model = model.to(device)
    model.device = device
    model.eval()
    return model


def encode_sequence(model, vocab, sequence, sample_q='max'):
    """One (string) sequence -> z: the posterior mean ('max') or sample_q draws from the posterior, [sample_q, z_dim]."""
    with torch.no_grad():
        mu, logvar = model.forward_encoder(vocab.to_ix(sequence).to(model.device))
        from cpg import ops
        ops.check_persistent()   # interactive call: a host sync is fine, a silently wrong encoding is not
        if sample_q == 'max':
            return mu
        return torch.cat([model.sample_z(mu, logvar) for _ in range(sample_q)], dim=0)


def sample_from_model(model, vocab, z=None, c=None, n_samples=2, print_special_tokens=True, **sample_kwargs):
    """generate_sentences wrapper -> {'predictions': [[tokens...] per hypothesis] per sample, 'z', 'c'}."""
    with torch.no_grad():
        samples, z, c = model.generate_sentences(n_samples, z=z, c=c, **sample_kwargs)
    if sample_kwargs['sample_mode'] == 'beam':
        predictions = [[vocab.to_word(hyp, print_special_tokens) for hyp in s] for s in samples]
    else:
        predictions = [[vocab.to_word(s, print_special_tokens)] for s in samples.cpu().tolist()]
    return {'predictions': predictions, 'z': z, 'c': c}


def interpolate_z(z_start, z_end, c=None, method='linear', n_samples=2):
    """z_start, z_end [1, z_dim] tensors -> (matrix [n_samples + 2, z_dim] of points from start to end, their weights).
    'linear': equally spaced; 'tanh': the same steps squashed through tanh(4w - 2); 'slerp': great-circle interpolation."""
    a, b = z_start.detach().cpu().numpy(), z_end.detach().cpu().numpy()
    steps = np.arange(1, n_samples + 1) / (n_samples + 1)
    if method == 'linear':
        weights = steps
        mids = [(1 - w) * a + w * b for w in weights]
    elif method == 'tanh':
        weights = (np.tanh(steps * 4 - 2) + 1) / 2
        mids = [(1 - w) * a + w * b for w in weights]
    elif method == 'slerp':
        weights = steps
        p0, p1 = a.squeeze(0), b.squeeze(0)
        omega = np.arccos(np.dot(p0 / np.linalg.norm(p0), p1 / np.linalg.norm(p1)))
        so = np.sin(omega)
        mids = [(np.sin((1.0 - w) * omega) / so * p0 + np.sin(w * omega) / so * p1)[None, :] for w in weights]
    else:
        raise ValueError("Please use another interpolation method.")
    return np.vstack([a] + mids + [b]), list(np.concatenate(([0.], weights, [1.])))


def generate_interpolated_samples(model, vocab, z_start, z_end, c=None, interpolation_method='linear',
                                  interpolation_samples=2, **sample_kwargs):
    z_list, weights = interpolate_z(z_start, z_end, c=c, method=interpolation_method, n_samples=interpolation_samples)
    if c is None:   # the reference sets attribute 1 for every interpolated sample
        c = torch.zeros((z_list.shape[0], 2))
        c[:, 1].fill_(1)
    samples = sample_from_model(model, vocab, z=torch.Tensor(z_list), c=c, n_samples=z_list.shape[0], **sample_kwargs)
    samples['interpolation'] = weights
    return samples


def recon_sequence(model, vocab, sequence, sample_q, c, **mb_sample_kwargs):
    z = encode_sequence(model, vocab, sequence, sample_q)
    return sample_from_model(model, vocab, z, c, z.shape[0], **mb_sample_kwargs)


def interpolate_peptides(model, vocab, sequence_start, sequence_end, interpolation_kwargs={}, mb_sample_kwargs={}):
    z_start = encode_sequence(model, vocab, sequence_start, sample_q='max')
    z_end = encode_sequence(model, vocab, sequence_end, sample_q='max')
    return generate_interpolated_samples(model, vocab, z_start, z_end, **interpolation_kwargs, **mb_sample_kwargs)


def pretty_print_samples(samples, print_all_hypotheses=True):
    res = []
    for i, sample in enumerate(samples):
        if len(sample) > 1 and not print_all_hypotheses:
            sample = sample[:1]
        if len(sample) == 1:
            res.append('i {}: {}'.format(i, ' '.join(sample[0])))
        else:
            res += ['i {} - hyp {}: {}'.format(i, j, ' '.join(hyp)) for j, hyp in enumerate(sample)]
    return '\n'.join(res)


def get_model_and_vocab_path():
    """(checkpoint of the final phase-1 iteration - or the highest saved one -, vocab.dict, run dir) under cfg.savepath."""
    base = cfg.savepath
    model_path = '{}/model_{}.pt'.format(base, cfg.vae.n_iter)
    files = os.listdir(base)
    if os.path.basename(model_path) not in files:
        LOG.info("Selected model folder does not have fully trained model!")
        highest = max(int(name.split("_")[1].split(".")[0]) for name in files if name.startswith("model_") and name.endswith(".pt"))
        LOG.info("Using iteration {} instead".format(highest))
        model_path = '{}/model_{}.pt'.format(base, highest)
    LOG.info('api.main() load up from rundir={} model={}'.format(base, model_path))
    return model_path, '{}/vocab.dict'.format(base), base


def get_result_for_model(model_path, print_results=False):
    """The result.json entry logged at the checkpoint's iteration ({} when there is none)."""
    with open(os.path.join(os.path.dirname(model_path), 'result.json'), 'r') as f:
        data = json.load(f)
    iteration = os.path.basename(model_path).split(".")[0].split("_")[1]
    stats = {}
    for res in data:
        if str(res['it']) == str(iteration):
            stats = res
    if not stats:
        LOG.info("No results for {} found.".format(model_path))
    if print_results:
        print("Results for model {}".format(model_path))
        print(json.dumps(stats, indent=2))
    return stats
