"""Inference helpers on the MI355X path (counterpart of the reference's api.py:27-149,241-274) - ADJACENT row
(SURVEY 8f rank 3): thin callers of the same kernels.  Vocab reads the `vocab.dict` main.py writes; load_trained_model
loads a reference-format checkpoint (identical state-dict keys) onto the GPU."""
import codecs

import numpy as np
import torch

import cfg
from models.model import RNN_VAE


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
    """load_state_dict that tolerates exactly what the reference's checkpoints legitimately lack or add - nothing on the
    VAE path.  The reference passes strict=False (api.py:93), which silently ignores any mismatch; a mismatched checkpoint
    would leave parts of the model randomly initialised, so every missing / unexpected key is an error here except the
    aliased `decoder.emb.weight` (same tensor as `word_emb.weight`)."""
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if k != 'decoder.emb.weight']
    if missing or unexpected:
        raise RuntimeError('checkpoint does not match the model: missing {}, unexpected {}'.format(missing, unexpected))
    return model


def load_trained_model(model_path, n_vocab, device=None):
    device = device or torch.device('cuda')
    model = RNN_VAE(n_vocab, max_seq_len=cfg.max_seq_len, **cfg.model)
    load_state_dict_checked(model, torch.load(model_path, map_location='cpu'))
    model = model.to(device)
    model.device = device
    model.eval()
    return model


@torch.no_grad()
def encode_sequence(model, vocab, seq, sample_q=False):
    ids = vocab.to_ix(seq).to(model.device)
    mu, logvar = model.forward_encoder(ids)
    return (model.sample_z(mu, logvar) if sample_q else mu), mu, logvar


@torch.no_grad()
def sample_from_model(model, vocab, z=None, c=None, n_samples=1, **sample_kwargs):
    if z is not None:
        n_samples = z.size(0)
    sents, z, c_ix = model.generate_sentences(n_samples, z, c, **sample_kwargs)
    if sample_kwargs.get('sample_mode') == 'beam':
        sents = [h[0] for h in sents]
    else:
        sents = [row.tolist() for row in sents.cpu()]
    return [" ".join(vocab.to_word(s, print_special_tokens=False)) for s in sents], z, c_ix


def recon_sequence(model, vocab, seq, **sample_kwargs):
    z, _, _ = encode_sequence(model, vocab, seq)
    return sample_from_model(model, vocab, z=z, **sample_kwargs)[0][0]


def interpolate_peptides(model, vocab, seq1, seq2, steps=5, **sample_kwargs):
    z1, _, _ = encode_sequence(model, vocab, seq1)
    z2, _, _ = encode_sequence(model, vocab, seq2)
    w = torch.linspace(0, 1, steps, device=model.device).view(-1, 1)
    return sample_from_model(model, vocab, z=(1 - w) * z1 + w * z2, **sample_kwargs)[0]
