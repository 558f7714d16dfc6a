"""Synthetic random-vocab peptide batches and a minimal loader with the interface train_vae / main /
sample_pipeline use from the reference's AttributeDataLoader (data_processing/dataset.py:285-300):
`next_batch(name).text`, `idx2sentence(s)`, `idx2sentences(...)`, `n_vocab`, `TEXT.vocab.{itos,stoi}`.

The reference's curated CSVs are not reproducible from its repo and torchtext 0.3.1 is not installable
here, so this loader is the data source of this build.  Vocabulary: ids 0..3 = <unk>,<pad>,<start>,<eos>
(models/mutils.py:5-8), 4..23 = the 20 amino acids.  Sequence: <start> aa{L} <eos> <pad>*, L ~ U{5..T-2}.
"""
import numpy as np
import torch

SPECIALS = ['<unk>', '<pad>', '<start>', '<eos>']
AMINO = list("ACDEFGHIKLMNPQRSTVWY")


def synth_ids(B, T, V=24, generator=None, device="cpu"):
    """int64 [B,T] peptide batch (vectorised; same distribution as tests/golden/make_golden.py:synth_ids)."""
    g = generator
    L = torch.randint(5, T - 1, (B,), generator=g)
    body = torch.randint(4, V, (B, T), generator=g)
    pos = torch.arange(T).unsqueeze(0)
    ids = torch.where(pos <= L.unsqueeze(1), body, torch.ones_like(body))
    ids[:, 0] = 2
    ids[torch.arange(B), L + 1] = 3
    ids = torch.where(pos > (L + 1).unsqueeze(1), torch.ones_like(ids), ids)
    return ids.to(device)


HYDROPHOBIC = "AFILMVWY"
ATTR_NAMES = ('amp', 'tox', 'sol', 'anticancer', 'antihyper', 'hormone')   # column order of cfg.attributes (cfg.py amp dataset)


def synth_labels(ids, generator=None, p_na=0.2):
    """int64 [N, 6] attribute labels in the reference's convention (1 pos / 0 neg / -1 'na'; column order = cfg.attributes) for
    synthetic peptides, as DETERMINISTIC functions of the sequence so that the latent classifiers of the CLaSS pipeline have
    something to learn: amp = net charge (#K + #R - #D - #E) >= 1, tox = hydrophobic fraction (AFILMVWY) >= 0.45.  A random
    p_na of the amp / tox entries is unlabelled (-1) like most of the reference's corpus; the other four columns are all -1."""
    ids = ids.cpu()
    stoi = {a: i + len(SPECIALS) for i, a in enumerate(AMINO)}
    cnt = lambda letters: sum((ids == stoi[a]).sum(1) for a in letters)
    n_res = (ids >= len(SPECIALS)).sum(1).clamp(min=1)
    amp = (cnt("KR") - cnt("DE") >= 1).long()
    tox = (cnt(HYDROPHOBIC).float() / n_res.float() >= 0.45).long()
    lab = torch.full((ids.shape[0], len(ATTR_NAMES)), -1, dtype=torch.int64)
    lab[:, 0], lab[:, 1] = amp, tox
    na = torch.rand(ids.shape[0], 2, generator=generator) < p_na
    lab[:, :2][na] = -1
    return lab


class _Vocab:
    def __init__(self):
        self.itos = SPECIALS + AMINO
        self.stoi = {w: i for i, w in enumerate(self.itos)}


class _Text:
    def __init__(self):
        self.vocab = _Vocab()


class _Batch:
    def __init__(self, text):
        self.text = text


class SyntheticPeptideLoader:
    def __init__(self, mbsize, max_seq_len, device, size=20000, seed=1238, rank=0, **_ignored):
        self.mbsize, self.T, self.device = mbsize, max_seq_len, device
        self.TEXT = _Text()
        self.n_vocab = len(self.TEXT.vocab.itos)
        g = torch.Generator().manual_seed(seed)
        pool = synth_ids(size, max_seq_len, self.n_vocab, g)
        self.labels = synth_labels(pool, g)                       # [size, n_attr], reference convention 1 / 0 / -1
        # split column like the reference's csv (`split=train|val|test`): 80 / 10 / 10 by position
        self.split = np.array(['train'] * size, dtype=object)
        self.split[int(0.8 * size):int(0.9 * size)] = 'val'
        self.split[int(0.9 * size):] = 'test'
        self.pool = pool.to(device)
        self.gen = torch.Generator().manual_seed(seed + 1000 * (rank + 1))

    def subset(self, split=None, query=None):
        """(ids, labels) of the pool rows in `split` ('train' | 'val' | 'test' | comma list | None = all) whose labels match
        `query` {attr: value} - the role of AttributeDataset.get_subset_iterators in the reference (dataset.py), for the encode
        passes of build_index.py / sample_pipeline.get_encodings_from_dataloader."""
        sel = np.ones(len(self.split), bool)
        if split is not None:
            sel &= np.isin(self.split, [t.strip() for t in split.split(',')])
        for attr, val in (query or {}).items():
            sel &= (self.labels[:, ATTR_NAMES.index(attr)] == val).numpy()
        idx = torch.from_numpy(np.nonzero(sel)[0])
        return self.pool[idx.to(self.pool.device)], self.labels[idx]

    def print_stats(self):
        print('SyntheticPeptideLoader: {} sequences, vocab {}, max_seq_len {}'.format(self.pool.shape[0], self.n_vocab, self.T))

    def next_batch(self, iterator_name):
        """Random batch WITH replacement, like the reference's weighted multinomial sampler (dataset.py:72-77)."""
        idx = torch.randint(0, self.pool.shape[0], (self.mbsize,), generator=self.gen).to(self.pool.device)
        return _Batch(self.pool[idx])

    def idx2sentence(self, idxs, print_special_tokens=True):
        toks = [int(i) for i in idxs]
        if not print_special_tokens:
            toks = [i for i in toks if i >= len(SPECIALS)]
        return ' '.join(self.TEXT.vocab.itos[i] for i in toks)

    def idx2sentences(self, batch, print_special_tokens=True):
        return [self.idx2sentence(s, print_special_tokens) for s in batch]

    def ids_to_letters(self, ids):
        """Integer array [N,L] (entries < 0 = padding) -> (letters uint8 [N,L]: the residues of each row, specials stripped,
        left-aligned, zero-filled; counts [N]).  Two rows give the same peptide string iff their letter rows are equal, so
        de-duplication can run on these fixed-width keys and strings are only built for rows that are kept."""
        ids = np.asarray(ids)
        itos = self.TEXT.vocab.itos
        assert all(len(w) == 1 for w in itos[len(SPECIALS):]), 'vectorised form needs one-letter residue tokens'
        lut = np.zeros(len(itos), np.uint8)
        for i in range(len(SPECIALS), len(itos)):
            lut[i] = ord(itos[i])
        keep = ids >= len(SPECIALS)
        order = np.argsort(~keep, axis=1, kind='stable')                     # residues first, original order kept
        letters = np.take_along_axis(np.where(keep, lut[np.clip(ids, 0, len(itos) - 1)], 0), order, 1).astype(np.uint8)
        return letters, keep.sum(1)

    @staticmethod
    def letters_to_peptides(letters, counts):
        """'A C D' strings (idx2sentences(rows, print_special_tokens=False)) from ids_to_letters output."""
        buf = np.full((letters.shape[0], 2 * letters.shape[1]), ord(' '), np.uint8)
        buf[:, 0::2] = letters
        return [buf[i, :max(2 * int(k) - 1, 0)].tobytes().decode('ascii') for i, k in enumerate(counts)]

    def ids_to_peptides(self, ids):
        """idx2sentences(rows, print_special_tokens=False) for an integer array [N,L] (entries < 0 = padding): the decode
        loops hand over arrays, and a per-token python loop over 10^5..10^6 hypotheses costs more than decoding them.
        Residues are single letters, so a row is built as bytes 'A C D' in one vectorised pass and trimmed per row."""
        return self.letters_to_peptides(*self.ids_to_letters(ids))
