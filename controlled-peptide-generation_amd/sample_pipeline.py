"""CLaSS sampling driver on the MI355X path (counterpart of the reference's sample_pipeline.py:129-324).

Round structure and stop rule are the reference's: rounds of `n_samples_per_round` proposals -> z-space rejection ->
decode -> drop duplicates within the round and against earlier rounds -> stop once `n_samples_acc` accepted peptides
exist (:299-322).  Differences, all documented in DESIGN.md:
  * decode mode is selectable ('beam' with beam_size 5 as the reference's decode_from_z :129-139, or 'greedy');
  * `decode_accepted_only=True` classifies first and decodes only accepted z with c fixed to [0,1] (the reference draws
    c ~ Cat(.5,.5) for every proposal, per 1024-chunk; the default array round draws it per proposal from the
    round's counter stream and reports it as column `c`);
  * encodings come from `states_<split>_<iter>.npz` (same field names as the reference's h5: src, z, mu, logvar, label,
    split) or are computed on the fly; modlamp descriptors (H, uH, charge) need modlamp, which is not installed: columns
    are filled when it is importable, skipped otherwise;
  * rounds are ARRAY frames end to end (`sample_round_arrays` / `run_rounds`): z, scores and the stripped residue rows stay
    numpy / device arrays, de-duplication runs on fixed-width residue keys, and the pandas DataFrame (with the reference's
    columns) is built once at the end - the reference builds `tuple(z.tolist())` and a string per proposal per round;
  * multi-GPU (one process per GPU, cpg.dist): every round's proposals are SHARDED by rows across the ranks (counter-based
    device streams: the union over ranks is the stream one rank would draw alone), every rank decodes its rows, the
    round's rows are all-gathered (cpg.dist.allgather_rows), and the de-duplication against earlier rounds (:312-314) and
    the stop rule (:303) then run on the gathered, identical-on-every-rank set.
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


SPLIT_CODE = {'train': 0, 'val': 1, 'test': 2}   # vis/scripts/build_index.py encodes the split as an integer column


def _states_path(split, savepath=None, n_iter=None, ext=None):
    base = os.path.join(savepath or cfg.savepath, 'states_{}_{}'.format(split, n_iter if n_iter is not None else cfg.vae.n_iter))
    if ext is not None:
        return base + ext
    for e in ('.h5', '.npz'):
        if os.path.exists(base + e):
            return base + e
    return base + '.npz'


def _require_h5py(what):
    """The reference's states files are gzip-9 HDF5 (vis/scripts/build_index.py:32-81).  h5py is an optional dependency here: the npz
    form with the same fields is what this package writes by default and what its tests exercise; asking for h5 without h5py is an
    error, never a silent switch of formats."""
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("cannot {}: the h5 form of the states dump needs h5py, which is not installed; use the npz form "
                           "(dump_encodings(fmt='npz'), the default) or install h5py".format(what)) from e
    return h5py


def get_encodings_from_states(query, split, attributes=None, savepath=None, n_iter=None):
    """mu, logvar (float64 tensors) of the dumped encodings whose labels match `query` ({attr: value}); reads the
    reference's `states_<split>_<iter>.h5` (when h5py is importable) or this package's npz with the same fields
    (reference reader: sample_pipeline.py:73-92)."""
    attributes = attributes if attributes is not None else cfg.attributes
    fn = _states_path(split, savepath, n_iter)
    assert os.path.exists(fn), 'need dumped states ({}); run dump_encodings first'.format(fn)
    if fn.endswith('.h5'):
        h5py = _require_h5py('read ' + fn)
        with h5py.File(fn, 'r') as f:
            mu_a, lv_a, lab_a = f['mu'][:], f['logvar'][:], f['label'][:]
    else:
        f = np.load(fn)
        mu_a, lv_a, lab_a = f['mu'], f['logvar'], f['label']
    mu, logvar, lab = torch.from_numpy(mu_a).double(), torch.from_numpy(lv_a).double(), torch.from_numpy(lab_a)
    col = {k: i for i, (k, _) in enumerate(attributes)}
    sel = torch.ones(lab.shape[0], dtype=torch.bool)
    for attr, val in query.items():
        sel &= lab[:, col[attr]] == val
    return mu[sel], logvar[sel]


@torch.no_grad()
def dump_encodings(model, ids, labels, split, savepath, n_iter, batch=4096, fmt=None):
    """The encode pass of vis/scripts/build_index.py:93-118 - `model(batch.text, q_c='classifier', sample_z='max')`, i.e.
    encoder + CNN classifier + teacher-forced decoder with z = mu - and its on-disk schema (:32-81): src int [N,T];
    z, mu, logvar float16 [N,Z]; label int [N,n_attr]; split int [N,1].  Written as a compressed npz with the reference's field names
    (fmt=None / 'npz': the tested interchange), or as the reference's gzip-9 h5 with fmt='h5' - which REQUIRES h5py and raises
    without it (the h5 branch has never run on a box of this build: h5py is absent from the image)."""
    was_training = model.training
    model.eval()
    zs, mus, lvs = [], [], []
    try:
        for chunk in torch.split(ids, batch):
            (mu, lv), (z, c), _ = model(chunk.to(model.device), q_c='classifier', sample_z='max')
            zs.append(z.cpu()), mus.append(mu.cpu()), lvs.append(lv.cpu())
    finally:
        model.train(was_training)
    from cpg import ops
    ops.check_persistent()   # the encoder's persistent launches: a timed-out wait must not end up in a states file
    f16 = lambda ts: torch.cat(ts).numpy().astype(np.float16)
    fields = dict(src=ids.cpu().numpy().astype(np.int64), z=f16(zs), mu=f16(mus), logvar=f16(lvs),
                  label=np.asarray(labels).astype(np.int64).reshape(ids.shape[0], -1),
                  split=np.full((ids.shape[0], 1), SPLIT_CODE.get(split, 0), np.int64))
    os.makedirs(savepath, exist_ok=True)
    if fmt is None:
        fmt = 'npz'      # the TESTED interchange of this package (same field names, dtypes and shapes as the reference's h5)
    if fmt not in ('npz', 'h5'):
        raise ValueError("dump_encodings: fmt must be 'npz' or 'h5', not {!r}".format(fmt))
    fn = _states_path(split, savepath, n_iter, '.' + fmt)
    if fmt == 'h5':
        h5py = _require_h5py('write ' + fn)
        with h5py.File(fn, 'w') as f:
            for k, v in fields.items():
                f.create_dataset(k, data=v, maxshape=(None, None), compression='gzip', compression_opts=9)
    else:
        np.savez_compressed(fn, **fields)
    return fn


def get_encodings_from_dataloader(query, split, model, dataloader, batch=4096):
    """mu, logvar of the loader's sequences in `split` matching `query`, encoded now (reference :45-70: the same
    `model(batch.text, q_c='classifier', sample_z='max')` pass, no states file needed)."""
    ids, _ = dataloader.subset(split, query)
    LOG.info('Start encoding {} samples from dataset'.format(ids.shape[0]))
    mus, lvs = [], []
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for chunk in torch.split(ids, batch):
                (mu, lv), _, _ = model(chunk.to(model.device), q_c='classifier', sample_z='max')
                mus.append(mu.double().cpu()), lvs.append(lv.double().cpu())
    finally:
        model.train(was_training)
    from cpg import ops
    ops.check_persistent()
    return torch.cat(mus, 0), torch.cat(lvs, 0)


def get_encodings(query, split, model=None, dataloader=None):
    if model is not None and dataloader is not None:
        return get_encodings_from_dataloader(query, split, model, dataloader)
    return get_encodings_from_states(query, split)


def fitQ_and_test(QClass, QKwargs, Q_select={}, negative_select={}, model=None, dataloader=None, n_eval=256):
    """Fit Q_xi^a(z) on the encodings selected by `Q_select` and report its cross-entropy on training and held-out encodings next
    to the prior's (reference :95-126).  The NLL estimate walks over at most n_eval points per set (the reference walks over all
    of them in python; the figure is a diagnostic)."""
    from collections import OrderedDict
    from density_modeling import evaluate_nll
    if model is not None and dataloader is not None:
        mu, logvar = get_encodings_from_dataloader(Q_select, 'train,val', model, dataloader)
    else:
        mu, logvar = get_encodings_from_states(query=Q_select, split='train')
    Q = QClass(mu, logvar, **QKwargs)
    LOG.info('Fitted {}  {} on selection {}'.format(QClass.__name__, str(QKwargs), str(Q_select)))
    metrics = OrderedDict()
    for name, split in (('a,tr', 'train'), ('a,hld', 'test')):
        pts = get_encodings(Q_select, split, model, dataloader)
        if pts[0].shape[0]:
            metrics[name] = evaluate_nll(Q, (pts[0][:n_eval], pts[1][:n_eval]))
    return Q, metrics


def build_clfZ(attr, zneg_mu=None, model=None, dataloader=None):
    """Logistic regression between attr=1 and attr=0 encodings of the training split (reference :169-192: labels -1 / 0 / 1 =
    na / neg / pos); host-side, one-off.  build_clfZ(zpos_mu, zneg_mu) with two tensors fits on given encodings."""
    from sklearn.linear_model import LogisticRegression
    if isinstance(attr, str):
        zpos_mu = get_encodings({attr: 1}, 'train', model, dataloader)[0]
        zneg_mu = get_encodings({attr: 0}, 'train', model, dataloader)[0]
    else:
        zpos_mu, attr = attr, '<given encodings>'
    X = torch.cat([zpos_mu, zneg_mu], 0).numpy()
    Y = np.concatenate([np.ones(zpos_mu.shape[0]), np.zeros(zneg_mu.shape[0])])
    clf = LogisticRegression(solver='lbfgs', max_iter=200)
    clf.fit(X, Y)
    LOG.info('Fitted LogReg classifier in z-space, on attr={}.'.format(attr))
    LOG.info('num samples: {} pos, {} neg. train accuracy={:.5f}'.format(zpos_mu.shape[0], zneg_mu.shape[0], clf.score(X, Y)))
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
    """Reference-format round (DataFrame per round, c drawn from the prior per 1024-chunk unless decode_accepted_only):
    kept for RNG-order parity with the reference (`Q.rng = 'numpy'`); the throughput path is sample_round_arrays."""
    import pandas as pd
    samples_z, scores_z, accept_z = Q.rejection_sample(n_samples=n_samples)
    if decode_accepted_only:
        keep = torch.from_numpy(np.nonzero(accept_z)[0])
        samples_z = samples_z[keep]
        scores_z = {k: v[accept_z] for k, v in scores_z.items()}
        accept_z = accept_z[accept_z]
    if samples_z.shape[0] == 0:   # a round may accept nothing
        return pd.DataFrame({'peptide': [], 'z': [], 'accept_z': np.zeros(0, bool), **{k: v[:0] for k, v in scores_z.items()}})
    c = torch.zeros(samples_z.shape[0], 2, device=model.device)
    c[:, 1] = 1.0
    samples = decode_from_z(samples_z, model, dataset, sample_mode=sample_mode, c=c if decode_accepted_only else None)
    return pd.DataFrame({'peptide': samples, 'z': list(samples_z.numpy()), 'accept_z': accept_z, **scores_z})


def one_sampling_round(model, dataset, Q, n_samples_per_round, **kw):
    df = compute_modlamp(get_new_samples(model, dataset, Q, n_samples_per_round, **kw))
    df['accept'] = df['accept_z']
    return df


# ------------------------------------------------------------------------------------------------ array rounds
def decode_ids_from_z(z, c, model, sample_mode='beam', beam_size=5, chunk=65536):
    """Device z [n,Z], c [n,2] -> (DEVICE ids int16 [n, T+1] with -1 padding: best beam hypothesis / greedy row incl. <start>,
    decoder row-step evaluations the decode needed).  Decoding is per-z independent, so the chunk size is a
    memory knob only."""
    from cpg import decode as cdecode
    T = model.MAX_SEQ_LEN
    out = torch.full((z.shape[0], T + 1), -1, dtype=torch.int16, device=z.device)
    evals = 0
    was_training = model.training
    model.eval()
    try:
        for i0 in range(0, z.shape[0], chunk):
            zz, cc = z[i0:i0 + chunk].float().contiguous(), c[i0:i0 + chunk].float().contiguous()
            if sample_mode == 'beam':
                hyps, lens, _ = cdecode.decode_beam_arrays(model.decoder, zz, cc, T, beam_size, 3, 1, device_out=True)
                best = hyps[:, 0, :]
                w = best.shape[1]
                live = torch.arange(w, device=z.device)[None, :] < lens[:, 0:1]
                out[i0:i0 + zz.shape[0], :w] = torch.where(live, best, torch.full_like(best, -1)).to(torch.int16)
                evals += int(beam_size) * int(cdecode.LAST_BEAM_STEPS)
            elif sample_mode == 'greedy':
                ids = cdecode.decode_hard(model.decoder, zz, cc, T)
                out[i0:i0 + zz.shape[0], :ids.shape[1]] = ids.to(torch.int16)
                evals += int(cdecode.LAST_GREEDY_STEPS)
            else:
                raise ValueError('array rounds decode with beam or greedy')
    finally:
        model.train(True)  # generate_sentences always leaves the model in train mode
    return out, evals


N_SPECIALS = 4   # <unk> <pad> <start> <eos> (models/mutils.py:5-8): ids below this are stripped from a peptide string


def residue_rows(ids, n_vocab):
    """ids int [n, L] (device or CPU tensor; < 0 = padding) -> (letters uint8 [n, L]: ids of the residues of each row, specials
    stripped, left-aligned, zero-filled; counts int32 [n]).  The compaction idx2sentences(..., print_special_tokens=False)
    does per row in python, as three tensor ops on the device the decode left the ids on.  Two rows give the same peptide
    string iff their letter rows are equal."""
    if ids.is_cuda and ids.shape[0] > 0:   # one pass on the device (cpg_residue_rows); the tensor-op form below serves host arrays
        from cpg import ops
        ids16 = ids.to(torch.int16).contiguous()
        letters = torch.empty(ids16.shape, dtype=torch.uint8, device=ids.device)
        counts = torch.empty(ids16.shape[0], dtype=torch.int32, device=ids.device)
        ops.call("cpg_residue_rows", ops._p(ids16), ids16.shape[0], ids16.shape[1], N_SPECIALS, ops._p(letters), ops._p(counts),
                 ops._stream())
        return letters, counts
    ids = ids.to(torch.int64)
    keep = ids >= N_SPECIALS
    pos = torch.cumsum(keep, 1) - 1
    letters = torch.zeros(ids.shape[0], ids.shape[1] + 1, dtype=torch.uint8, device=ids.device)
    letters.scatter_(1, torch.where(keep, pos, torch.full_like(pos, ids.shape[1])), torch.where(keep, ids, torch.zeros_like(ids)).to(torch.uint8))
    return letters[:, :ids.shape[1]].contiguous(), keep.sum(1).to(torch.int32)


def sample_round_arrays(model, dataset, Q, n_samples, sample_mode='beam', decode_accepted_only=False, shard=(0, 1)):
    """One sampling round (reference get_new_samples :195-207) as a frame of DEVICE tensors; with shard=(rank, world) this
    rank proposes / scores / decodes its rows of the round only."""
    z, probs, accum, acc = Q.rejection_sample(n_samples, return_device=True, shard=shard)
    names = Q.score_names()
    n_prop = z.shape[0]
    if decode_accepted_only:
        keep = torch.nonzero(acc).squeeze(1)
        z, probs, accum, acc = z[keep], probs[:, keep], accum[keep], acc[keep]
    T = model.MAX_SEQ_LEN
    if z.shape[0] == 0:
        ids, evals = torch.full((0, T + 1), -1, dtype=torch.int16, device=z.device), 0
    else:
        c = torch.zeros(z.shape[0], 2, device=z.device)
        if decode_accepted_only:
            c[:, 1] = 1.0      # documented difference: the accepted-only form fixes c = [0,1] (module docstring)
        else:
            # decode_from_z -> generate_sentences(z) with c=None: c ~ Cat(.5,.5) per proposal (reference models/model.py:121-126,
            # 208-209; it draws per 1024-chunk with numpy).  One Bernoulli per ROW of the round's counter stream, so a rank's
            # shard carries exactly the c's those rows have in the single-rank round.
            rank, world = shard
            row0 = rank * (n_samples // world) if world > 1 else 0
            seed, off = Q._next_philox(n_samples)
            from cpg import ops
            bit = ops.rng_bernoulli((z.shape[0],), 0.5, seed, off + row0 // 4, z.device).long()
            c.scatter_(1, bit.unsqueeze(1), 1.0)
        ids, evals = decode_ids_from_z(z, c, model, sample_mode)
    letters, n_res = residue_rows(ids, dataset.n_vocab)
    frame = {'letters': letters, 'n_res': n_res, 'z': z, 'accept_z': acc.to(torch.bool), names[0]: accum}
    frame['c'] = c.argmax(1).to(torch.uint8) if z.shape[0] else torch.zeros(0, dtype=torch.uint8, device=z.device)
    for i, nm in enumerate(names[1:]):
        frame[nm] = probs[i]
    return frame, dict(proposed=n_prop, decoded=int(z.shape[0]), decoder_evals=evals)


def gather_frame(frame):
    """All-gather of a round's rows over the ranks (cpg.dist.allgather_rows: counts first, then padded payload)."""
    from cpg import dist as cdist
    import torch.distributed as tdist
    if not (tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1):
        return frame
    on_gpu = tdist.get_backend() == 'nccl'
    out = {}
    for k, t in frame.items():
        as_u8 = t.dtype == torch.bool
        t = t.to(torch.uint8) if as_u8 else t
        g = cdist.allgather_rows(t.contiguous() if on_gpu else t.cpu().contiguous())
        out[k] = g.to(torch.bool) if as_u8 else g
    return out


def _keys(letters, n_vocab):
    """Residue rows (uint8 ids, 0 = past the end) -> [n, W] int64 keys: row equality = key equality.  The key form is a function of
    the RUN - vocabulary size and row width - never of a round's contents, so every round of a run (and every rank) keys alike
    (round-5 advisor finding: a data-dependent choice let rounds disagree).  n_vocab <= 25 and rows of up to 26 residues pack 13 ids
    to a word in base 25 (25^13 < 2^63: the usual 20-residue vocabulary with its 4 specials fits a row into TWO words - two sorts in
    dedup_frame instead of one per 8 letters); wider vocabularies / longer rows take 8 letters (bytes) per word."""
    n, L = letters.shape
    dev = letters.device
    if L <= 26 and int(n_vocab) <= 25:
        pad = torch.zeros(n, 26, dtype=torch.int64, device=dev)
        pad[:, :L] = letters
        w = (25 ** torch.arange(13, device=dev, dtype=torch.int64))
        return torch.stack([(pad[:, :13] * w).sum(1), (pad[:, 13:] * w).sum(1)], 1)
    W = -(-L // 8)
    pad = torch.zeros(n, W * 8, dtype=torch.uint8, device=dev)
    pad[:, :L] = letters
    return pad.view(torch.int64).reshape(n, W)


def dedup_frame(frame, seen, n_vocab=24):
    """drop_duplicates within the round (first occurrence kept, original order) and against earlier rounds (reference
    :312-314), on the stripped residue rows, on the device the frame lives on.  seen: key tensor of every row kept so far;
    returns (frame, new seen).
    Exact, sort-based: the rows' keys (seen rows first) are ordered by one STABLE sort per key word, last word first - rows with
    equal keys end up adjacent IN THEIR ORIGINAL ORDER, so the first row of every run is the occurrence drop_duplicates keeps; a
    round of 10^6 rows costs two 10^6-element sorts (torch.unique(dim=0) + scatter-min before: 11 ms of a 177 ms round)."""
    keys = _keys(frame['letters'], n_vocab)
    n = keys.shape[0]
    if n == 0:
        return frame, seen
    if seen is not None and seen.shape[1] != keys.shape[1]:      # a key form changed between rounds (longer rows): re-key is not possible
        raise ValueError("dedup_frame: rounds disagree on the key width (%d vs %d words)" % (seen.shape[1], keys.shape[1]))
    n_seen = 0 if seen is None else seen.shape[0]
    allk = keys if n_seen == 0 else torch.cat([seen, keys], 0)
    perm = torch.arange(allk.shape[0], device=keys.device)
    for wd in range(allk.shape[1] - 1, -1, -1):                   # LSD order: stable sorts by word W-1, ..., 0
        _, idx = torch.sort(allk[perm, wd], stable=True)
        perm = perm[idx]
    ks = allk[perm]
    first = torch.ones(allk.shape[0], dtype=torch.bool, device=keys.device)
    first[1:] = (ks[1:] != ks[:-1]).any(1)
    mask = torch.zeros(allk.shape[0], dtype=torch.bool, device=keys.device)
    mask[perm[first]] = True                                      # first occurrence of every distinct row, by original index
    keep = torch.nonzero(mask[n_seen:]).squeeze(1)                # this round's rows that are new (ascending = original order)
    out = {k: v[keep] for k, v in frame.items()}
    kept = keys[keep]
    return out, (kept if n_seen == 0 else torch.cat([seen, kept], 0))


def _peptide_column(letters, n_res, itos):
    """Residue rows (DEVICE uint8 [n, L] ids, int32 [n] counts) -> the 'A C D' peptide strings of
    idx2sentences(..., print_special_tokens=False) as a pandas column.  The characters, the separating blanks and the
    compaction into one flat byte buffer + offsets happen on the device; with pyarrow the column is built over that buffer
    without creating a python string per row (176 k kept rows of a 1 M-proposal round: 1 ms instead of 50-130 ms), otherwise
    by slicing one decoded string."""
    import pandas as pd
    n, L = letters.shape
    lut = torch.zeros(256, dtype=torch.uint8)
    for i in range(N_SPECIALS, len(itos)):
        assert len(itos[i]) == 1, 'vectorised form needs one-letter residue tokens'
        lut[i] = ord(itos[i])
    dev = letters.device
    buf = torch.full((n, 2 * L), ord(' '), dtype=torch.uint8, device=dev)
    buf[:, 0::2] = lut.to(dev)[letters.long()]
    lens = (2 * n_res.long() - 1).clamp_(min=0)
    flat = buf[torch.arange(2 * L, device=dev)[None, :] < lens[:, None]].cpu().numpy()
    off = np.zeros(n + 1, np.int64)
    off[1:] = torch.cumsum(lens, 0).cpu().numpy()
    if os.environ.get('CPG_ARROW_STRINGS', '1') != '0' and off[-1] < 2 ** 31:
        try:
            import pyarrow as pa
            arr = pa.StringArray.from_buffers(n, pa.py_buffer(off.astype(np.int32)), pa.py_buffer(flat))
            return pd.Series(arr, dtype=pd.ArrowDtype(pa.string()))
        except (ImportError, AttributeError):
            pass
    text, o = flat.tobytes().decode('ascii'), off.tolist()
    return [text[o[i]:o[i + 1]] for i in range(n)]


def _to_host(tensors):
    """Device tensors -> numpy arrays.  (A pinned staging buffer was measured: 1.5 ms instead of 7.4 ms for the 75 MB of z rows of a
    1 M-proposal round ONCE the block is cached, but its first use page-locks the block - 10-25 ms - and the table is built once per
    run; plain copies are the faster form here.)"""
    return {k: v.cpu().numpy() for k, v in tensors.items()}


def _z_column(z):
    """The table's `z` column: one latent vector per row.  The reference builds a python list of arrays (an object column: one numpy
    view per row - 177 k objects, ~20 ms of a 1 M-proposal round); with pyarrow the same rows are ONE fixed-size-list array over the
    [n, Z] buffer (no per-row object; `df['z'][i]` is the row).  CPG_ARROW_STRINGS=0 keeps the object column."""
    if os.environ.get('CPG_ARROW_STRINGS', '1') != '0':
        try:
            import pandas as pd
            import pyarrow as pa
            flat = pa.array(np.ascontiguousarray(z).reshape(-1))
            return pd.Series(pa.FixedSizeListArray.from_arrays(flat, int(z.shape[1])), dtype=pd.ArrowDtype(pa.list_(flat.type, int(z.shape[1]))))
        except (ImportError, AttributeError, TypeError):
            pass
    return list(z)


def frames_to_dataframe(frames, dataset):
    """The reference's sample table (peptide, z, accept_z, clfZ_*, accept) from the kept frames."""
    import pandas as pd
    if not frames:
        return pd.DataFrame(columns=['peptide', 'z', 'accept_z', 'accept'])
    dev = {k: (frames[0][k] if len(frames) == 1 else torch.cat([f[k] for f in frames], 0)) for k in frames[0]}
    peptide = _peptide_column(dev.pop('letters'), dev.pop('n_res'), dataset.TEXT.vocab.itos)
    cat = _to_host(dev)
    df = pd.DataFrame({'peptide': peptide, 'z': _z_column(cat['z']), 'accept_z': cat['accept_z'].astype(bool),
                       **{k: v for k, v in cat.items() if k not in ('z', 'accept_z')}})
    df = compute_modlamp(df)
    df['accept'] = df['accept_z']
    return df


def run_rounds(model, dataset, Q, n_samples_per_round, n_samples_acc, max_rounds=1000, sample_mode='beam',
               decode_accepted_only=False, return_stats=False):
    """Rounds until n_samples_acc distinct accepted peptides exist (reference main loop :299-322).  Array frames when the
    loader offers `ids_to_letters` and the proposal draws on the device (the multi-GPU form); otherwise the reference-order
    DataFrame rounds."""
    from cpg import dist as cdist
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
    rank = tdist.get_rank() if world > 1 else 0
    arrays = hasattr(dataset, 'ids_to_letters') and sample_mode in ('beam', 'greedy') and (Q.rng == 'device' or world > 1)
    stats = dict(rounds=0, proposed=0, decoded=0, decoder_evals=0, kept=0, accepted=0)
    if not arrays:
        assert world == 1, 'reference-order rounds are single-process'
        import pandas as pd
        samples = pd.DataFrame(columns=['peptide', 'accept', 'accept_z'])

        def finished(df):
            return len(df) >= n_samples_acc and df['accept'].sum() >= n_samples_acc
        while not finished(samples) and stats['rounds'] < max_rounds:
            stats['rounds'] += 1
            LOG.info("Round #{}".format(stats['rounds']))
            new = one_sampling_round(model, dataset, Q, n_samples_per_round, sample_mode=sample_mode,
                                     decode_accepted_only=decode_accepted_only)
            new = new.loc[new.peptide.drop_duplicates().index]
            new = new[~new['peptide'].isin(samples['peptide'])]
            samples = pd.concat([samples, new], ignore_index=True, sort=False)
            LOG.info('Q_xi(z|a) rejection sampling acceptance rate: {}/{}'.format(samples['accept_z'].sum(), len(samples)))
        stats.update(kept=len(samples), accepted=int(samples['accept_z'].sum()) if len(samples) else 0)
        return (samples, stats) if return_stats else samples
    frames, seen = [], None
    while not (stats['kept'] >= n_samples_acc and stats['accepted'] >= n_samples_acc) and stats['rounds'] < max_rounds:
        stats['rounds'] += 1
        LOG.info("Round #{}".format(stats['rounds']))
        frame, st = sample_round_arrays(model, dataset, Q, n_samples_per_round, sample_mode, decode_accepted_only, (rank, world))
        frame = {k: torch.as_tensor(v) for k, v in frame.items()}
        frame = gather_frame(frame)            # identical on every rank from here on
        frame, seen = dedup_frame(frame, seen, dataset.n_vocab)
        frames.append(frame)
        for k in ('proposed', 'decoded', 'decoder_evals'):
            stats[k] += st[k] * world if world > 1 else st[k]   # ranks do equal shares (decoded: this rank's count scaled)
        stats['kept'] += len(frame['accept_z'])
        stats['accepted'] += int(frame['accept_z'].sum())
        LOG.info('Q_xi(z|a) rejection sampling acceptance rate: {}/{}'.format(stats['accepted'], stats['kept']))
    samples = frames_to_dataframe(frames, dataset)
    return (samples, stats) if return_stats else samples


def _write_table(table, stem):
    table.drop(columns='z').to_csv(stem + '.csv', index_label='idx')
    if not (len(table) == 0 or table['z'].dtype == object):
        # the pickled table is the reference's interchange file (:149-160): its consumers expect an object column of numpy rows, not the
        # Arrow list column the in-memory table carries (round-5 advisor finding).  One numpy view per row, at write time only.
        table = table.copy(deep=False)
        pa_arr = getattr(table['z'].array, '_pa_array', None)
        z = (pa_arr.combine_chunks().flatten().to_numpy(zero_copy_only=False) if pa_arr is not None
             else np.asarray(table['z'].tolist(), dtype=np.float32))
        z = z.reshape(len(table), -1)
        col = np.empty(len(table), dtype=object)
        for i in range(len(table)):
            col[i] = z[i]
        table['z'] = col
    table.to_pickle(stem + '.pkl')


def save_samples(samples, basedir, fn_prefix):
    """The reference's output set (:149-160): <prefix>_<date>.plain.txt / .csv / .pkl with every kept sample, and
    <prefix>_<date>.accepted.<n>.csv / .pkl with the accepted ones (csv without the z column)."""
    stem = '{}_{}'.format(os.path.join(basedir, fn_prefix), datetime.date.today().isoformat())
    os.makedirs(basedir, exist_ok=True)
    with open(stem + '.plain.txt', 'w') as fh:
        fh.write(samples['peptide'].to_string(index=False))
    _write_table(samples, stem)
    LOG.info('Full sample list written to {}.pkl/csv'.format(stem))
    accepted = samples[samples['accept'].astype(bool)]
    acc_stem = '{}.accepted.{}'.format(stem, len(accepted))
    _write_table(accepted, acc_stem)
    LOG.info('Accepted sample list written to {}.pkl/csv'.format(acc_stem))
    return stem


def main(args):
    """Reference main :236-324: load the trained model + vocabulary of the run (api.get_model_and_vocab_path /
    load_trained_model - a missing or mismatched checkpoint is an error), fit Q_xi^a(z) and the amp / tox z-space classifiers
    on the dumped states (main.py --phase 1 writes them) or straight from the loader (--Q_from_full_dataloader 1), sample in
    rounds until n_samples_acc accepted peptides exist, write the sample lists."""
    import json
    from api import Vocab, get_model_and_vocab_path, get_result_for_model, load_trained_model
    from cpg import dist as cdist
    from cpg import ops
    from cpg.synth import SyntheticPeptideLoader
    world, rank, local = cdist.init()   # one process per GPU (torch.distributed.run); world 1 = plain run
    torch.cuda.set_device(cdist.local_device(local))
    device = torch.device('cuda', cdist.local_device(local))
    model_path, vocab_path, _ = get_model_and_vocab_path()
    LOG.info('Load model, vocab, dataloader.')
    vocab = Vocab(vocab_path)
    model = load_trained_model(model_path, vocab.size(), device=device)
    LOG.info('Loaded model succesfully.')
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    dataset = SyntheticPeptideLoader(cfg.vae.batch_size, cfg.max_seq_len, device, size=cfg.hw.synthetic_size, seed=cfg.seed)
    assert dataset.n_vocab == vocab.size(), 'vocab.dict of the run does not match the loader'
    LOG.info('Model metrics: {}'.format(get_result_for_model(model_path, print_results=False)))
    LOG.info('Fit attribute-conditioned marginal posterior Q_xi^a(z)')
    qkw = dict(Q_KWARGS)
    for k in qkw:
        if hasattr(args, 'Q_' + k):
            qkw[k] = getattr(args, 'Q_' + k)
    select, negative = ({'amp': 1}, {'amp': 0}) if getattr(args, 'Q_select_amppos', 0) else ({}, {})
    live = bool(getattr(args, 'Q_from_full_dataloader', 0))
    Q, q_metrics = fitQ_and_test(Q_CLASS, qkw, select, negative, model if live else None, dataset if live else None)
    Q.device = device
    Q._upload()
    LOG.info('Q Fit metrics: ')
    print(json.dumps(q_metrics, indent=4))
    z_clfs = {attr: build_clfZ(attr, model=model if live else None, dataloader=dataset if live else None) for attr in ['amp', 'tox']}
    Q.init_attr_classifiers(z_clfs, clf_targets={'amp': 1, 'tox': 0})
    if world > 1 or getattr(args, 'device_rng', False):
        Q.rng = 'device'   # counter-based streams: rounds shard by rows across the ranks
    samples = run_rounds(model, dataset, Q, args.n_samples_per_round, args.n_samples_acc, sample_mode=getattr(args, 'sample_mode', 'beam'),
                         decode_accepted_only=getattr(args, 'decode_accepted_only', False))
    ops.check_persistent()
    if rank == 0:
        save_samples(samples, cfg.savepath, args.samples_outfn_prefix)
    return samples


def build_parser():
    parser = argparse.ArgumentParser(argument_default=argparse.SUPPRESS, description='Override config float & string values')
    cfg._cfg_import_export(parser, cfg, mode='fill_parser')
    parser.add_argument('--QClass', default='mogQ')
    parser.add_argument('--Q_n_components', type=int, default=100)
    parser.add_argument('--Q_covariance_type', default='diag')
    parser.add_argument('--Q_select_amppos', type=int, default=0)
    parser.add_argument('--Q_from_full_dataloader', type=int, default=0)
    parser.add_argument('--n_samples_per_round', type=int, default=5000)
    parser.add_argument('--n_samples_acc', type=int, default=100)
    parser.add_argument('--samples_outfn_prefix', default='samples')
    parser.add_argument('--sample_mode', default='beam')
    parser.add_argument('--decode_accepted_only', action='store_true', default=False)
    parser.add_argument('--device_rng', action='store_true', default=False,
                        help="draw proposals with the on-device counter streams (always on with more than one rank)")
    return parser


def run(argv=None):
    """`python sample_pipeline.py [--<cfg.key> value] [--Q_* ...]`: parse, apply the cfg overrides like main.py, sample."""
    a = build_parser().parse_args(argv)
    cfg._override_config(a, cfg)
    cfg._update_cfg()
    cfg._print(cfg)
    return main(a)


if __name__ == "__main__":
    LOG.info("Sample pipeline. Fit Q_xi(z), Sample from it, score samples.")
    run()
