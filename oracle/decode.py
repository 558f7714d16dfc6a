"""Autoregressive decoding in numpy float32 (oracle; test infrastructure only).

  GRUDecoder.forward_sample   models/decoder.py:86-109   (one decode step = one "decoder eval")
  RNN_VAE.sample_G greedy     models/model.py:225-363    (argmax; finished rows forced to PAD; early break)
  beam mode                   models/model.py:258-276,314-328,364-376 ; models/Beam.py:15-132
"""
import numpy as np

from .gru import gru_cell_fwd, F32

UNK, PAD, START, EOS = 0, 1, 2, 3


def n_dec_layers(P):
    """Layers of the decoder RNN in a state dict (1 in the reference, models/decoder.py:40-41; > 1 = this build's extension)."""
    L = 1
    while f"decoder.rnn.weight_hh_l{L}" in P:
        L += 1
    return L


def init_state(P, zc):
    """Decoder state before the first step: [z;c] (models/decoder.py:77); [L,N,H] with every layer = [z;c] for the multi-layer
    EXTENSION (torch.nn.GRU(num_layers=L) semantics; not a reference component - parity unpinned)."""
    L = n_dec_layers(P)
    return zc.copy() if L == 1 else np.stack([zc] * L)


def _upper_layers(P, x, h, cst=None):
    """Layers 1..L-1 of the multi-layer extension on layer 0's new state x: h [L,N,H] (slot 0 already updated by the caller is
    ignored), returns (top output, new h [L,N,H], new c)."""
    from .lstm import lstm_cell_fwd
    hn, cn = [x], [None if cst is None else cst[0]]
    for l in range(1, h.shape[0]):
        gi = (x @ P[f"decoder.rnn.weight_ih_l{l}"].T + P[f"decoder.rnn.bias_ih_l{l}"]).astype(F32)
        if cst is None:
            x, _ = gru_cell_fwd(gi, h[l], P[f"decoder.rnn.weight_hh_l{l}"], P[f"decoder.rnn.bias_hh_l{l}"])
        else:
            x, cl, _ = lstm_cell_fwd(gi, h[l], cst[l], P[f"decoder.rnn.weight_hh_l{l}"], P[f"decoder.rnn.bias_hh_l{l}"])
            cn.append(cl)
        hn.append(x)
    return x, np.stack(hn), (None if cst is None else np.stack(cn))


def decoder_step(P, tok, zc, h, keep=None, p_out=0.3):
    """logits [N,V], h' [N,H] for current tokens tok [N] (GRUDecoder.forward_sample, models/decoder.py:86-109).
    Multi-layer extension: h is [L,N,H] in and out (see init_state).
    Skip connections when the model has them (:103-105): output := skip_weight_x(output) + skip_weight_z([z;c]).
    keep (optional 0/1 [N,H]): the out-dropout mask of a step sampled in TRAIN mode (generate_sentences(eval_mode=False),
    models/model.py:216-221: nn.Dropout(p_out) in front of the vocabulary projection is then live); eval mode: None."""
    x = np.concatenate([P["word_emb.weight"][tok], zc], 1).astype(F32)
    gi = (x @ P["decoder.rnn.weight_ih_l0"].T + P["decoder.rnn.bias_ih_l0"]).astype(F32)
    if h.ndim == 3:
        h0n, _ = gru_cell_fwd(gi, h[0], P["decoder.rnn.weight_hh_l0"], P["decoder.rnn.bias_hh_l0"])
        out, h, _ = _upper_layers(P, h0n, h)
    else:
        h, _ = gru_cell_fwd(gi, h, P["decoder.rnn.weight_hh_l0"], P["decoder.rnn.bias_hh_l0"])
        out = h
    if "decoder.skip_weight_x.weight" in P:
        out = ((out @ P["decoder.skip_weight_x.weight"].T).astype(F32) + (zc @ P["decoder.skip_weight_z.weight"].T).astype(F32)).astype(F32)
    if keep is not None:
        out = (out * (keep.astype(F32) * F32(1.0 / (1.0 - p_out)))).astype(F32)
    logits = (out @ P["decoder.fc.1.weight"].T + P["decoder.fc.1.bias"]).astype(F32)
    return logits, h


def lstm_decoder_step(P, tok, zc, h, cst):
    """The LSTM extension's decode step (torch.nn.LSTM semantics; NOT a reference component - parity unpinned)."""
    from .lstm import lstm_cell_fwd
    x = np.concatenate([P["word_emb.weight"][tok], zc], 1).astype(F32)
    gi = (x @ P["decoder.rnn.weight_ih_l0"].T + P["decoder.rnn.bias_ih_l0"]).astype(F32)
    if h.ndim == 3:   # multi-layer extension: h, cst [L,N,H]
        h0n, c0n, _ = lstm_cell_fwd(gi, h[0], cst[0], P["decoder.rnn.weight_hh_l0"], P["decoder.rnn.bias_hh_l0"])
        cst = cst.copy()
        cst[0] = c0n
        out, h, cst = _upper_layers(P, h0n, h, cst)
    else:
        h, cst, _ = lstm_cell_fwd(gi, h, cst, P["decoder.rnn.weight_hh_l0"], P["decoder.rnn.bias_hh_l0"])
        out = h
    logits = (out @ P["decoder.fc.1.weight"].T + P["decoder.fc.1.bias"]).astype(F32)
    return logits, h, cst


def greedy(P, z, c, max_len, prevent_empty=False, min_length=1, return_logits=False, out_keep=None, p_out=0.3, cell="gru"):
    """ids [N, 1+steps] int64 with column 0 = START; steps <= max_len (stops once every row has emitted EOS).
    out_keep (optional [steps,N,H]): per-step out-dropout masks of a train-mode decode (decoder_step)."""
    N = z.shape[0]
    zc = np.concatenate([z, c], 1).astype(F32)
    h = init_state(P, zc)
    cst = np.zeros_like(h)
    tok = np.full(N, START, np.int64)
    finished = np.zeros(N, bool)
    cols, all_logits = [tok], []
    for i in range(max_len):
        if cell == "lstm":   # the LSTM extension (torch.nn.LSTM semantics, c0 = 0); not a reference component
            logits, h, cst = lstm_decoder_step(P, tok, zc, h, cst)
        else:
            logits, h = decoder_step(P, tok, zc, h, None if out_keep is None else out_keep[i], p_out)
        if prevent_empty and i == 0:
            neg = F32(-2.0) * np.abs(logits.min())
            logits[:, [PAD, START, EOS]] = neg
        all_logits.append(logits.copy())
        tok = logits.argmax(1).astype(np.int64)
        tok[finished] = PAD
        finished |= tok == EOS
        cols.append(tok)
        if finished.all() and len(cols) >= min_length:
            break
    ids = np.stack(cols, 1)
    return (ids, np.stack(all_logits, 1)) if return_logits else ids


def categorical(P, z, c, max_len, uniforms, temp=1.0, prevent_empty=False, min_length=1):
    """RNN_VAE.sample_G 'categorical' (models/model.py:308-309,350-353,362-363) with the uniform draw of every (step, row)
    given: torch.distributions.Categorical(logits=logits/temp).sample() = torch.multinomial(softmax, 1), which picks the first
    index whose cumulative probability exceeds its uniform.  uniforms [steps, N] float64."""
    N = z.shape[0]
    zc = np.concatenate([z, c], 1).astype(F32)
    h = zc.copy()
    tok = np.full(N, START, np.int64)
    finished = np.zeros(N, bool)
    cols = [tok]
    for i in range(max_len):
        logits, h = decoder_step(P, tok, zc, h)
        if prevent_empty and i == 0:
            logits[:, [PAD, START, EOS]] = F32(-2.0) * np.abs(logits.min())
        x = (logits / F32(temp)).astype(F32)
        p = np.exp((x - x.max(1, keepdims=True)).astype(F32)).astype(np.float64)
        cum = np.cumsum(p, 1)
        thr = uniforms[i].astype(np.float64) * cum[:, -1]
        tok = np.minimum((cum <= thr[:, None]).sum(1), p.shape[1] - 1).astype(np.int64)
        tok[finished] = PAD
        finished |= tok == EOS
        cols.append(tok)
        if finished.all() and len(cols) >= min_length:
            break
    return np.stack(cols, 1)


def _log_softmax(x):
    m = x.max(1, keepdims=True)
    return (x - (m + np.log(np.exp(x - m).sum(1, keepdims=True)))).astype(F32)


class _Beam:
    """Per-sentence beam bookkeeping (restates models/Beam.py)."""

    def __init__(self, size, n_best, min_length):
        self.size, self.n_best, self.min_length = size, n_best, min_length
        self.scores = np.zeros(size, F32)
        self.prev_ks = []
        first = np.full(size, PAD, np.int64)
        first[0] = START
        self.next_ys = [first]
        self.finished = []
        self.eos_top = False
        self.min_margin = np.inf   # smallest gap between neighbours of the ranked candidates, down to the first one NOT taken

    def done(self):
        return self.eos_top and len(self.finished) >= self.n_best

    def advance(self, logp):
        V = logp.shape[1]
        logp = logp.copy()
        if len(self.next_ys) < self.min_length:
            logp[:, EOS] = F32(-1e20)
        logp[:, START] = F32(-1e20)  # never predict BOS (Beam.py:72)
        if self.prev_ks:
            cand = (logp + self.scores[:, None]).astype(F32)
            cand[self.next_ys[-1] == EOS] = F32(-1e20)  # EOS has no children (Beam.py:78-80)
        else:
            cand = logp[0:1]  # first step: only beam 0 is live
        flat = cand.reshape(-1)
        order = np.argsort(-flat, kind="stable")[: self.size + 1]
        top = flat[order].astype(np.float64)
        live = top > -1e19                   # masked candidates (-1e20) tie with each other by construction: not a numeric tie
        gaps = [top[i] - top[i + 1] for i in range(len(top) - 1) if live[i]]
        if gaps:
            self.min_margin = min(self.min_margin, float(min(gaps)))
        order = order[: self.size]
        self.scores = flat[order].astype(F32)
        prev = order // V
        self.prev_ks.append(prev)
        self.next_ys.append(order - prev * V)
        for i in range(self.size):
            if self.next_ys[-1][i] == EOS:
                self.finished.append((self.scores[i], len(self.next_ys) - 1, i))
        if self.next_ys[-1][0] == EOS:
            self.eos_top = True

    def best(self):
        fin = list(self.finished)
        i = 0
        while len(fin) < self.n_best:  # pad from the live beam (Beam.py:111-117)
            fin.append((self.scores[i], len(self.next_ys) - 1, i))
            i += 1
        fin.sort(key=lambda a: -a[0])  # stable, raw summed log-prob, no length norm
        hyps, scores = [], []
        for s, t, k in fin[: self.n_best]:
            hyp = []
            for j in range(len(self.prev_ks[:t]) - 1, -2, -1):
                hyp.append(int(self.next_ys[j + 1][k]))
                k = self.prev_ks[j][k] if j >= 0 else k
            hyps.append(hyp[::-1])
            scores.append(float(s))
        return hyps, scores


def beam(P, z, c, max_len, beam_size=5, n_best=3, min_length=1, return_history=False, cell="gru", return_margins=False,
         out_keep=None, p_out=0.3):
    """Returns (hyps, scores): hyps[i][j] = token list incl. leading START.  cell='lstm': the LSTM extension's decoder
    (h0 = [z;c], c0 = 0), the cell state reordered by the same back-pointers.
    return_history adds (tok, prev, score) arrays [steps,N,K] (tok=-1 where a sentence was not advanced): the record the
    device beam kernel keeps, used to test the host-side hypothesis reconstruction."""
    N = z.shape[0]
    zc1 = np.concatenate([z, c], 1).astype(F32)
    zc = np.tile(zc1, (beam_size, 1))  # beam-major [beam*N] (model.py:262-263)
    h = init_state(P, zc)
    cst = np.zeros_like(h)
    beams = [_Beam(beam_size, n_best, min_length) for _ in range(N)]
    tok = np.stack([b.next_ys[-1] for b in beams]).T.reshape(-1)
    hist = []
    for step in range(max_len):
        if cell == "lstm":
            logits, h, cst = lstm_decoder_step(P, tok, zc, h, cst)
        else:   # out_keep [steps, beam*N, H]: train-mode decode, rows beam-major like the states
            logits, h = decoder_step(P, tok, zc, h, None if out_keep is None else out_keep[step], p_out)
        lg = logits.reshape(beam_size, N, -1)
        nl = h.shape[0] if h.ndim == 3 else 1
        hv = h.reshape(nl, beam_size, N, -1)       # views: the reorder below writes through to h / cst (every layer's state)
        cv = cst.reshape(nl, beam_size, N, -1)
        ht = np.full((N, beam_size), -1, np.int64)
        hp = np.zeros((N, beam_size), np.int64)
        hsc = np.zeros((N, beam_size), F32)
        hist.append((ht, hp, hsc))
        for j, b in enumerate(beams):
            if not b.done():
                b.advance(_log_softmax(lg[:, j]))
                ht[j], hp[j], hsc[j] = b.next_ys[-1], b.prev_ks[-1], b.scores
            hv[:, :, j] = hv[:, b.prev_ks[-1], j]  # _update_hidden (model.py:387-404), applied even when done
            cv[:, :, j] = cv[:, b.prev_ks[-1], j]
        tok = np.stack([b.next_ys[-1] for b in beams]).T.reshape(-1)
        if all(b.done() for b in beams):
            break
    out = [b.best() for b in beams]
    if return_margins:
        # per sentence: the smallest score gap any of its top-k selections rested on (a gap below float32 resolution of the
        # scores means either order is a correct evaluation: tests allow a differing hypothesis set only there)
        return [o[0] for o in out], [o[1] for o in out], np.array([b.min_margin for b in beams])
    if return_history:
        return [o[0] for o in out], [o[1] for o in out], tuple(np.stack([h[i] for h in hist]) for i in range(3))
    return [o[0] for o in out], [o[1] for o in out]


def soft_sample(P, z, c, max_len, mode, temp=1.0, min_length=1, sampled=None, cell="gru"):
    """RNN_VAE.sample_G soft modes, models/model.py:337-359 with decoder.forward_sample's soft branch (decoder.py:87-89) and
    mutils.soft_embed (:39-45).  mode: 'none_softmax' | 'greedy_softmax' | 'categorical_softmax' (the hard draws of the
    latter are passed in as `sampled` [N, 1+steps], column 0 = <start>, to replay a recorded run).
    Returns (ids int64 [N,1+steps], soft f32 [N,1+steps,V]).  Quirks kept: 'none_softmax' never updates the hard token
    (ids stay <start>, nothing ever finishes); the soft row of the step that emits <eos> is already zeroed; a zeroed soft
    row embeds to the zero vector, not to a token's embedding."""
    N = z.shape[0]
    V = P["decoder.fc.1.weight"].shape[0]
    zc = np.concatenate([z, c], 1).astype(F32)
    h = init_state(P, zc)       # [N,H]; [L,N,H] for the multi-layer extension
    cst = np.zeros_like(h)
    tok = np.full(N, START, np.int64)
    finished = np.zeros(N, bool)
    onehot = np.zeros((N, V), F32)
    onehot[:, START] = 1
    cols, softs = [tok.copy()], [onehot]
    soft_in = None
    emb_w, w_ih, b_ih = P["word_emb.weight"], P["decoder.rnn.weight_ih_l0"], P["decoder.rnn.bias_ih_l0"]
    for i in range(max_len):
        e = emb_w[tok] if soft_in is None else (soft_in @ emb_w).astype(F32)
        x = np.concatenate([e, zc], 1).astype(F32)
        gi = (x @ w_ih.T + b_ih).astype(F32)
        multi = h.ndim == 3
        h_l0, c_l0 = (h[0], cst[0]) if multi else (h, cst)
        if cell == "lstm":   # the LSTM extension's cell (torch.nn.LSTM semantics, c0 = 0); not a reference component
            from .lstm import lstm_cell_fwd
            h_l0, c_l0, _ = lstm_cell_fwd(gi, h_l0, c_l0, P["decoder.rnn.weight_hh_l0"], P["decoder.rnn.bias_hh_l0"])
        else:
            h_l0, _ = gru_cell_fwd(gi, h_l0, P["decoder.rnn.weight_hh_l0"], P["decoder.rnn.bias_hh_l0"])
        if multi:
            if cell == "lstm":
                cst = cst.copy()
                cst[0] = c_l0
            top, h, cst_n = _upper_layers(P, h_l0, h, cst if cell == "lstm" else None)
            cst = cst_n if cell == "lstm" else cst
        else:
            top, h, cst = h_l0, h_l0, c_l0
        logits = (top @ P["decoder.fc.1.weight"].T + P["decoder.fc.1.bias"]).astype(F32)
        sm = _log_softmax((logits / F32(temp)).astype(F32))
        soft = np.exp(sm).astype(F32)
        if mode == "greedy_softmax":
            tok = logits.argmax(1).astype(np.int64)
        elif mode == "categorical_softmax":
            tok = sampled[:, i + 1].astype(np.int64).copy()   # recorded draw (already masked: idempotent below)
        elif mode != "none_softmax":
            raise ValueError(mode)
        tok = tok.copy()
        tok[finished] = PAD
        finished = finished | (tok == EOS)
        soft = soft.copy()
        soft[finished] = 0
        cols.append(tok.copy())
        softs.append(soft)
        soft_in = soft
        if finished.all() and len(cols) >= min_length:
            break
    return np.stack(cols, 1), np.stack(softs, 1)
