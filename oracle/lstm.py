"""LSTM sequence forward / backward in numpy float32 (oracle; test infrastructure only).

**Parity unpinned against the reference**: IBM/controlled-peptide-generation contains no LSTM (every RNN is
nn.GRU).  BASELINE.json's configs name an LSTM cell, so the build offers one; this restatement is pinned to
torch.nn.LSTM only (tests/test_lstm.py).  Gate row order i, f, g, o:
    i = sigmoid(.), f = sigmoid(.), g = tanh(.), o = sigmoid(.)  of  W_i* x + b_i* + W_h* h + b_h*
    c' = f*c + i*g ;  h' = o*tanh(c')
"""
import numpy as np

from .gru import sigmoid, F32


def lstm_cell_fwd(gi, h, c, w_hh, b_hh):
    H = h.shape[1]
    a = gi + h @ w_hh.T + b_hh
    i, f = sigmoid(a[:, :H]), sigmoid(a[:, H:2 * H])
    g, o = np.tanh(a[:, 2 * H:3 * H]).astype(F32), sigmoid(a[:, 3 * H:])
    c_new = (f * c + i * g).astype(F32)
    tc = np.tanh(c_new).astype(F32)
    h_new = (o * tc).astype(F32)
    return h_new, c_new, (i, f, g, o, c, tc, h)


def lstm_cell_bwd(dh, dc_carry, cache, w_hh):
    i, f, g, o, c_prev, tc, h_prev = cache
    do = dh * tc
    dc = dc_carry + dh * o * (1.0 - tc * tc)
    dgates = np.concatenate([dc * g * i * (1 - i), dc * c_prev * f * (1 - f), dc * i * (1 - g * g), do * o * (1 - o)], 1).astype(F32)
    return dgates, (dgates @ w_hh).astype(F32), (dc * f).astype(F32)


def lstm_seq_fwd(gi_seq, h0, c0, w_hh, b_hh, reverse=False):
    B, T, _ = gi_seq.shape
    H = h0.shape[1]
    hs = np.zeros((B, T, H), F32)
    caches = [None] * T
    h, c = h0.astype(F32), c0.astype(F32)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        h, c, caches[t] = lstm_cell_fwd(gi_seq[:, t], h, c, w_hh, b_hh)
        hs[:, t] = h
    return hs, h, c, caches


def lstm_seq_bwd(dhs, caches, w_hh, reverse=False):
    """dhs [B,T,H] gradient on every step output.  Returns dgates [B,T,4H], dh0, dc0, dW_hh, db_hh."""
    B, T, H = dhs.shape
    dG = np.zeros((B, T, 4 * H), F32)
    dW = np.zeros_like(w_hh)
    db = np.zeros(4 * H, F32)
    dh = np.zeros((B, H), F32)
    dc = np.zeros((B, H), F32)
    for t in (range(T) if reverse else range(T - 1, -1, -1)):
        dg, dh_prev, dc = lstm_cell_bwd(dh + dhs[:, t], dc, caches[t], w_hh)
        dG[:, t] = dg
        dW += dg.T @ caches[t][6]
        db += dg.sum(0)
        dh = dh_prev
    return dG, dh, dc, dW.astype(F32), db.astype(F32)
