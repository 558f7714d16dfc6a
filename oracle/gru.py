"""GRU sequence forward / backward in numpy float32 (oracle; test infrastructure only).

Restates torch.nn.GRU as the reference uses it (models/encoder.py:25-30,42;
models/decoder.py:40-41,77,98).  Gate row order in weight_ih / weight_hh is r, z, n:
    r  = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
    z  = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
    n  = tanh  (W_in x + b_in + r * (W_hn h + b_hn))
    h' = (1 - z) * n + z * h
"""
import numpy as np

F32 = np.float32


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(F32)


def gru_cell_fwd(gi, h, w_hh, b_hh):
    """gi = W_ih x + b_ih already formed, [B,3H]; returns h', cache."""
    H = h.shape[1]
    gh = h @ w_hh.T + b_hh
    r = sigmoid(gi[:, :H] + gh[:, :H])
    z = sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    hn = gh[:, 2 * H:]
    n = np.tanh(gi[:, 2 * H:] + r * hn).astype(F32)
    h_new = ((1.0 - z) * n + z * h).astype(F32)
    return h_new, (r, z, n, hn, h)


def gru_cell_bwd(dh_new, cache, w_hh):
    """Returns dgi [B,3H], dgh [B,3H], dh_prev [B,H]."""
    r, z, n, hn, h = cache
    dn = dh_new * (1.0 - z)
    dz = dh_new * (h - n)
    dh_prev = dh_new * z
    dn_pre = dn * (1.0 - n * n)
    dz_pre = dz * z * (1.0 - z)
    dr = dn_pre * hn
    dhn = dn_pre * r
    dr_pre = dr * r * (1.0 - r)
    dgi = np.concatenate([dr_pre, dz_pre, dn_pre], 1).astype(F32)
    dgh = np.concatenate([dr_pre, dz_pre, dhn], 1).astype(F32)
    dh_prev = (dh_prev + dgh @ w_hh).astype(F32)
    return dgi, dgh, dh_prev


def gru_seq_fwd(gi_seq, h0, w_hh, b_hh, reverse=False):
    """gi_seq [B,T,3H] (input-side pre-activations), h0 [B,H].
    Returns hs [B,T,H] indexed by time position (not by processing order), h_last, caches."""
    B, T, _ = gi_seq.shape
    H = h0.shape[1]
    hs = np.zeros((B, T, H), F32)
    caches = [None] * T
    h = h0.astype(F32)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h, caches[t] = gru_cell_fwd(gi_seq[:, t], h, w_hh, b_hh)
        hs[:, t] = h
    return hs, h, caches


def gru_seq_bwd(dhs, dh_last, caches, w_hh, reverse=False):
    """dhs [B,T,H]: gradient arriving at each step's output; dh_last: extra gradient on the final state.
    Returns dgi_seq [B,T,3H], dh0, dW_hh, db_hh."""
    B, T, H = dhs.shape
    dgi_seq = np.zeros((B, T, 3 * H), F32)
    dW_hh = np.zeros_like(w_hh)
    db_hh = np.zeros(3 * H, F32)
    dh = dh_last.astype(F32).copy() if dh_last is not None else np.zeros((B, H), F32)
    order = range(T) if reverse else range(T - 1, -1, -1)  # reverse of processing order
    for t in order:
        dh = dh + dhs[:, t]
        dgi, dgh, dh = gru_cell_bwd(dh, caches[t], w_hh)
        dgi_seq[:, t] = dgi
        dW_hh += dgh.T @ caches[t][4]
        db_hh += dgh.sum(0)
    return dgi_seq, dh, dW_hh.astype(F32), db_hh.astype(F32)
