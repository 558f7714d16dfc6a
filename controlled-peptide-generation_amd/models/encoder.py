"""GRU encoder: (bi)GRU over the embedded sequence, final hidden state -> (mu, logvar).

Counterpart of the reference's models/encoder.py:13-52.  `self.rnn` is a torch.nn.GRU used ONLY as the parameter
container (same state-dict keys, same default initialisation / RNG consumption as the reference); the recurrence
itself runs in the HIP kernels behind cpg.ops.GruSeqFn.  Like the reference (no pack_padded_sequence) the
encoder steps through ALL T positions including pads.
"""
import torch
import torch.nn as nn

from cpg import ops


def build_encoder(enc_type, **E_args):
    if enc_type != 'gru':
        raise ValueError('Please use GRU Encoder')
    return GRUEncoder(**E_args)


class GRUEncoder(nn.Module):
    def __init__(self, emb_dim, h_dim, z_dim, biGRU, layers, p_dropout, cell='gru'):
        super().__init__()
        # cell='lstm' is an extension with torch.nn.LSTM semantics (the reference is GRU-only)
        assert cell in ('gru', 'lstm')
        self.cell = cell
        rnn_cls = nn.GRU if cell == 'gru' else nn.LSTM
        self.rnn = rnn_cls(input_size=emb_dim, hidden_size=h_dim, num_layers=layers, dropout=p_dropout,
                           bidirectional=biGRU, batch_first=True)
        self.biGRU = biGRU
        self.biGRU_factor = 2 if biGRU else 1
        self.h_dim, self.layers, self.p_dropout = h_dim, layers, p_dropout
        self.q_mu = nn.Linear(self.biGRU_factor * h_dim, z_dim)
        self.q_logvar = nn.Linear(self.biGRU_factor * h_dim, z_dim)
        self.rng = None   # DeviceRng set by RNN_VAE.use_device_rng: the inter-layer dropout masks then come from the device streams

    def _layer_keep(self, l, T, B, enc_keep, dev):
        """Keep masks (uint8, time-major [T*B, h_dim] per direction) of the dropout IN FRONT OF layer l >= 1, or None.
        nn.GRU(dropout=p_dropout) (models/encoder.py:25-30) drops the concatenated output of every layer but the last while the
        module is in train mode.  enc_keep: injected masks - [B,T,dirs*h_dim] for a 2-layer encoder, [layers-1,B,T,dirs*h_dim]
        in general (parity tests replay the reference's draw); otherwise drawn here (device stream, or torch's generator)."""
        if l == 0 or self.p_dropout <= 0 or not (self.training or enc_keep is not None):
            return None
        D = self.biGRU_factor * self.h_dim
        if enc_keep is not None:
            k = enc_keep if enc_keep.dim() == 3 else enc_keep[l - 1]
            assert tuple(k.shape) == (B, T, D), (tuple(k.shape), (B, T, D))
            k = ops.transpose01_u8(k.to(torch.uint8))                         # [B,T,D] -> [T,B,D]
        elif self.rng is not None:
            k = self.rng.bernoulli((T, B, D), 1.0 - self.p_dropout, dev)
        else:
            k = (torch.rand(T, B, D, device=dev) >= self.p_dropout).to(torch.uint8)
        k = k.view(T * B, D)
        return [k[:, d * self.h_dim:(d + 1) * self.h_dim].contiguous() for d in range(self.biGRU_factor)]

    def _dirs(self):
        return [("", False), ("_reverse", True)] if self.biGRU else [("", False)]

    def _w(self, name, layer, sfx):
        return getattr(self.rnn, f"{name}_l{layer}{sfx}")

    def _run(self, T, tok=None, emb_weight=None, dense_x=None, enc_keep=None):
        """Either (tok int32 [T,B], emb_weight) - token-table path - or dense_x [T,B,E] embeddings."""
        slabs = finals = None
        dev = self.q_mu.weight.device
        for l in range(self.layers):
            pre = []
            keeps = self._layer_keep(l, T, (tok.shape[1] if tok is not None else dense_x.shape[1]), enc_keep, dev)
            scale = 1.0 / (1.0 - self.p_dropout) if keeps is not None else 1.0
            ximg = None   # f16-pair image of the layer's input rows [xf | xb] (ops.Linear2PlanesFn): built once, shared by both directions
            gates = 3 if self.cell == 'gru' else 4
            if l > 0 and self.biGRU and keeps is None:
                Bq = slabs[0].shape[1]
                if ops.planes_ok(T * Bq, 2 * self.h_dim, gates * self.h_dim):
                    with torch.no_grad():
                        ximg = ops.pair_rows(slabs[0][1:].reshape(T * Bq, -1), slabs[1][:T].reshape(T * Bq, -1))
            tabs = None
            if l == 0 and tok is not None:
                # [V,3H] per direction: W_ih emb[v] + b_ih for every token - both directions in one launch (ops.TokenTablesFn)
                wb = [t for sfx, _ in self._dirs() for t in (self._w("weight_ih", l, sfx), self._w("bias_ih", l, sfx))]
                leaf, pad = ops.emb_leaf(emb_weight)
                tabs = ops.TokenTablesFn.apply(emb_weight, (0, emb_weight.shape[1], leaf, pad), *wb)
            for d_, (sfx, rev) in enumerate(self._dirs()):
                w_ih, b_ih = self._w("weight_ih", l, sfx), self._w("bias_ih", l, sfx)
                tab = dense = None
                if tabs is not None:
                    tab = tabs[d_]
                elif l == 0:
                    B = dense_x.shape[1]
                    dense = ops.LinearFn.apply(dense_x.reshape(T * B, -1), w_ih, b_ih).view(T, B, -1)
                else:
                    B = slabs[0].shape[1]
                    xf = slabs[0][1:].reshape(T * B, -1)            # forward direction: h_t at slot t+1
                    xb = slabs[1][:T].reshape(T * B, -1) if self.biGRU else None   # reverse direction: h_t at slot t
                    if keeps is not None:                           # inter-layer dropout fused into the projection's operand load
                        dense = ops.MaskedLinear2Fn.apply(xf, keeps[0], xb, keeps[1] if self.biGRU else None, scale, w_ih,
                                                          b_ih).view(T, B, -1)
                    elif self.biGRU and ximg is not None:
                        dense = ops.Linear2PlanesFn.apply(xf, xb, ximg, w_ih, b_ih, gates).view(T, B, -1)
                    elif self.biGRU:
                        dense = ops.Linear2Fn.apply(xf, xb, w_ih, b_ih).view(T, B, -1)
                    else:
                        dense = ops.LinearFn.apply(xf, w_ih, b_ih).view(T, B, -1)
                pre.append((tab, dense))
            new = []
            if self.biGRU and ops.OVERLAP:
                # both directions share one launch per time step (GruBiSeqFn; LstmBiSeqFn for the LSTM extension)
                (tab_f, dense_f), (tab_r, dense_r) = pre
                top = l == self.layers - 1   # the top layer is read for its two final states only
                BiFn = ops.GruBiSeqFn if self.cell == 'gru' else ops.LstmBiSeqFn
                new = list(BiFn.apply(tok if l == 0 else None, tab_f, tab_r, dense_f, dense_r,
                                                self._w("weight_hh", l, ""), self._w("bias_hh", l, ""),
                                                self._w("weight_hh", l, "_reverse"), self._w("bias_hh", l, "_reverse"), T, top))
                if top:
                    finals = new
                slabs = new
                continue
            for d, (sfx, rev) in enumerate(self._dirs()):
                tab, dense = pre[d]
                w_hh, b_hh = self._w("weight_hh", l, sfx), self._w("bias_hh", l, sfx)
                if self.cell == 'gru':
                    new.append(ops.GruSeqFn.apply(tok if l == 0 else None, tab, None, dense, None, w_hh, b_hh, T, rev))
                else:
                    new.append(ops.LstmSeqFn.apply(tok if l == 0 else None, tab, None, dense, None, None, w_hh, b_hh, T, rev))
            slabs = new
        # final states of the top layer: forward slab slot T, reverse slab slot 0 (reference: cat(h[-2], h[-1]))
        if finals is None:
            finals = [slabs[0][T]] + ([slabs[1][0]] if self.biGRU else [])
        # gradient-bucket boundary: the two heads' gradients are final behind it.  The heads read the two final states where they are
        # (no torch.cat) and run as one grouped launch forward, three launches backward (ops.EncoderHeadsFn)
        hf, hr = finals[0], (finals[1] if len(finals) > 1 else None)
        if hr is not None:
            hf, hr = ops.grad_boundary('encoder_heads', hf, hr)
        else:
            hf = ops.grad_boundary('encoder_heads', hf)
        return ops.EncoderHeadsFn.apply(hf, hr, self.q_mu.weight, self.q_mu.bias, self.q_logvar.weight, self.q_logvar.bias)

    def forward_tokens(self, ids, emb_weight, enc_keep=None):
        """ids int64 [B,T]: fused path, W_ih emb[tok] is a V-row lookup table (never a [B,T,E] GEMM).
        enc_keep: inter-layer dropout masks to inject (_layer_keep)."""
        tok = ops.tokens_prepare(ids)
        return self._run(ids.shape[1], tok=tok, emb_weight=emb_weight, enc_keep=enc_keep)

    def forward(self, x, enc_keep=None):
        """x: embeddings [mbsize, seq_len, emb_dim] (reference signature; used for soft inputs)."""
        T = x.shape[1]
        return self._run(T, dense_x=x.transpose(0, 1).contiguous(), enc_keep=enc_keep)
