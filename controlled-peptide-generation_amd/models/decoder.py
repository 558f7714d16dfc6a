"""GRU decoder with [emb; z; c] input at every step and h0 = [z; c].

Counterpart of the reference's GRUDecoder / WordDropout (models/decoder.py:23-133).  `self.rnn`, `self.fc` are torch
modules used only as parameter containers (state-dict keys `decoder.rnn.*`, `decoder.fc.1.*`).  The input projection
is split exactly: W_ih [emb(tok); z; c] + b_ih = tab[tok] + rowc with tab = emb @ W_ih[:, :E]^T + b_ih (V rows) and
rowc = [z;c] @ W_ih[:, E:]^T (constant over time), so the [B,T,E+Hd] concatenation is never built.
The DeconvDecoder alternative (G_class='deconv', decoder.py:136-323) is not on the hot path and not provided.
"""
import numpy as np
import torch
import torch.nn as nn

from cpg import ops
from models.mutils import UNK_IDX


def build_decoder(G_class, GRU_args, deconv_args, **common_args):
    if G_class == 'gru':
        kw = dict(GRU_args)
        kw.update(common_args)
        return GRUDecoder(**kw)
    if G_class == 'deconv':
        raise NotImplementedError("G_class='deconv' is outside the MI355X hot path (SURVEY section 2: out of scope)")
    raise ValueError('Please use one of the following for dec_type: gru | deconv.')


class WordDropout(nn.Module):
    """Replaces tokens by <unk> with probability p - in train AND eval mode, like the reference (no self.training
    check, decoder.py:117-133).  `sample_mask` draws the mask; callers may inject one instead (parity tests)."""

    def __init__(self, p_word_dropout):
        super().__init__()
        self.p = p_word_dropout
        self.rng = None  # (seed, counter) device stream when set by the model; numpy global stream otherwise

    def sample_mask(self, x):
        if self.rng is not None:
            return self.rng.bernoulli(tuple(x.shape), self.p, x.device)
        m = np.random.binomial(1, p=self.p, size=tuple(x.size())).astype('uint8')  # reference's generator & call order
        return torch.from_numpy(m).to(x.device)

    def forward(self, x, mask=None):
        mask = self.sample_mask(x) if mask is None else mask
        out = x.clone()
        out[mask.bool()] = UNK_IDX
        return out


class GRUDecoder(nn.Module):
    def __init__(self, embedding, emb_dim, output_dim, h_dim, p_word_dropout, p_out_dropout, skip_connetions, cell='gru',
                 layers=1):
        super().__init__()
        self.emb = embedding
        # cell='lstm': extension with torch.nn.LSTM semantics, h0 = [z;c], c0 = 0 (the reference is GRU-only)
        assert cell in ('gru', 'lstm')
        self.cell = cell
        # layers > 1: EXTENSION (BASELINE.json configs[4] names a "2-layer dec"; the reference hard-wires one layer,
        # models/decoder.py:40-41, models/model.py:283-284 - parity unpinned against the reference).  Semantics =
        # torch.nn.GRU / nn.LSTM(num_layers=layers, dropout=0): layer l >= 1 reads layer l-1's step outputs, EVERY layer starts
        # from h0 = [z;c] (the reference's `init_h.unsqueeze(0)` repeated per layer; c0 = 0 for the LSTM), the vocabulary projection
        # reads the top layer.  State-dict keys decoder.rnn.*_l{l}.
        assert layers >= 1
        self.layers = int(layers)
        self.rnn = (nn.GRU if cell == 'gru' else nn.LSTM)(emb_dim, h_dim, num_layers=self.layers, batch_first=True)
        self.fc = nn.Sequential(nn.Dropout(p_out_dropout), nn.Linear(h_dim, output_dim))
        self.word_dropout = WordDropout(p_word_dropout)
        self.p_out = p_out_dropout
        self.h_dim = h_dim
        self.skip_connetions = skip_connetions
        if skip_connetions:   # models/decoder.py:48-51: two bias-free [h_dim,h_dim] maps (same construction order = same init stream)
            self.skip_weight_x = nn.Linear(h_dim, h_dim, bias=False)
            self.skip_weight_z = nn.Linear(h_dim, h_dim, bias=False)
        self.rng = None
        # Length-sorted ("ragged") teacher forcing: rows are visited longest first and a row leaves the recurrence once all
        # its remaining targets are <pad> (those positions carry no loss and no gradient, losses.py:27).  The logits of
        # such positions are then fc(0) instead of the reference's values, so this is OFF for the model API and switched
        # on by the trainer (train_vae.train_step), which only consumes the loss.
        self.ragged = False
        # ids [B,T] whose reconstruction loss the caller will ask for (losses.recon_dec on this forward's logits), or None: set by
        # train_vae.train_step around its forward pass - the decoder then ends in ops.VocabReconFn
        self.recon_targets = None

    def init_hidden(self, z, c):
        return torch.cat([z, c], dim=1)

    def _emb_w(self):
        if self.emb.padding_idx is None:
            return self.emb.weight
        return ops.tag_emb(ops.ZeroRowGradFn.apply(self.emb.weight, self.emb.padding_idx), self.emb.weight, self.emb.padding_idx)

    def _tables(self, zc, emb_w=None):
        E = self.emb.weight.shape[1]
        w_ih = self.rnn.weight_ih_l0
        if emb_w is None:
            emb_w = self._emb_w()
        leaf, pad = ops.emb_leaf(emb_w)
        tab, = ops.TokenTablesFn.apply(emb_w, (0, E, leaf, pad), w_ih, self.rnn.bias_ih_l0)   # [V,3H]
        rowc = ops.LinearColsFn.apply(zc, w_ih, None, E, w_ih.shape[1])            # [B,3H]
        return tab, rowc

    def forward(self, x, z, c, wd_mask=None, out_keep=None, emb_w=None, zc=None):
        """Teacher forcing.  x ids [B,T]; returns logits [B,T,V].
        wd_mask uint8 [B,T] / out_keep uint8 [B,T,H] inject the two dropout masks (otherwise sampled here).
        emb_w: the pad-masked embedding matrix shared with the step's other consumers (RNN_VAE.forward), or None.
        zc: [z;c] when the caller has it already (ops.LatentFn writes it beside z), else formed here."""
        B, T = x.shape
        if zc is None:
            zc = self.init_hidden(z, c)
        if wd_mask is None:
            wd_mask = self.word_dropout.sample_mask(x)
        tok = ops.tokens_prepare(x, wd_mask)
        # gradient-bucket boundary: once the gradients of [z;c] and of the embedding rows the decoder reads are complete, every
        # gradient of the decoder's own parameters has been enqueued (cpg.optim starts their all-reduce there)
        emb_w = self._emb_w() if emb_w is None else emb_w
        leaf = ops.emb_leaf(emb_w)
        zc, emb_w = ops.grad_boundary('decoder', zc, emb_w)
        emb_w = ops.tag_emb(emb_w, *leaf)
        tab, rowc = self._tables(zc, emb_w)
        ragged = self.ragged and self.cell == 'gru' and self.layers == 1 and torch.is_grad_enabled()
        perm = inv = step_rows = None
        if ragged:
            # step t consumes x[:, t] and is scored against x[:, t+1]: live while some target at or after t is not <pad>
            # (pads only ever trail).  All on the device: no host sync.
            last = (x != self.emb.padding_idx).sum(1) - 2                       # index of the last scored step
            perm = torch.argsort(last, descending=True)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(B, device=x.device)
            steps = torch.arange(T, device=x.device)
            step_rows = (last[None, :] >= steps[:, None]).sum(1).to(torch.int32)  # [T] live rows per step (a prefix)
            tok = tok.index_select(1, perm).contiguous()
            rowc = rowc.index_select(0, perm)
            zc = zc.index_select(0, perm)
        if self.cell == 'gru':
            outs = ops.GruSeqFn.apply(tok, tab, rowc, None, zc, self.rnn.weight_hh_l0, self.rnn.bias_hh_l0, T, False, True,
                                      step_rows, True)                    # step outputs [T,B,H] (slots 1..T of the slab)
        else:
            outs = ops.LstmSeqFn.apply(tok, tab, rowc, None, zc, None, self.rnn.weight_hh_l0, self.rnn.bias_hh_l0, T, False)[1:]
        for l in range(1, self.layers):
            # upper layer l: dense input term W_ih_l h^{l-1}_t + b_ih_l over all T*B rows, then the same recurrence from h0 = [z;c]
            w_ih, b_ih = getattr(self.rnn, f"weight_ih_l{l}"), getattr(self.rnn, f"bias_ih_l{l}")
            w_hh, b_hh = getattr(self.rnn, f"weight_hh_l{l}"), getattr(self.rnn, f"bias_hh_l{l}")
            xin = outs.reshape(T * B, self.h_dim)
            gates = 3 if self.cell == 'gru' else 4
            if ops.planes_ok(T * B, self.h_dim, gates * self.h_dim):   # many rows: conversion-free products on f16-pair images (csrc/planes.hip)
                with torch.no_grad():
                    ximg = ops.pair_rows(xin)
                dense = ops.Linear2PlanesFn.apply(xin, None, ximg, w_ih, b_ih, gates).view(T, B, -1)
            else:
                dense = ops.LinearFn.apply(xin, w_ih, b_ih).view(T, B, -1)
            if self.cell == 'gru':
                outs = ops.GruSeqFn.apply(None, None, None, dense, zc, w_hh, b_hh, T, False, True, None, True)
            else:
                outs = ops.LstmSeqFn.apply(None, None, None, dense, zc, None, w_hh, b_hh, T, False)[1:]
        hs = outs.reshape(T * B, self.h_dim)
        if self.skip_connetions:
            # rnn_out := skip_weight_x(rnn_out) + skip_weight_z([z;c]) (models/decoder.py:80-81); the second term is constant over time
            sz = ops.LinearFn.apply(zc, self.skip_weight_z.weight, None)
            hs = ops.SkipAddFn.apply(hs, self.skip_weight_x.weight, sz, T)
        keep, scale = None, 1.0
        if out_keep is not None:
            keep = ops.transpose01_u8(out_keep.to(torch.uint8))          # [B,T,H] -> [T,B,H]
            if ragged:
                keep = keep.index_select(1, perm).contiguous()
            scale = 1.0 / (1.0 - self.p_out) if self.p_out > 0 else 1.0
        elif self.training and self.p_out > 0:
            keep = self._sample_keep((T, B, self.h_dim), x.device)
            scale = 1.0 / (1.0 - self.p_out)
        fc = self.fc[1]
        if self.recon_targets is not None and not ragged:
            # the trainer's form (train_vae.train_step sets recon_targets = the batch): projection + cross-entropy as one node over the
            # time-major rows; the returned logits are a [B,T,V] VIEW of them that carries the loss (losses.recon_dec picks it up)
            import losses as _losses
            loss, ltm = ops.VocabReconFn.apply(hs, keep, scale, fc.weight, fc.bias, self.recon_targets,
                                               _losses.global_target_count(self.recon_targets))
            logits = ltm.view(T, B, -1).transpose(0, 1)
            logits._cpg_recon = (self.recon_targets, loss)
            return logits
        logits_tm = ops.VocabFcFn.apply(hs, keep, scale, fc.weight, fc.bias).view(T, B, -1)
        if ragged:
            logits_tm = logits_tm.index_select(1, inv)                   # back to the caller's row order
        return ops.Transpose01Fn.apply(logits_tm)

    def _sample_keep(self, shape, device):
        if self.rng is not None:
            return self.rng.bernoulli(shape, 1.0 - self.p_out, device)
        return (torch.rand(shape, device=device) >= self.p_out).to(torch.uint8)

    def forward_sample(self, sampleSoft, sampleHard, z, c, h):
        """One decode step (reference signature, decoder.py:86-109): h is [layers,N,H] ([1,N,H] in the reference); returns logits
        [N,V], h [layers,N,H].  sampleSoft [N,V] (a softmax row, or zeros) takes the soft-embedding branch; inference only (no
        autograd tape).  LSTM extension: the state is torch.nn.LSTM's pair - h = (h, c) in, the same pair out."""
        from cpg.decode import UpperLayers
        zc = self.init_hidden(z, c)
        lstm = self.cell == 'lstm'
        if lstm:
            assert isinstance(h, (tuple, list)) and len(h) == 2, "LSTM decoder: pass the state as (h, c), as torch.nn.LSTM takes it"
            h, cst = h
            assert cst.shape[0] == self.layers
        assert h.shape[0] == self.layers, "state must be [layers, N, H]"
        with torch.no_grad():
            tab, rowc = self._tables(zc)
            h_prev = h[0].contiguous()
            N, H = h_prev.shape
            hs = torch.stack([h_prev, torch.empty_like(h_prev)])
            cs = torch.stack([cst[0].contiguous(), torch.empty_like(h_prev)]) if lstm else None
            rnn = self.rnn
            if sampleSoft is not None:
                # soft_embed (mutils.py:39-45): W_ih[:, :E] (soft @ emb) = soft @ (emb W_e^T), the step's dense input term
                E = self.emb.weight.shape[1]
                w_soft = ops.LinearFn.apply(rnn.weight_ih_l0[:, :E].contiguous(), self.emb.weight, None)
                dense = ops.LinearFn.apply(sampleSoft.float().contiguous(), w_soft.contiguous(), rnn.bias_ih_l0)
                if lstm:
                    ops.call("cpg_lstm_seq_fwd", 1, N, H, 0, ops._p(rnn.weight_hh_l0), ops._p(rnn.bias_hh_l0), None, None,
                             ops._p(rowc.contiguous()), ops._p(dense.contiguous()), ops._p(hs), ops._p(cs), None, ops._stream())
                else:
                    ops.call("cpg_gru_seq_fwd", 1, N, H, 0, ops._p(rnn.weight_hh_l0), ops._p(rnn.bias_hh_l0), None, None,
                             ops._p(rowc.contiguous()), ops._p(dense.contiguous()), ops._p(hs), None, 0, N, None,
                             ops._p(ops.weight_exp(rnn.weight_hh_l0)), ops._stream())
            else:
                tok = sampleHard.to(torch.int32).contiguous()
                if lstm:
                    ops.lstm_step(tok, tab, rowc, hs[0], cs[0], hs[1], cs[1], rnn.weight_hh_l0, rnn.bias_hh_l0)
                else:
                    ops.gru_step(tok, tab, rowc, hs[0], hs[1], rnn.weight_hh_l0, rnn.bias_hh_l0)
            top, h_new, c_new = hs[1], hs[1].unsqueeze(0), (cs[1].unsqueeze(0) if lstm else None)
            if self.layers > 1:
                upper = UpperLayers(self, h_prev, lstm)
                for j in range(self.layers - 1):
                    upper.hs[j][0].copy_(h[j + 1])
                    if lstm:
                        upper.cs[j][0].copy_(cst[j + 1])
                top = upper.step(hs[1])
                uh, uc = upper.states()
                h_new = torch.cat([h_new, uh], 0)
                c_new = torch.cat([c_new, uc], 0) if lstm else None
            logits = self.project(top, self.skip_term(zc))
        return logits, ((h_new, c_new) if lstm else h_new)

    def skip_term(self, zc):
        """skip_weight_z([z;c]) [N,H] of a decode (constant over its steps), or None without skip connections.  Inference only."""
        if not self.skip_connetions:
            return None
        return ops.linear_raw(zc.contiguous(), self.skip_weight_z.weight, None)

    def step_keep(self, rows, device):
        """Out-dropout keep mask [rows,H] of ONE decode step when the module is in train mode (the reference's nn.Dropout(p_out) in
        front of the vocabulary projection is live in forward_sample whenever the model has not been put in eval mode:
        generate_sentences(eval_mode=False), models/model.py:216-221), else None."""
        if not (self.training and self.p_out > 0):
            return None
        return self._sample_keep((rows, self.h_dim), device)

    def project(self, h, sz=None, keep='sample', logits=None):
        """The tail of GRUDecoder.forward_sample (models/decoder.py:101-108) for step outputs h [N,H]: skip connections when the
        model has them (output := skip_weight_x(output) + skip_weight_z([z;c]); sz = skip_term(zc)), out-dropout when in train mode
        (keep: uint8 [N,H] to inject, None for none, 'sample' = step_keep), vocabulary projection -> logits [N,V].  No tape."""
        out = h
        if self.skip_connetions:
            out = ops.linear_raw(h, self.skip_weight_x.weight, None, out=sz.clone(), accumulate=True)
        if isinstance(keep, str):
            keep = self.step_keep(h.shape[0], h.device)
        scale = 1.0 / (1.0 - self.p_out) if (keep is not None and self.p_out > 0) else 1.0
        fc = self.fc[1]
        N, H = out.shape
        if logits is None:
            logits = torch.empty(N, fc.weight.shape[0], device=h.device, dtype=torch.float32)
        ops.call("cpg_vocab_fc_fwd", ops._p(out), ops._p(keep), float(scale), ops._p(fc.weight), ops._p(fc.bias), ops._p(logits), N, H,
                 logits.shape[1], ops._stream())
        return logits
