"""RNN_VAE on the MI355X hot path.

Mirrors the public surface of the reference's models/model.py (ctor signature :23-35, forward :146-195,
forward_encoder :96-105, sample_z / sample_z_prior / sample_c_prior :107-126, generate_sentences :197-223,
sample_G :225-385, the parameter groups :75-94) so main.py / train_vae.py / sample_pipeline.py / api.py style callers
work unchanged, with the same state-dict keys and the same default initialisation.  All arithmetic runs in the HIP
kernels of libcpg_hip.so through cpg.ops / cpg.decode; torch modules below are parameter containers.

Differences that are deliberate and documented:
  * `forward` accepts optional keyword `rnd=dict(eps=, c=, wd_mask=, out_mask=, enc_keep=)` to inject the step's random draws
    (the reference mixes torch and numpy generators; parity tests inject its captured draws);
  * `.device` defaults to cuda (as in the reference, model.py:41) and stays an assignable attribute (api.py:96);
  * flows (flow>0) are not on this path and raise; the soft sampling modes run forward only (no autograd tape:
    the reference never trains through them).
"""
from itertools import chain

import numpy as np
import torch
import torch.nn as nn

from cpg import ops
from cpg import decode as cdecode
from models.decoder import build_decoder
from models.encoder import build_encoder
from models.classifier import build_classifier
from models.mutils import UNK_IDX, PAD_IDX, START_IDX, EOS_IDX

HARD_MODES = ('categorical', 'greedy', 'beam')
SOFT_MODES = ('gumbel_soft', 'gumbel_ST', 'greedy_softmax', 'categorical_softmax', 'none_softmax')


DeviceRng = ops.DeviceRng   # (seed, step-relative offset, device base): see cpg.ops


class RNN_VAE(nn.Module):
    def __init__(self, n_vocab, max_seq_len, z_dim, c_dim, emb_dim, pretrained_emb, freeze_embeddings, flow, flow_type,
                 E_args, G_args, C_args):
        super().__init__()
        self.MAX_SEQ_LEN = max_seq_len
        self.n_vocab = n_vocab
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.emb_dim = emb_dim
        self.device = torch.device('cuda')
        self.UNK_IDX, self.PAD_IDX, self.START_IDX, self.EOS_IDX = UNK_IDX, PAD_IDX, START_IDX, EOS_IDX

        self.word_emb = nn.Embedding(n_vocab, emb_dim, PAD_IDX)
        if pretrained_emb is not None:
            assert emb_dim == pretrained_emb.size(1), 'emb dim dont match with pretrained'
            self.word_emb = nn.Embedding(n_vocab, emb_dim, PAD_IDX)
            self.word_emb.weight.data.copy_(pretrained_emb)
        if freeze_embeddings:
            self.word_emb.weight.requires_grad = False

        self.encoder = build_encoder('gru', emb_dim=emb_dim, z_dim=z_dim, **E_args)
        self.decoder = build_decoder(embedding=self.word_emb, emb_dim=emb_dim + z_dim + c_dim, output_dim=n_vocab,
                                     h_dim=z_dim + c_dim, **G_args)
        self.classifier = build_classifier('cnn', emb_dim, **C_args)
        self.use_flow = flow > 0
        if self.use_flow:
            raise NotImplementedError('normalizing flows are dead code in the reference training path '
                                      '(models/model.py:173-177 raises) and are not provided')
        self.rng = None  # DeviceRng -> on-device Philox draws; None -> the reference's torch/numpy generators
        # train_vae.train_step switches this on around its forward pass: fused latent / reconstruction nodes whose outputs are consumed
        # through the losses.* functions only (same values and parameter gradients; intermediate tensors such as z no longer see every
        # gradient path through autograd).  Off: every intermediate behaves exactly as in the reference's graph.
        self.fused_train = False

    # ------------------------------------------------------------------ randomness
    def use_device_rng(self, seed):
        """Draw eps / z_prior / c / dropout masks with the on-device counter-based streams (no host RNG, no H2D)."""
        self.rng = DeviceRng(seed)
        self.decoder.rng = self.rng
        self.decoder.word_dropout.rng = self.rng
        self.encoder.rng = self.rng
        return self

    def _randn(self, n, d):
        if self.rng is not None:
            return self.rng.normal((n, d), self.device)
        return torch.randn(n, d).to(self.device)

    # ------------------------------------------------------------------ parameter groups (model.py:75-94)
    def classifier_params(self):
        return filter(lambda p: p.requires_grad, self.classifier.parameters())

    def decoder_params(self):
        return filter(lambda p: p.requires_grad, self.decoder.parameters())

    def encoder_params(self):
        return filter(lambda p: p.requires_grad, chain(self.word_emb.parameters(), self.encoder.parameters()))

    def vae_params(self):
        # NB: word_emb.weight appears twice (decoder.emb is the same module) exactly like the reference
        return filter(lambda p: p.requires_grad,
                      chain(self.word_emb.parameters(), self.encoder.parameters(), self.decoder.parameters()))

    # ------------------------------------------------------------------ pieces of the forward pass
    def _emb_weight(self):
        """The embedding matrix as the step's graph sees it: row PAD gets no gradient (nn.Embedding(padding_idx), models/model.py:47).
        The leaf behind it rides along (ops.emb_leaf): kernels that produce an embedding gradient inside FusedAdamClip.backward add it
        straight into the parameter's gradient buffer, PAD row skipped."""
        return ops.tag_emb(ops.ZeroRowGradFn.apply(self.word_emb.weight, PAD_IDX), self.word_emb.weight, PAD_IDX)

    def forward_encoder(self, inputs, emb_w=None, enc_keep=None):
        """ids [mbsize, seq_len] -> (mu, logvar);  soft inputs [mbsize, seq_len, n_vocab] go through soft_embed.
        emb_w: the embedding matrix as this step's graph sees it (_emb_weight()), when the caller shares one between the
        encoder, the decoder and the classifier - one pad-row mask and one accumulation per step instead of one per consumer.
        enc_keep: inter-layer dropout masks of a multi-layer encoder to inject (GRUEncoder._layer_keep; parity tests)."""
        if inputs.dim() == 2:
            return self.encoder.forward_tokens(inputs, self._emb_weight() if emb_w is None else emb_w, enc_keep=enc_keep)
        from models.mutils import soft_embed
        return self.encoder(soft_embed(self.word_emb, inputs), enc_keep=enc_keep)

    def sample_z(self, mu, logvar, eps=None):
        if eps is None:
            eps = self._randn(mu.size(0), self.z_dim)
        return ops.ReparamFn.apply(mu, logvar, eps)

    def sample_z_prior(self, mbsize):
        return self._randn(mbsize, self.z_dim)

    def sample_c_prior(self, mbsize):
        """c ~ Cat([.5,.5]) one-hot [mbsize, 2]."""
        if self.rng is not None:
            return self.rng.onehot2(mbsize, 0.5, self.device)   # the draws of bernoulli((mbsize,), 0.5) as one-hot rows, one launch
        return torch.from_numpy(np.random.multinomial(1, [0.5, 0.5], mbsize).astype('float32')).to(self.device)

    def forward_decoder(self, inputs, z, c, wd_mask=None, out_keep=None, emb_w=None, zc=None):
        return self.decoder(inputs, z, c, wd_mask=wd_mask, out_keep=out_keep, emb_w=emb_w, zc=zc)

    def forward_classifier(self, inputs, emb_w=None):
        """Token inputs run the HIP path (token-table convolutions), differentiable like the reference's: with q_c='classifier'
        gradients reach the classifier and word_emb through c (models/model.py:135-144,186-188)."""
        if inputs.dim() == 2:
            return self.classifier.forward_tokens(inputs, self._emb_weight() if emb_w is None else emb_w)
        else:
            from models.mutils import soft_embed
            x = soft_embed(self.word_emb, inputs)
            return self.classifier(x)

    def forward(self, sequences, q_c='prior', sample_z=1, rnd=None):
        """-> ((mu, logvar), (z, c), dec_logits [mbsize, seq_len, n_vocab])"""
        rnd = rnd or {}
        mbsize = sequences.size(0)
        emb_w = self._emb_weight() if sequences.dim() == 2 else None
        mu, logvar = self.forward_encoder(sequences, emb_w, enc_keep=rnd.get('enc_keep'))
        assert mu.size(0) == logvar.size(0) == mbsize
        if sample_z != 'max':
            assert sample_z == 1, 'sample_z > 1 is not supported (reference: TODO)'
        c = None
        if 'c' in rnd:
            c = rnd['c']
        elif isinstance(q_c, torch.Tensor):
            c = torch.zeros(mbsize, 2, device=self.device)
            c.scatter_(1, q_c.unsqueeze(1), 1)
        elif q_c == 'classifier':
            c = torch.softmax(self.forward_classifier(sequences, emb_w), dim=1)
        elif q_c != 'prior':
            raise ValueError("q_c is not labels, prior, or classifier")
        zc = None
        fused = (self.fused_train and sample_z == 1 and (c is None or not c.requires_grad) and mu.is_cuda
                 and (self.rng is not None or (rnd.get('eps') is not None and c is not None)))
        if fused:
            # the trainer's form (self.fused_train, set by train_vae.train_step): ONE node for reparameterisation + class prior + [z;c] +
            # the analytic latent penalties (ops.LatentFn); losses.latent_terms finds the penalties on `mu` (same values as its own
            # pass, one backward launch for all of them).  z then receives only the gradients of its direct consumers (the MMD terms);
            # what arrives through the decoder's [z;c] goes to mu / logvar inside the node (z._cpg_zc carries that tensor).
            z, zc, c, kl, klmu, l1, sums5 = ops.LatentFn.apply(mu, logvar, rnd.get('eps'), c, self.rng)
            mu._cpg_latent = (logvar, kl, klmu, l1, sums5)
            z._cpg_zc = zc
        else:
            z = mu if sample_z == 'max' else self.sample_z(mu, logvar, rnd.get('eps'))
            if c is None:
                c = self.sample_c_prior(mbsize)
        dec_logits = self.forward_decoder(sequences, z, c, wd_mask=rnd.get('wd_mask'), out_keep=rnd.get('out_mask'), emb_w=emb_w, zc=zc)
        return (mu, logvar), (z, c), dec_logits

    # ------------------------------------------------------------------ generation
    def generate_sentences(self, mbsize, z=None, c=None, eval_mode=True, **sample_kwargs):
        if z is None:
            z = self.sample_z_prior(mbsize)
        if c is None:
            c = self.sample_c_prior(mbsize)
        if eval_mode:
            self.eval()
        sentences = self.sample_G(mbsize, z, c, **sample_kwargs)
        if eval_mode:
            self.train()  # the reference ALWAYS returns to train mode here
        return sentences, z, c.argmax(dim=1)

    def sample_G(self, mbsize, z, c, sample_mode='categorical', temp=1.0, gumbel_temp=1.0, prepend_start_idx=True,
                 prevent_empty=False, min_length=1, beam_size=5, n_best=3, uniforms=None, out_keep=None):
        """uniforms (extra, categorical only): f64 [MAX_SEQ_LEN, mbsize] draws to inject (parity tests).
        out_keep (extra): uint8 [steps, rows, h_dim] out-dropout masks of a TRAIN-mode decode to inject (rows = mbsize, or
        beam_size * mbsize beam-major for 'beam').  Called in train mode - generate_sentences(eval_mode=False),
        models/model.py:216-221 - the decoder's nn.Dropout(p_out) is live in every step, exactly as in the reference; the
        masks are drawn per step from the model's streams unless injected."""
        if sample_mode in ('gumbel_soft', 'gumbel_ST'):
            raise NotImplementedError('gumbel_soft / gumbel_ST are placeholders in the reference too (models/model.py:330-336 '
                                      'leaves sampleSoftIx unset and fails)')
        if sample_mode not in HARD_MODES + SOFT_MODES:
            raise Exception('Sample mode {} not implemented.'.format(sample_mode))
        assert beam_size >= n_best, "Can't return more than max hypothesis"
        assert mbsize == z.size(0) == c.size(0), 'oops sizes dont match {} {} {}'.format(mbsize, z.size(0), c.size(0))
        z, c = z.to(self.device).float(), c.to(self.device).float()
        if sample_mode == 'beam':
            return cdecode.decode_beam(self.decoder, z, c, self.MAX_SEQ_LEN, beam_size, n_best, min_length, out_keep=out_keep)
        if sample_mode in SOFT_MODES:
            assert not prevent_empty, 'cant prevent_empty when soft sampling'
            ids, soft = cdecode.decode_soft(self.decoder, z, c, self.MAX_SEQ_LEN, mode=sample_mode, temp=temp,
                                            min_length=min_length, out_keep=out_keep)
            return (ids, soft) if prepend_start_idx else (ids[:, 1:], soft[:, 1:])
        ids = cdecode.decode_hard(self.decoder, z, c, self.MAX_SEQ_LEN, mode=sample_mode, temp=temp,
                                  prevent_empty=prevent_empty, min_length=min_length, uniforms=uniforms,
                                  prepend_start_idx=prepend_start_idx, out_keep=out_keep)
        return ids if prepend_start_idx else ids[:, 1:]
