"""Phase-1 WAE/VAE training loop on the MI355X hot path (counterpart of the reference's train_vae.py:13-68).

Per iteration, as in the reference: batch -> model forward (encoder, reparameterisation, teacher-forced decoder) ->
recon CE + KL + full-kernel MMD + random-feature MMD (all four are evaluated every step; cfgv.z_regu_loss picks the
one that regularises) -> loss = recon + beta*regu + lambda_L1*|logvar|_1 + lambda_KL*KL(N(mu,s)||N(mu,I)) -> backward
-> global-norm clip -> Adam, with n_iter+1 iterations, the same metric names, logging cadence and checkpoint naming.
Differences: the optimiser is the fused HIP clip+Adam (same observable semantics incl. the duplicate-embedding quirk,
SURVEY F6); metrics are read back from the device only on logging iterations (the reference calls .item() ten times
every step); `train_step` is exposed so bench.py times exactly this body.
"""
import sys

import torch

import losses
import cfg
import utils
from cpg.optim import FusedAdamClip
from models.mutils import save_model
from tb_json_logger import log_value


def make_optimizer(cfgv, model, reduce_fn=None, world=1, async_reduce_fn=None):
    """Gradient buckets in the order the backward pass completes them (models.model marks the boundaries): the decoder's own
    parameters, then the encoder heads; the shared embedding and the encoder recurrence are final only at the very end."""
    emb = model.word_emb.weight
    buckets = [('decoder', [p for p in model.decoder.parameters() if p is not emb]),
               ('encoder_heads', list(model.encoder.q_mu.parameters()) + list(model.encoder.q_logvar.parameters()))]
    if async_reduce_fn is None and world > 1:
        from cpg import dist as cdist
        async_reduce_fn = cdist.allreduce_sum_async if cdist.is_initialized() else None
    return FusedAdamClip(model.vae_params(), lr=cfgv.lr, max_norm=cfgv.clip_grad, reduce_fn=reduce_fn, world=world, buckets=buckets,
                         async_reduce_fn=async_reduce_fn)


def train_step(cfgv, model, trainer, text, it, rnd=None, z_priors=(None, None), weights_dev=None):
    """One full iteration.  Returns a dict of device scalars (no host sync).  weights_dev (optional, device float[4] =
    (1, beta, lambda_L1, lambda_KL)): the loss weights are read from that tensor by the kernels instead of being launch arguments -
    what a captured step needs (GraphedTrainStep rewrites it between replays)."""
    beta = utils.anneal(cfgv.beta, it)
    # the trainer consumes the logits only through recon_dec (pad targets ignored): the decoder may skip dead rows
    ragged_before, fused_before = model.decoder.ragged, model.fused_train
    model.decoder.ragged = bool(cfg.hw.ragged_decoder)
    model.fused_train = True     # the trainer consumes the step's intermediates through losses.* only: fused latent / reconstruction nodes
    model.decoder.recon_targets = text
    try:
        (z_mu, z_logvar), (z, c), dec_logits = model(text, q_c='prior', sample_z=1, rnd=rnd)
    finally:
        model.decoder.ragged, model.fused_train, model.decoder.recon_targets = ragged_before, fused_before, None
    recon_loss = losses.recon_dec(text, dec_logits)
    kl_loss, z_logvar_KL_penalty, z_logvar_L1 = losses.latent_terms(z_mu, z_logvar)
    # the full-kernel MMD couples every pair of rows of the GLOBAL batch: as the regulariser it is evaluated on the all-gathered
    # batch (exact under data parallelism); as a logged-only value it stays rank-local
    wae_mmd_loss = losses.wae_mmd_gaussianprior(z, method='full_kernel', z_prior=z_priors[0],
                                                global_batch=(cfgv.z_regu_loss == 'mmd' and trainer.world > 1))
    wae_mmdrf_loss = losses.wae_mmd_gaussianprior(z, method='rf', z_prior=z_priors[1])
    z_regu_loss = {'kl': kl_loss, 'mmd': wae_mmd_loss, 'mmdrf': wae_mmdrf_loss}[cfgv.z_regu_loss]
    # loss = recon + beta * regu + lambda_L1 * L1 + lambda_KL * KLpenalty (train_vae.py:35-37), one launch
    from cpg.ops import WeightedSumFn
    weights = weights_dev if weights_dev is not None else (1.0, beta, cfgv.lambda_logvar_L1, cfgv.lambda_logvar_KL)
    loss = WeightedSumFn.apply(weights, recon_loss, z_regu_loss, z_logvar_L1, z_logvar_KL_penalty)
    trainer.zero_grad()
    trainer.backward(loss)
    if model.rng is not None:
        trainer.attach_rng(model.rng)   # step() advances the Philox base together with its own iteration counter: one launch
    trainer.step()
    if model.rng is not None:
        model.rng.end_step()     # the device-side Philox base moves past this step's draws; host offsets restart at 0 (a no-op
        #                          when step() has already done it)
    return dict(z_mu=z_mu, z_logvar=z_logvar, z_logvar_L1=z_logvar_L1, z_logvar_KL_penalty=z_logvar_KL_penalty, L_vae=loss,
                L_vae_recon=recon_loss, L_vae_kl=kl_loss, L_wae_mmd=wae_mmd_loss, L_wae_mmdrf=wae_mmdrf_loss, beta=beta)


class GraphedTrainStep:
    """train_step captured ONCE into a hipGraph (torch.cuda.CUDAGraph) and replayed: ~185 launches become one graph launch, the
    host's work per step shrinks to a batch copy, one fill and the replay (what small configurations - the reference's default
    batch of 32 - are bound by).  What makes a replay a NEW step although its launch arguments are frozen:
      * the batch lives in a static buffer that is overwritten before every replay;
      * the Philox offsets of every random draw are relative to a device-side base that the step itself advances (DeviceRng);
      * Adam's step number is formed on the device from an iteration counter the step increments (FusedAdamClip.iter_dev);
      * the annealed beta is read from device memory (`weights`), rewritten by a fill launch before the replay.
    Needs model.use_device_rng(...) (host generators cannot be captured) and a single rank (collectives stay eager)."""

    def __init__(self, cfgv, model, trainer, warmup=3):
        assert model.rng is not None, "a captured step needs the device random streams (model.use_device_rng)"
        assert trainer.world == 1, "captured steps are single-rank (the gradient all-reduce stays eager)"
        self.cfgv, self.model, self.trainer, self.warmup = cfgv, model, trainer, int(warmup)
        self.text = self.weights = self.graph = self.out = None
        self.calls = 0

    def release(self):
        """Drop the captured graph (the object can be called again: it re-captures); parked scratch buffers are freed with the last
        live graph (cpg.ops.graph_released)."""
        if self.graph is not None:
            self.graph = None
            from cpg import ops as _ops
            _ops.graph_released()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _eager(self, it):
        self.weights[1:2].fill_(float(utils.anneal(self.cfgv.beta, it)))
        return train_step(self.cfgv, self.model, self.trainer, self.text, it, weights_dev=self.weights)

    def __call__(self, text, it):
        """Iteration `it` on batch `text`.  The first `warmup` calls run eagerly ON THE CAPTURE STREAM (stream-keyed workspaces,
        persistent-kernel scratch, side streams and LDS opt-ins then exist before the capture: allocating pinned memory or querying
        occupancy inside a capture is not allowed); the next call captures the step - a capture records, it does not execute -
        and every call from then on is one graph replay."""
        if self.text is None:
            dev = text.device
            self.text = text.clone()
            self.weights = torch.tensor([1.0, 0.0, float(self.cfgv.lambda_logvar_L1), float(self.cfgv.lambda_logvar_KL)], device=dev)
            self.stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream()
        if self.graph is None:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.text.copy_(text, non_blocking=True)
                if self.calls < self.warmup:
                    out = self._eager(it)
                    cur.wait_stream(self.stream)
                    self.calls += 1
                    return out
            cur.wait_stream(self.stream)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            from cpg import ops as _ops
            _ops.graph_captured()   # scratch buffers the graph points into must outlive any later growth (cpg.ops._retire)
            try:
                with torch.cuda.graph(graph, stream=self.stream):
                    # recorded, not run: the replay below is this iteration.  (No beta fill in here: it would be frozen into the graph.)
                    self.out = train_step(self.cfgv, self.model, self.trainer, self.text, it, weights_dev=self.weights)
            except BaseException:
                _ops.graph_released()   # a failed capture pins nothing
                raise
            self.graph = graph
        else:
            self.text.copy_(text, non_blocking=True)
        self.weights[1:2].fill_(float(utils.anneal(self.cfgv.beta, it)))
        self.graph.replay()
        self.calls += 1
        out = dict(self.out)
        out['beta'] = utils.anneal(self.cfgv.beta, it)
        return out


def train_vae(cfgv, model, dataset, reduce_fn=None, world=1, rank=0):
    print('Training base vae ...')
    trainer = make_optimizer(cfgv, model, reduce_fn, world)
    # cfg.hw.graph: replay the step from one captured hipGraph (single rank, device random streams, dense decoder batches)
    graphed = None
    if cfg.hw.graph:
        unmet = [why for cond, why in ((world == 1, 'more than one rank (the gradient all-reduce stays eager)'),
                                       (model.rng is not None, 'host random generators (needs model.use_device_rng)'),
                                       (not cfg.hw.ragged_decoder, 'cfg.hw.ragged_decoder (per-step live-row counts change per batch)'))
                 if not cond]
        if unmet:
            print('WARNING: cfg.hw.graph requested but NOT applied - eager steps instead: ' + '; '.join(unmet), file=sys.stderr)
        else:
            graphed = GraphedTrainStep(cfgv, model, trainer)
    for it in range(cfgv.s_iter, cfgv.s_iter + cfgv.n_iter + 1):
        logging_it = it % cfgv.cheaplog_every == 0 or it % cfgv.expsvlog_every == 0
        inputs = dataset.next_batch('train_vae')
        t = graphed(inputs.text, it) if graphed is not None else train_step(cfgv, model, trainer, inputs.text, it)
        if logging_it:
            from cpg import ops
            ops.check_persistent()   # a timed-out inter-workgroup wait of a persistent launch raises here (logging syncs anyway)
        if logging_it and rank == 0:
            from cpg.ops import latent_sums
            with torch.no_grad():
                s = latent_sums(t['z_mu'].detach(), t['z_logvar'].detach()).cpu()
            n = t['z_mu'].numel()
            vals = {'z_mu_L1': s[3].item() / n, 'z_logvar': s[4].item() / n,
                    'z_logvar_L1': t['z_logvar_L1'].item(), 'z_logvar_KL_penalty': t['z_logvar_KL_penalty'].item(),
                    'L_vae': t['L_vae'].item(), 'L_vae_recon': t['L_vae_recon'].item(), 'L_vae_kl': t['L_vae_kl'].item(),
                    'L_wae_mmd': t['L_wae_mmd'].item(), 'L_wae_mmdrf': t['L_wae_mmdrf'].item(), 'beta': t['beta']}
            for k, v in vals.items():
                log_value('train_' + k, v, it)
            print('ITER {} TRAINING (phase 1). loss_vae: {:.4f}; loss_recon: {:.4f}; loss_kl: {:.4f}; loss_mmd: {:.4f}; '
                  'Grad_norm: {:.4e} '.format(it, vals['L_vae'], vals['L_vae_recon'], vals['L_vae_kl'], vals['L_wae_mmd'],
                                              trainer.grad_norm().item()))
            log_sent, _, _ = model.generate_sentences(1, sample_mode='categorical')
            if model.rng is not None:
                model.rng.end_step()     # draws made outside a training step: move the device base past them as well
            print('Sample (cat T=1.0): "{}"'.format(dataset.idx2sentence(log_sent.squeeze())))
            sys.stdout.flush()
        if it % cfgv.expsvlog_every == 0 and it > 0 and rank == 0:
            save_model(model, cfgv.chkpt_path.format(it))
    return trainer
