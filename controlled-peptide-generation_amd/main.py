"""Phase-1 entry point on the MI355X path: `python main.py [--tiny 1] [--phase 1] [--<cfg.key> value]`.

Counterpart of the reference's main.py:30-86 - same flag syntax and the same files under output/<run>/
(config_overrides.json, config_complete.json, vocab.dict, model_<it>.pt, vae_gen.txt, result.json, vae_result.json).
Stages: configuration -> seeding -> metric log -> data -> model (-> checkpoint) -> train_vae -> prior samples -> export.
Data: the reference's torchtext-0.3.1 CSV loader and its curated files are not reproducible; batches come
from cpg.synth.SyntheticPeptideLoader.  `python -m torch.distributed.run --nproc-per-node N main.py ...` trains data-parallel.
"""
import argparse
import logging
import os
import random

import numpy as np
import torch

import cfg
import losses
import tb_json_logger
import utils
from cpg import dist as cdist
from cpg.synth import SyntheticPeptideLoader
from models.model import RNN_VAE
from train_vae import train_vae

log = logging.getLogger()


def _console_logging():
    log.setLevel(logging.INFO)
    if log.handlers:
        return
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter('%(asctime)s - %(levelname)s(%(name)s): %(message)s'))
    log.addHandler(handler)


def _configure(argv):
    parser = argparse.ArgumentParser(argument_default=argparse.SUPPRESS, description='Override config float & string values')
    cfg._cfg_import_export(parser, cfg, mode='fill_parser')
    overrides = parser.parse_args(argv)
    cfg._override_config(overrides, cfg)
    cfg._update_cfg()
    return overrides


def _seed_everything(rank):
    if not cfg.seed:
        cfg.seed = random.randint(1, 10000)
    log.info('Random seed: {}'.format(cfg.seed))
    torch.manual_seed(cfg.seed)          # identical initial weights on every rank (re-seeded per rank after the model is built)
    np.random.seed(cfg.seed + rank)      # host-side draws differ per rank
    random.seed(cfg.seed)


def _build_model(n_vocab, device, rank, world):
    from cpg import ops
    ops.set_compute_mode(cfg.hw.dtype)
    model = RNN_VAE(n_vocab=n_vocab, max_seq_len=cfg.max_seq_len, **cfg.model).to(device)
    model.device = device
    log.info(model)
    if cfg.loadpath:
        model.load_state_dict(torch.load(cfg.loadpath, map_location=device))
        log.info('Loaded model from ' + cfg.loadpath)
    losses.rf.clear()
    # the random-feature basis must be THE SAME on every rank (its feature sums are all-reduced): draw it now, from the
    # common torch seed, then give the per-step host draws (eps, z_prior when device_rng is off) a rank-distinct stream
    losses._rf_basis(torch.zeros(1, cfg.model.z_dim, device=device), cfg.losses.wae_mmd.rf_dim, False)
    if world > 1:
        if cfg.losses.wae_mmd.rf_resample:
            raise NotImplementedError('rf_resample redraws the basis per call from rank-local streams: not data-parallel exact')
        torch.manual_seed(cfg.seed + 104729 * rank)
    if cfg.hw.device_rng:
        model.use_device_rng(cfg.seed + 7919 * rank)
        losses.set_prior_sampler(lambda z: model._randn(z.shape[0], z.shape[1]))
    cdist.broadcast_params(model.parameters())
    reduce_fn = cdist.allreduce_sum if world > 1 else None
    if reduce_fn is not None:
        losses.set_distributed(reduce_fn, world, gather_fn=cdist.allgather_equal, rank=rank)
    return model, reduce_fn


def run(argv=None):
    _console_logging()
    overrides = _configure(argv)
    world, rank, local = cdist.init()
    lead = rank == 0
    if lead:
        cfg._print(cfg)
        cfg._save_config(overrides, cfg, cfg.savepath)
    if cfg.ignore_gpu or not torch.cuda.is_available():
        raise RuntimeError('the MI355X build has no CPU path (cfg.ignore_gpu / no visible GPU)')
    local = cdist.local_device(local)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    _seed_everything(rank)
    resume = os.path.join(cfg.savepath, 'result.json') if cfg.resume_result_json else None
    tb_json_logger.configure(cfg.tbpath, resume)

    dataset = SyntheticPeptideLoader(cfg.vae.batch_size, cfg.max_seq_len, device, size=cfg.hw.synthetic_size,
                                     seed=cfg.seed, rank=rank)
    dataset.print_stats()
    if lead:
        utils.save_vocab(dataset.TEXT.vocab, cfg.vocab_path)
    model, reduce_fn = _build_model(dataset.n_vocab, device, rank, world)

    if cfg.phase in [1]:
        train_vae(cfg.vae, model, dataset, reduce_fn=reduce_fn, world=world, rank=rank)
        if lead:
            log.info("Evaluating base vae...")
            with torch.no_grad():
                samples, _, _ = model.generate_sentences(cfg.evals.sample_size, sample_mode='categorical')
            utils.write_gen_samples(dataset.idx2sentences(samples.cpu(), False), cfg.vae.gen_samples_path)
            if cfg.hw.dump_states:
                # the encode pass the reference runs offline (vis/scripts/build_index.py:93-118, "run static_eval first"):
                # states_<split>_<n_iter> under savepath is what sample_pipeline.py fits Q_xi(z) and the z-space classifiers on
                from sample_pipeline import dump_encodings
                for split in ('train', 'val', 'test'):
                    ids, labels = dataset.subset(split)
                    fn = dump_encodings(model, ids, labels, split, cfg.savepath, cfg.vae.n_iter)
                    log.info('encodings of {} {} sequences -> {}'.format(ids.shape[0], split, fn))
    if lead:
        log.info('saving result.json and vae_result.json at {}'.format(cfg.savepath))
        tb_json_logger.export_to_json(os.path.join(cfg.savepath, 'result.json'))
        tb_json_logger.export_to_json(os.path.join(cfg.savepath, 'vae_result.json'),
                                      it_filter=lambda it, row: it <= cfg.vae.n_iter)


if __name__ == "__main__":
    run()
