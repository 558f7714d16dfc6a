"""Phase-1 entry point on the MI355X path: `python main.py [--tiny 1] [--phase 1] [--<cfg.key> value]`
(counterpart of the reference's main.py:30-86; same flag syntax, same files under output/<run>/).

cfg -> seeds -> json logger -> data -> RNN_VAE -> (load) -> train_vae -> sample `cfg.evals.sample_size` peptides ->
vae_gen.txt -> result.json / vae_result.json.  Data: the reference's torchtext-0.3.1 CSV loader and its curated files
are not reproducible, so batches come from cpg.synth.SyntheticPeptideLoader (cfg.hw.synthetic_data).
Launch N ranks with `python -m torch.distributed.run --nproc-per-node N main.py ...` for data-parallel training.
"""
import argparse
import logging
import random
from os.path import join as pjoin

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
log.setLevel(logging.INFO)
if not log.handlers:
    h = logging.StreamHandler()
    h.setFormatter(logging.Formatter('%(asctime)s - %(levelname)s(%(name)s): %(message)s'))
    log.addHandler(h)


def run(argv=None):
    parser = argparse.ArgumentParser(argument_default=argparse.SUPPRESS, description='Override config float & string values')
    cfg._cfg_import_export(parser, cfg, mode='fill_parser')
    args = parser.parse_args(argv)
    cfg._override_config(args, cfg)
    cfg._update_cfg()
    world, rank, local = cdist.init()
    if rank == 0:
        cfg._print(cfg)
        cfg._save_config(args, cfg, cfg.savepath)
    if not torch.cuda.is_available() or cfg.ignore_gpu:
        raise RuntimeError('the MI355X build has no CPU path (cfg.ignore_gpu / no visible GPU)')
    local = cdist.local_device(local)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    cfg.seed = cfg.seed if cfg.seed else random.randint(1, 10000)
    log.info('Random seed: {}'.format(cfg.seed))
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed + rank)
    random.seed(cfg.seed)
    tb_json_logger.configure(cfg.tbpath, pjoin(cfg.savepath, 'result.json') if cfg.resume_result_json else None)

    dataset = SyntheticPeptideLoader(cfg.vae.batch_size, cfg.max_seq_len, device, size=cfg.hw.synthetic_size,
                                     seed=cfg.seed, rank=rank)
    dataset.print_stats()
    if rank == 0:
        utils.save_vocab(dataset.TEXT.vocab, cfg.vocab_path)

    model = RNN_VAE(n_vocab=dataset.n_vocab, max_seq_len=cfg.max_seq_len, **cfg.model).to(device)
    model.device = device
    log.info(model)
    if cfg.loadpath:
        model.load_state_dict(torch.load(cfg.loadpath, map_location=device))
        log.info('Loaded model from ' + cfg.loadpath)
    losses.rf.clear()
    if cfg.hw.device_rng:
        model.use_device_rng(cfg.seed + 7919 * rank)
        losses.set_prior_sampler(lambda z: model._randn(z.shape[0], z.shape[1]))
    cdist.broadcast_params(model.parameters())
    reduce_fn = None
    if world > 1:
        reduce_fn = cdist.allreduce_sum
        losses.set_distributed(reduce_fn, world)

    if cfg.phase in [1]:
        train_vae(cfg.vae, model, dataset, reduce_fn=reduce_fn, world=world, rank=rank)
        if rank == 0:
            log.info("Evaluating base vae...")
            with torch.no_grad():
                samples, _, _ = model.generate_sentences(cfg.evals.sample_size, sample_mode='categorical')
            utils.write_gen_samples(dataset.idx2sentences(samples.cpu(), False), cfg.vae.gen_samples_path)
    if rank == 0:
        log.info('saving result.json and vae_result.json at {}'.format(cfg.savepath))
        tb_json_logger.export_to_json(pjoin(cfg.savepath, 'result.json'))
        tb_json_logger.export_to_json(pjoin(cfg.savepath, 'vae_result.json'), it_filter=lambda k, v: k <= cfg.vae.n_iter)


if __name__ == "__main__":
    run()
