"""Module-as-config for the MI355X build: same attribute tree, flag syntax and JSON files as the reference's cfg.py.

  * every scalar (float/str/int/bool) leaf, also inside nested Bunch groups, becomes a `--a.b.c value` flag typed by
    the default's Python type (so, as in the reference, any non-empty string given to a bool flag is True);
  * `_override_config` applies a Namespace / dict of overrides, `_update_cfg` post-processes (tiny, partN, shared ->
    vae/full injection, paths, seed bump, dataset switch), `_save_config` writes config_overrides.json and
    config_complete.json, `_print` dumps the tree (reference cfg.py:14-147);
  * defaults are the reference's (cfg.py:150-372).
New keys for this build live under `hw` (device RNG seed stream, data-parallel world) and do not alter defaults.
"""
import json
import os

SCALARS = (float, str, int, bool)


class Bunch(dict):
    """dict with attribute access (config group)."""

    def __init__(self, *args, **kwds):
        super().__init__(*args, **kwds)
        self.__dict__ = self


def _bunchify(d):
    return Bunch({k: _bunchify(v) if isinstance(v, dict) else v for k, v in d.items()})


def _public(obj):
    return [k for k in dir(obj) if not k.startswith('_')]


def _cfg_import_export(cfg_interactor, cfg_, prefix='', mode='fill_parser'):
    """Walk the config tree.  mode: fill_parser (argparse flags) | fill_dict (flatten) | override (apply values)."""
    for key in _public(cfg_):
        val = getattr(cfg_, key)
        flat = prefix + key
        if type(val) in SCALARS:
            if mode == 'fill_parser':
                cfg_interactor.add_argument('--' + flat, type=type(val), help='default: {}'.format(val))
            elif mode == 'fill_dict':
                cfg_interactor[flat] = val
            elif mode == 'override':
                if flat in cfg_interactor:
                    setattr(cfg_, key, getattr(cfg_interactor, flat))
            else:
                raise ValueError(mode)
        elif type(val) == Bunch:
            _cfg_import_export(cfg_interactor, val, prefix=flat + '.', mode=mode)


def _override_config(args, cfg):
    _cfg_import_export(args, cfg, mode='override')


def _override_config_from_json(cfg, config_json):
    if config_json:
        with open(config_json) as fh:
            _cfg_import_export(Bunch(json.load(fh)), cfg, mode='override')


def _dump(obj, path):
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, 'w') as fh:
        json.dump(obj, fh, indent=2, sort_keys=True)


def _save_config(cfg_overrides, cfg_complete, savepath):
    _dump(vars(cfg_overrides), os.path.join(savepath, 'config_overrides.json'))
    flat = {}
    _cfg_import_export(flat, cfg_complete, mode='fill_dict')
    _dump(flat, os.path.join(savepath, 'config_complete.json'))


def _copy_to_nested_dict(cfg_):
    out = {}
    for key in _public(cfg_):
        val = getattr(cfg_, key)
        if type(val) in SCALARS:
            out[key] = val
        elif type(val) == Bunch:
            out[key] = _copy_to_nested_dict(val)
    return out


def _print(cfg_, prefix=''):
    for key in _public(cfg_):
        val = getattr(cfg_, key)
        if type(val) in SCALARS:
            print('{}{}\t{}'.format(prefix, key, val))
        elif type(val) == Bunch:
            print('{}{}:'.format(prefix, key))
            _print(val, prefix + '  |- ')


def _update_cfg():
    """Post-process special values after overrides have been applied."""
    global savepath, tbpath, resume_result_json, vocab_path, loadpath, seed
    savepath = os.path.join(savepath_toplevel, runname)
    tbpath = os.path.join(tb_toplevel, runname)
    if tiny:  # debug run: 100 iterations of batch 5, log every 10, checkpoint every 25, 30 samples
        shared.update(n_iter=100, cheaplog_every=10, expsvlog_every=25, batch_size=5)
        evals.sample_size = 30
        full.s_iter = shared.n_iter
        resume_result_json = False
    if partN > 1:
        assert phase > 0, 'split in parts only makes sense when doing per-phase split'
        cv = vae if phase == 1 else full
        cv.n_iter = cv.n_iter // partN
        cv.s_iter += part * cv.n_iter
        cv.expsvlog_every = min(cv.expsvlog_every, cv.n_iter)
        assert (cv.s_iter + cv.n_iter) % cv.expsvlog_every == 0, \
            'Final model wont be saved; n_iter={}, expsvlog_every {}'.format(cv.n_iter, cv.expsvlog_every)
    vae.update(shared)
    full.update(shared)
    if vocab_path == 'auto':
        vocab_path = os.path.join(savepath, 'vocab.dict')
    ckpt = os.path.join(savepath, 'model_{}.pt')
    vae.chkpt_path = full.chkpt_path = ckpt
    if loadpath == 'auto':
        if part == 0 and phase != 2:
            loadpath = ''
        else:
            loadpath = ckpt.format((vae if phase == 1 else full).s_iter)
    if seed and phase > 0:  # distinct seed per sub-run
        seed += (phase - 1) * partN + part
    for group, names in ((vae, dict(gen_samples_path='vae_gen.txt', eval_path='vae_eval.txt',
                                    fasta_gen_samples_path='vae_gen.fasta')),
                         (full, dict(gen_samples_path='full_gen.txt', samez_samples_path='full_samez.txt',
                                     posz_samples_path='full_posz.txt', interp_samples_path='full_interp.txt',
                                     eval_path='full_eval.txt', pos_eval_path='full.pos_eval.txt',
                                     fasta_gen_samples_path='full_gen.fasta', fasta_pos_samples_path='pos_gen.fasta'))):
        for field, fn in names.items():
            group[field] = os.path.join(savepath, fn)
    _set_dataset(dataset)


# ------------------------------------------------------------------------------------------------ defaults
config_json = ''
ignore_gpu = False
seed = 1238
tiny = False

tb_toplevel = 'tb'
savepath_toplevel = 'output'
runname = 'default'
datapath = 'data'
loadpath = 'auto'
vocab_path = 'auto'
phase = -1
part = 0
partN = 1
resume_result_json = True

_VAE_ITERS = 200000
vae = _bunchify(dict(
    batch_size=32, lr=1e-3, s_iter=0, n_iter=_VAE_ITERS,
    beta=dict(start=dict(val=1.0, iter=0), end=dict(val=2.0, iter=_VAE_ITERS // 5)),
    lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
    z_regu_loss='mmdrf',       # kl | mmd | mmdrf
    cheaplog_every=500, expsvlog_every=20000,
))

_FULL_ITERS = 50000
full = _bunchify(dict(
    batch_size=32, lrE=3e-4, lrG=3e-4, lrC=3e-4,
    n_iter=_FULL_ITERS, s_iter=_VAE_ITERS, classifier_min_length=5,
    beta=dict(start=dict(val=2.0, iter=_VAE_ITERS), end=dict(val=2.0, iter=_VAE_ITERS + _FULL_ITERS)),
    z_regu_loss='mmdrf',
    C_hard_sample_kwargs=dict(sample_mode='categorical'),
    G_soft_sample_kwargs=dict(sample_mode='none_softmax'),
    softmax_temp=dict(start=dict(iter=_VAE_ITERS, val=1.0), end=dict(iter=_VAE_ITERS + _FULL_ITERS, val=1.0)),
    lambda_e=0.1, lambda_c=1.0, lambda_z=0.1, lambda_u=0.1,
    lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
    cheaplog_every=50, expsvlog_every=2000,
))

shared = Bunch(clip_grad=5.0)

evals = _bunchify(dict(sample_size=2000, sample_modes=dict(beam=dict(sample_mode='beam', beam_size=5, n_best=3))))

losses = _bunchify(dict(wae_mmd=dict(sigma=7.0, kernel='gaussian', rf_dim=500, rf_resample=False)))

max_seq_len = 25

model = _bunchify(dict(
    z_dim=100, c_dim=2, emb_dim=150, pretrained_emb=None, freeze_embeddings=False, flow=0, flow_type='',
    E_args=dict(h_dim=80, biGRU=True, layers=1, p_dropout=0.0, cell='gru'),   # cell: 'gru' (reference) | 'lstm' (extension)
    G_args=dict(
        G_class='gru',
        GRU_args=dict(p_word_dropout=0.3, p_out_dropout=0.3, skip_connetions=False, cell='gru', layers=1),   # layers > 1: extension
        deconv_args=dict(max_seq_len=max_seq_len, num_filters=100, kernel_size=4, num_deconv_layers=3, useRNN=False,
                         temperature=1.0, use_batch_norm=True, num_conv_layers=2, add_final_conv_layer=True),
    ),
    C_args=dict(min_filter_width=3, max_filter_width=5, num_filters=100, dropout=0.5),
))

# MI355X build additions (absent from the reference)
hw = _bunchify(dict(
    device_rng=True,      # draw eps / dropout masks / z_prior on the device (Philox) instead of torch+numpy host streams
    world_size=1,         # data-parallel ranks (set by the launcher from WORLD_SIZE)
    synthetic_data=True,  # random-vocab peptide batches (the reference's curated CSVs are not reproducible)
    synthetic_size=20000,
    graph=False,          # replay the training step from ONE captured hipGraph (train_vae.GraphedTrainStep): pays when the host
                          # enqueue bounds the step (small batches); single rank, device_rng, dense decoder batches
    dump_states=True,     # main.py --phase 1 ends with the encode pass (states_<split>_<n_iter>: what sample_pipeline.py reads)
    dtype='f32',          # 'f32': f32-grade recurrent products (the parity path) | 'bf16': bf16 recurrent products - operands
                          # rounded to bf16, one bf16 MFMA per block, f32 accumulation / storage / master weights
                          # (BASELINE.json configs[1]/[4]); --hw.dtype bf16
    ragged_decoder=False,  # training only: decoder rows leave the recurrence once their remaining targets are <pad>.
                           # Exact (tests/test_gpu_parity.py) but OFF: at batch 2048 a step launch is one round of workgroups,
                           # i.e. latency-bound - dropping 40 % of them left its duration unchanged
))

dataset = 'amp'
data_kwargs, data_prefixes, attributes = None, None, None

DATA_ROOT = './PATH_TO_DATA/'
_FACTORS = {'amp=amp_posc': 20, 'amp=amp_posnc': 10, 'amp=amp_negc': 20, 'amp=amp_negnc': 10,
            'tox=tox_posc': 20, 'tox=tox_posnc': 10, 'tox=tox_negc': 20, 'tox=tox_negnc': 10,
            'sol': 20, 'anticancer': 20, 'antihyper': 20, 'hormone': 20}


def _iter(subset, weighted=False):
    spec = Bunch(subset=subset)
    if weighted:
        spec.update(weighted_random_sample=True, sample_prob_factors=_FACTORS)
    return spec


amp = Bunch(
    data_kwargs=Bunch(
        lower=False,
        data_path=os.environ.get('DATA_PATH_AMP', DATA_ROOT + 'amp/'),
        data_format='csv',
        csv_files=['unlab.csv', 'amp_lab.csv', 'tox_lab.csv', 'sol_lab.csv', 'anticancer.csv', 'antihypertensive.csv',
                   'cell-cell.csv'],
        iteratorspecs=Bunch(
            train_vae=_iter(['split=train'], True),
            train_amp_lab=_iter(['split=train', 'amp'], True),
            hld_vae=_iter(['split=val'], True),
            hld_unl=_iter(['split=val', '^amp']),
            hld_amppos=_iter(['split=val', 'amp=amp_posc,amp_posnc']),
            hld_ampneg=_iter(['split=val', 'amp=amp_negc,amp_negnc']),
        ),
        fixed_vocab_path=DATA_ROOT + 'amp/vocab.dict',
        split_seed=1288,
    ),
    data_prefixes=Bunch(dataset_type='bio', dataset_unl='amp_unlabeled', dataset_lab='amp_labeled'),
    attributes=[
        ('amp', {'amp_negnc': 0, 'amp_negc': 0, 'amp_posc': 1, 'amp_posnc': 1, 'na': -1}),
        ('tox', {'tox_negc': 0, 'tox_negnc': 0, 'tox_posc': 1, 'tox_posnc': 1, 'na': -1}),
        ('sol', {'sol_neg': 0, 'sol_pos': 1, 'na': -1}),
        ('anticancer', {'anticancer': 1, 'na': -1}),
        ('antihyper', {'antihyper': 1, 'na': -1}),
        ('hormone', {'cell': 1, 'na': -1}),
    ],
)


def _set_dataset(name):
    global data_kwargs, data_prefixes, attributes
    if name != 'amp':
        raise ValueError('unknown dataset ' + name)
    data_kwargs, data_prefixes, attributes = amp.data_kwargs, amp.data_prefixes, amp.attributes


_set_dataset(dataset)
