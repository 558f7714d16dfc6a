"""JSON metric log with the reference's interface (tb_json_logger.py:12-84): configure / log_value / export_to_json.
TensorBoard output (tensorboard_logger, not installed here) is dropped; the in-memory {it: {metric: value}} dict and
its result.json export - what api.get_result_for_model reads - are kept."""
import json
import os
import warnings

_configured = False
_log_dic = {}


def configure(logdir, json_fn=None, flush_secs=2):
    global _configured
    if _configured:
        raise ValueError('default logger already configured')
    if _log_dic:
        raise ValueError('_log_dic not empty! ' + str(_log_dic))
    _configured = True
    if json_fn and os.path.exists(json_fn):
        try:
            with open(json_fn) as fh:
                _log_dic.update({e['it']: e for e in json.load(fh)})
        except json.decoder.JSONDecodeError as e:
            warnings.warn('Couldnt decode {}: {}'.format(json_fn, str(e)))


def reset():
    global _configured
    _configured = False
    _log_dic.clear()


def log_value(name, value, step=None):
    if not _configured:
        raise ValueError('default logger is not configured. Call tb_json_logger.configure(logdir)')
    assert not _log_dic or step >= max(_log_dic.keys()), 'logging into the past: {} < {}'.format(step, max(_log_dic.keys()))
    _log_dic.setdefault(step, {'it': step})[name] = float(value)


def get_logged_values(step):
    return _log_dic[step]


def get_last_logged_values():
    return _log_dic[max(_log_dic.keys())] if _log_dic else {}


def export_to_json(json_fn, it_filter=lambda k, v: True, trunc_tail=None, write_empty=False):
    keep_from = (max(_log_dic.keys()) - trunc_tail) if (trunc_tail and _log_dic) else None
    rows = [_log_dic[it] for it in sorted(_log_dic)
            if it_filter(it, _log_dic[it]) and (keep_from is None or it >= keep_from)]
    if rows or write_empty:
        with open(json_fn, 'w') as fh:
            json.dump(rows, fh, indent=1)
