"""Metric log of the training run, JSON only.

Interface kept from the reference (tb_json_logger.py:12-84) because train_vae / main call it by these names:
`configure(logdir, json_fn)`, `log_value(name, value, step)`, `get_logged_values(step)`, `get_last_logged_values()`,
`export_to_json(json_fn, it_filter, trunc_tail, write_empty)`.  The TensorBoard event file the reference also writes
(tensorboard_logger, not installed in this image) is dropped; what `api.get_result_for_model` reads - result.json, a list
of {it: ..., metric: value, ...} rows ordered by iteration - is produced identically.
"""
import json
import os
import warnings


class _MetricStore:
    """{iteration: {'it': iteration, metric: value, ...}} with monotone iteration numbers."""

    def __init__(self):
        self.rows = {}
        self.active = False

    def open(self, resume_from=None):
        if self.active:
            raise ValueError('default logger already configured')
        if self.rows:
            raise ValueError('_log_dic not empty! ' + str(self.rows))
        self.active = True
        if resume_from and os.path.exists(resume_from):
            try:
                with open(resume_from) as fh:
                    for row in json.load(fh):
                        self.rows[row['it']] = row
            except json.decoder.JSONDecodeError as err:
                warnings.warn('Couldnt decode {}: {}'.format(resume_from, err))

    def put(self, name, value, step):
        if not self.active:
            raise ValueError('default logger is not configured. Call tb_json_logger.configure(logdir) first')
        if self.rows:
            newest = max(self.rows)
            assert step >= newest, 'logging into the past: {} < {}'.format(step, newest)
        row = self.rows.get(step)
        if row is None:
            row = self.rows[step] = {'it': step}
        row[name] = float(value)

    def select(self, keep, tail):
        its = sorted(self.rows)
        if tail and its:
            its = [i for i in its if i >= its[-1] - tail]
        return [self.rows[i] for i in its if keep(i, self.rows[i])]


_store = _MetricStore()
_log_dic = _store.rows  # the reference exposes the raw dict under this name


def configure(logdir, json_fn=None, flush_secs=2):
    _store.open(json_fn)


def reset():
    """Forget everything (tests; the reference has no equivalent because it runs once per process)."""
    _store.rows.clear()
    _store.active = False


def log_value(name, value, step=None):
    _store.put(name, value, step)


def get_logged_values(step):
    return _store.rows[step]


def get_last_logged_values():
    return _store.rows[max(_store.rows)] if _store.rows else {}


def export_to_json(json_fn, it_filter=lambda k, v: True, trunc_tail=None, write_empty=False):
    rows = _store.select(it_filter, trunc_tail)
    if not rows and not write_empty:
        return
    with open(json_fn, 'w') as fh:
        json.dump(rows, fh, indent=1)
