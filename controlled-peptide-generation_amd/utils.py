"""Small host helpers (counterpart of the reference's utils.py:17-31,42-67): beta schedule, sample/vocab writers."""
import codecs
import os


def check_dir_exists(fn):
    d = os.path.dirname(fn)
    if d and not os.path.exists(d):
        os.makedirs(d)


def interpolate(start_val, end_val, start_iter, end_iter, current_iter):
    """Piecewise-linear schedule: start_val before start_iter, end_val from end_iter on (utils.py:51-58)."""
    if current_iter < start_iter:
        return start_val
    if current_iter >= end_iter:
        return end_val
    frac = (current_iter - start_iter) / (end_iter - start_iter)
    return start_val + (end_val - start_val) * frac


def anneal(cfgan, it):
    return interpolate(cfgan.start.val, cfgan.end.val, cfgan.start.iter, cfgan.end.iter, it)


def _open_for_write(fn, encoding=None):
    check_dir_exists(fn)
    return open(fn, 'w', encoding=encoding)


def write_gen_samples(samples, fn, c_lab=None):
    """vae_gen.txt format (what the reference's utils.py:17-31 writes and its evals read back): one sample per line, or - with
    labels - a `label: <y>` line ahead of each sample."""
    n = len(samples)
    if c_lab is not None and c_lab.nelement() != n:
        raise AssertionError('sizes dont match')
    with _open_for_write(fn) as out:
        if c_lab is None:
            out.write(''.join(s + '\n' for s in samples))
        else:
            for y, s in zip(c_lab, samples):
                out.write('label: {}\n'.format(y) + s + '\n')   # format(): a 0-dim tensor prints as its number
    print('Saving %d samples %s labels' % (n, 'without' if c_lab is None else 'with'))


def save_vocab(vocab, fn):
    """vocab.dict format (utils.py:42-47): `<token> <index>` per line, UTF-8, in the vocabulary's own order."""
    with _open_for_write(fn, encoding='utf-8') as out:
        out.writelines('%s %d\n' % (tok, ix) for tok, ix in vocab.stoi.items())
    print('Saved vocab to ' + fn)
