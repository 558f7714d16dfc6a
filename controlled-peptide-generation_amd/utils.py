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


def write_gen_samples(samples, fn, c_lab=None):
    check_dir_exists(fn)
    with open(fn, 'w+') as f:
        if c_lab is not None:
            assert c_lab.nelement() == len(samples), 'sizes dont match'
            print("Saving %d samples with labels" % len(samples))
            f.writelines('label: {}\n{}\n'.format(y, s) for y, s in zip(c_lab, samples))
        else:
            print("Saving %d samples without labels" % len(samples))
            f.write('\n'.join(samples) + '\n')


def save_vocab(vocab, fn):
    check_dir_exists(fn)
    with codecs.open(fn, "w", "utf-8") as f:
        for word, ix in vocab.stoi.items():
            f.write(word + " " + str(ix) + "\n")
    print('Saved vocab to ' + fn)
