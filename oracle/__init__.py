"""CPU oracle: a numpy restatement of the reference's WAE-training + CLaSS-sampling hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is the checker the HIP path is compared against; it
is never the thing shipped or measured.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product path (the package under
`controlled-peptide-generation_amd/`) never imports it and fails loudly when the HIP library
is missing.

Pinning: every function here is checked in `tests/test_oracle_golden.py` against vectors the
real reference produced when imported in the build container (`tests/golden/make_golden.py`,
fixtures `tests/golden/*.npz`).  The GRU cell itself lives in a third-party dependency that is
absent from /root/reference (PyTorch `nn.GRU`, pinned `pytorch=1.7.1` in amp_gen.yml:8); its
published algorithm (gate order r,z,n; `n = tanh(W_in x + b_in + r*(W_hn h + b_hn))`;
`h' = (1-z)*n + z*h`) is restated in `oracle/gru.py` and anchored on the reference's call
sites models/encoder.py:25-30,42 and models/decoder.py:40-41,77,98.

The LSTM cell (`oracle/lstm.py`) has NO counterpart in the reference (every RNN
there is nn.GRU): **parity unpinned** against the reference; it is pinned to torch.nn.LSTM only.
"""

import contextlib as _contextlib


@_contextlib.contextmanager
def precision(dtype):
    """`with oracle.precision(numpy.float64): ...` - the restatement's working dtype (F32 in gru / wae / decode / optim: every
    intermediate is cast to it) for the duration of the block.  With float64 parameters and inputs handed in, the whole
    evaluation accumulates in float64: the reference value the f32-grade product forms of the HIP path are measured against
    where float32 rounding of the oracle itself would blur the comparison (saturated gates, trained weights, long sequences).
    Default float32 = the reference's arithmetic."""
    import numpy as _np
    from . import decode, gru, optim, wae
    mods = (gru, wae, decode, optim)
    prev = [m.F32 for m in mods]
    for m in mods:
        m.F32 = _np.dtype(dtype).type
    try:
        yield
    finally:
        for m, p in zip(mods, prev):
            m.F32 = p


def as_f64(tree):
    """dict of arrays -> the same dict with every floating array as float64 (integer / mask arrays untouched)."""
    import numpy as _np
    return {k: (v.astype(_np.float64) if isinstance(v, _np.ndarray) and v.dtype.kind == "f" else v) for k, v in tree.items()}
