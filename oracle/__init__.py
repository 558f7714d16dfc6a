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
