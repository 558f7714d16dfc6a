#!/usr/bin/env python3
"""Stage times of one CLaSS round (sample_pipeline.run_rounds at bench.py's configs[3] setup) with a device sync after every
stage - where the wall time outside the decode kernel goes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import sample_pipeline as sp  # noqa: E402

dev = torch.device("cuda")
m, Q, ds = bench.class_setup(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
mode = sys.argv[2] if len(sys.argv) > 2 else "beam"
sp.run_rounds(m, ds, Q, 65536, 10 ** 9, max_rounds=1, sample_mode=mode)


def t(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"{label:28s} {1e3 * (time.perf_counter() - t0):8.1f} ms")
    return r


for rep in range(2):
    print("--- round", rep)
    t0 = time.perf_counter()
    z, probs, accum, acc = t("rejection_sample", lambda: Q.rejection_sample(N, return_device=True, shard=(0, 1)))
    c = torch.zeros(z.shape[0], 2, device=dev)
    c[:, 1] = 1.0
    ids, evals = t("decode_ids_from_z", lambda: sp.decode_ids_from_z(z, c, m, mode))
    letters, n_res = t("residue_rows", lambda: sp.residue_rows(ids, ds.n_vocab))
    names = Q.score_names()
    frame = {'letters': letters, 'n_res': n_res, 'z': z, 'accept_z': acc.to(torch.bool), names[0]: accum}
    for i, nm in enumerate(names[1:]):
        frame[nm] = probs[i]
    frame, seen = t("dedup_frame", lambda: sp.dedup_frame(frame, None))
    df = t("frames_to_dataframe", lambda: sp.frames_to_dataframe([frame], ds))
    print(f"{'total':28s} {1e3 * (time.perf_counter() - t0):8.1f} ms   kept {len(df)} accepted {int(df['accept'].sum())}")
