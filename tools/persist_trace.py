#!/usr/bin/env python3
"""Per-wave timeline of the persistent forward kernel from a trace build (tools/variant_build.sh gru_persist.hip trace
-DCPG_PERSIST_TRACE=1;  CPG_LIB_PATH=build_variants/libcpg_trace.so python tools/persist_trace.py).
Stamps per step (10-ns ticks): 0 step top, 1 wait done, 2 first k-block multiplied (first A loads back), 3 last MFMA issued,
4 cell done (accumulators read), 5 publish drained, 6 arrival + gate / state stores issued."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cpg import ops  # noqa: E402

B, H, T, V, W = 2048, 512, 25, 24, 8
dev = torch.device("cuda")
ops.set_compute_mode(os.environ.get("KB_DTYPE", "f32"))
g = torch.Generator().manual_seed(0)
w_hh = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(dev)
b_hh = (torch.randn(3 * H, generator=g) * 0.1).to(dev)
tab = (torch.randn(V, 3 * H, generator=g) * 0.3).to(dev)
rowc = (torch.randn(B, 3 * H, generator=g) * 0.3).to(dev)
tok = torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev)
hs = torch.zeros(T + 1, B, H, device=dev)
hs[0] = torch.randn(B, H, generator=g).to(dev)
gates = torch.empty(T, 4, B, H, device=dev)
for _ in range(3):
    ops.gru_seq_fwd_persistent(T, B, H, False, w_hh, b_hh, tok, tab, rowc, None, hs, gates)
torch.cuda.synchronize()
ent = next(v for k, v in ops._persist_scratch.items() if k[0] == "gru")
nwg = (B // 256) * (H // 16)
n = nwg * W * T * 8
tr = ent[0][-n * 8:].view(torch.int64).cpu().numpy().reshape(nwg, W, T, 8).astype(np.float64) * 0.01   # us
t0 = tr[:, :, :, 0].min()
names = ["wait", "first-kb", "mfma-loop", "cell", "publish+drain", "arrive+stores"]
steps = slice(5, 22)
d = np.diff(tr[:, :, steps, :7], axis=3)                     # [wg, wave, step, 6]
print("mean phase durations over workgroups / waves / steps 5..21 (us):")
for i, nm in enumerate(names):
    print(f"  {nm:16s} {d[..., i].mean():6.2f}   (p10 {np.percentile(d[..., i], 10):5.2f}  p90 {np.percentile(d[..., i], 90):5.2f})")
per = np.diff(tr[:, :, steps, 0], axis=2)
print(f"step period {per.mean():.2f} us (p10 {np.percentile(per, 10):.2f}, p90 {np.percentile(per, 90):.2f})")
for wg in (0, 1, 8, 255):
    print(f"workgroup {wg}: step-top offsets of waves 0..7 relative to wave 0 at step 10 (us):",
          np.round(tr[wg, :, 10, 0] - tr[wg, 0, 10, 0], 2).tolist())
    print("   step 10 stamps of wave 0 / wave 4 (same SIMD), relative to wave 0's step top:")
    for w in (0, 4):
        print("     wave", w, np.round(tr[wg, w, 10, :7] - tr[wg, 0, 10, 0], 2).tolist())
# how far apart are the arrivals (stamp 5) of the 32 column-tile workgroups of one row tile?
arr = tr[:, :, steps, 5].reshape(H // 16, B // 256, W, -1)     # block b: g = b % groups, ct = b // groups
spread = arr.max(0) - arr.min(0)
print(f"arrival spread across the 32 producers of a row tile: mean {spread.mean():.2f} us, p90 {np.percentile(spread, 90):.2f}")
