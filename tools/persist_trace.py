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

B, H, T, V = 2048, 512, 25, 24
W = 16 if os.environ.get("KB_DTYPE", "f32") == "bf16" else 8    # waves per workgroup (csrc/gru_persist.hip: waves_of)
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
gates = torch.empty(T, 4, B, H, device=dev, dtype=ops.gates_dtype(B, H, False))
for _ in range(3):
    ops.gru_seq_fwd_persistent(T, B, H, False, w_hh, b_hh, tok, tab, rowc, None, hs, gates)
torch.cuda.synchronize()
ent = next(v for k, v in ops._persist_scratch.items() if k[0] == "gru")
nwg = (B // 256) * (H // 16)
n = nwg * W * T * 8
area = (B // 256) * (H // 8) * 16 * T * 8 * 8                # bytes of the trace area (sized for 8-unit column tiles, 16 waves)
tr = ent[0][-area:][:n * 8].view(torch.int64).cpu().numpy().reshape(nwg, W, T, 8).astype(np.float64) * 0.01   # us
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
# is the lateness of a producer systematic (same workgroup late at every step) or noise?
late = arr - arr.mean(0, keepdims=True)                       # [ct, g, wave, step]
sysm = late.mean(3)                                           # per producer, over steps
print(f"lateness of a producer vs its row tile's mean arrival: systematic part (mean over steps) std {sysm.std():.2f} us, "
      f"range {sysm.min():.2f} .. {sysm.max():.2f}; residual (step-to-step) std {(late - sysm[..., None]).std():.2f} us")
print("systematic lateness by column tile (mean over row groups / waves):", np.round(sysm.mean((1, 2)), 2).tolist())
print("systematic lateness by wave index (mean over column tiles / row groups):", np.round(sysm.mean((0, 1)), 2).tolist())
print("systematic lateness by row group:", np.round(sysm.mean((0, 2)), 2).tolist())
last = arr.argmax(0)                                          # which ct is last, [g, wave, step]
print("how often each column tile is the LAST arriver (of %d tile-steps):" % last.size, np.bincount(last.ravel(), minlength=H // 16).tolist())
# step-top skew: how far apart do the 32 consumers of a row tile START a step (stamp 0)?
top = tr[:, :, steps, 0].reshape(H // 16, B // 256, W, -1)
print(f"step-top spread across the 32 workgroups of a row tile: mean {(top.max(0) - top.min(0)).mean():.2f} us")
# wait-done (stamp 1) minus the last arrival of the previous step's producers = visibility latency of the counter
prev_last = arr.max(0)[:, :, :-1]                              # last arrival for step p   [g, wave, step-1]
seen = tr[:, :, steps, 1].reshape(H // 16, B // 256, W, -1)[:, :, :, 1:]   # wait done at step p+1 per consumer
vis = seen - prev_last[None]
print(f"wait-done minus last arrival (counter visibility + poll granularity): mean {vis.mean():.2f} us, p10 {np.percentile(vis, 10):.2f}, p90 {np.percentile(vis, 90):.2f}")
waiting = top[:, :, :, 1:] < prev_last[None]                   # consumer was already polling when the last producer arrived
print(f"consumers already waiting at the last arrival: {waiting.mean() * 100:.0f} %; their delay from last arrival to wait-done: "
      f"mean {vis[waiting].mean():.2f} us, p10 {np.percentile(vis[waiting], 10):.2f}, p50 {np.percentile(vis[waiting], 50):.2f}, p90 {np.percentile(vis[waiting], 90):.2f}")
dd = d.reshape(H // 16, B // 256, W, -1, 6)
slow = np.zeros(H // 16, bool)
slow[[6, 7, 22, 23]] = True
for i, nm in enumerate(names):
    print(f"  {nm:16s} slow tiles {dd[slow][..., i].mean():6.2f}   other tiles {dd[~slow][..., i].mean():6.2f}")
pp = np.diff(tr[:, :, steps, 0], axis=2).reshape(H // 16, B // 256, W, -1)
print("step period by column tile:", np.round(pp.mean((1, 2, 3)), 2).tolist())
# busy time (step top of wait-done .. arrival) by column tile: what a tile needs when it does not wait
busy = (tr[:, :, steps, 5] - tr[:, :, steps, 1]).reshape(H // 16, B // 256, W, -1)
print("wait-done -> arrival (own work) by column tile:", np.round(busy.mean((1, 2, 3)), 2).tolist())
print("own work by row group (XCD):", np.round(busy.mean((0, 2, 3)), 2).tolist())
print("own work by wave:", np.round(busy.mean((0, 1, 3)), 2).tolist())
pw = np.diff(tr[:, :, :, 0], axis=2)                           # all steps
print("step period by wave index (all steps):", np.round(pw.mean((0, 2)), 2).tolist())
end = tr[:, :, T - 1, 6] - t0
print("chain end (last step's stores issued) by wave index, us from the first stamp: mean", np.round(end.mean(0), 1).tolist(), " max", np.round(end.max(0), 1).tolist())
print("step period by wave over steps 0-7 / 8-15 / 16-23:", [np.round(pw[:, :, a:a + 8].mean((0, 2)), 2).tolist() for a in (0, 8, 16)])
