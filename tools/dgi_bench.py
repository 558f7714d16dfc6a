"""Times cpg_gru_dgi_reduce at config-B size (T=25, B=2048, H=512, V=24) with and without the over-time sums."""
import os, sys, torch
sys.path.insert(0, "controlled-peptide-generation_amd")
from cpg import ops
T, B, H, V = 25, 2048, 512, 24
d = torch.device("cuda")
torch.manual_seed(0)
dG = torch.randn(T, B, 4 * H, device=d)
tok = torch.randint(0, V, (T, B), device=d, dtype=torch.int32)
dtab, dsum, drowc = torch.empty(V, 3 * H, device=d), torch.empty(4 * H, device=d), torch.empty(B, 3 * H, device=d)
ws = ops.workspace(ops.query("cpg_gru_wgrad_workspace", T, B, H, V), d)
for rowc in (False, True):
    def run():
        ops.call("cpg_gru_dgi_reduce", T, B, H, ops._p(dG), ops._p(tok), V, ops._p(dtab), ops._p(dsum), ops._p(drowc) if rowc else None, 0,
                 ops._p(ws), ws.numel(), 0, ops._stream())
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("dgi_reduce %s: %.1f us  (%.2f TB/s of dG)  knobs %s" % ("+drowc" if rowc else "      ", us, dG.numel() * 4 / us / 1e6,
          {k: v for k, v in os.environ.items() if k.startswith("CPG_DGI")}))
