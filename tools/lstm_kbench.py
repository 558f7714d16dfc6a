import os, sys, torch
sys.path.insert(0, "controlled-peptide-generation_amd")
from cpg import ops
B, H, T, V = 2048, 512, 25, 24
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
w = (torch.randn(4*H, H, generator=g) / H**0.5).to(dev); b = torch.zeros(4*H, device=dev)
tab = (torch.randn(V, 4*H, generator=g)*0.3).to(dev); rowc = (torch.randn(B, 4*H, generator=g)*0.3).to(dev)
tok = torch.randint(0, V, (T, B), generator=g).to(torch.int32).to(dev)
hs = torch.zeros(T+1, B, H, device=dev); cs = torch.zeros(T+1, B, H, device=dev); gates = torch.empty(T, 4, B, H, device=dev)
def run(): ops.lstm_seq_fwd_persistent(T, B, H, False, w, b, tok, tab, rowc, None, hs, cs, gates)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("lstm fwd-persistent %.1f us/step" % (e0.elapsed_time(e1) / 10 / T * 1e3))
