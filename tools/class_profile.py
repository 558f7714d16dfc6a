"""CLaSS decode loops only (config A), for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
sys.path.insert(0, ROOT)
from bench import model_kwargs
from cpg import ops, decode as cdecode
from models.model import RNN_VAE

dev = torch.device("cuda:0")
torch.manual_seed(1238)
m = RNN_VAE(n_vocab=24, max_seq_len=25, **model_kwargs(100, 80)).to(dev)
m.device = dev
N = int(os.environ.get("N", 262144))
mode = os.environ.get("MODE", "greedy")
z = ops.rng_normal((N, 100), 99, 0, dev)
c = torch.zeros(N, 2, device=dev); c[:, 1] = 1
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "greedy":
        m.generate_sentences(N, z, c, sample_mode="greedy")
    else:
        cdecode.decode_beam_arrays(m.decoder, z, c, 25)
    torch.cuda.synchronize()
    print(mode, N, f"{1e3*(time.perf_counter()-t0):.2f} ms")
