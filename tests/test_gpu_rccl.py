"""The RCCL code path is loaded and exercised at least once on the GPU box: torch.distributed backend "nccl" (= RCCL on
ROCm) with world_size 1 - process-group creation, the flat-gradient all-reduce, the statistic exchanges, the variable-length
row gather and the barrier that the N > 1 runs use (the driver owns the 8-GPU runs; multi-rank logic is covered on gloo in
tests/test_dist_gloo.py and tests/test_gpu_dp.py)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path[:0] = [os.environ["CPG_ROOT"], os.path.join(os.environ["CPG_ROOT"], "controlled-peptide-generation_amd")]
import torch, torch.distributed as dist
from cpg import dist as cdist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)       # world 1 is skipped by cdist.init(): force the backend up
assert dist.get_backend() == "nccl"
t = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
dist.all_reduce(t)                                                    # RCCL kernel launch (identity at world 1)
assert torch.equal(t, torch.arange(1 << 20, device="cuda", dtype=torch.float32))
rows = torch.randn(37, 5, device="cuda")
n = torch.tensor([rows.shape[0]], device="cuda")
outs = [torch.zeros_like(n)]
dist.all_gather(outs, n)
assert int(outs[0].item()) == 37
p = torch.nn.Parameter(torch.ones(4, device="cuda"))
dist.broadcast(p.data, 0)
dist.barrier()
torch.cuda.synchronize()
# the training step's data-parallel plumbing with a live nccl group (world 1: reduce_fn is the real all-reduce)
import cfg, losses, train_vae as tv
from bench import model_kwargs
from cpg.synth import synth_ids
from models.model import RNN_VAE
torch.manual_seed(0)
m = RNN_VAE(n_vocab=24, max_seq_len=25, **model_kwargs(62, 64)).cuda()
m.device = torch.device("cuda")
m.use_device_rng(3)
losses.rf.clear()
losses.set_distributed(lambda x: dist.all_reduce(x), 1)
cfgv = cfg.Bunch(lr=1e-3, clip_grad=5.0, z_regu_loss='mmdrf', lambda_logvar_L1=0.0, lambda_logvar_KL=1e-3,
                 beta=cfg.Bunch(start=cfg.Bunch(val=1.0, iter=0), end=cfg.Bunch(val=2.0, iter=100)))
tr = tv.make_optimizer(cfgv, m, lambda x: dist.all_reduce(x), 1)
ids = synth_ids(128, 25, 24, torch.Generator().manual_seed(1)).cuda()
out = tv.train_step(cfgv, m, tr, ids, 0)
assert torch.isfinite(out["L_vae"]).item()
# the library's own communicator (cpg_comm_*, cpg_allreduce_f32, cpg_allgatherv) at world 1: librccl bound at run time (the copy
# torch has loaded), id exchange, collectives as plain stream launches
from cpg import ops
assert ops.query("cpg_comm_available") == 1
c = cdist.LibComm(0, 1)
g = torch.arange(1 << 18, device="cuda", dtype=torch.float32)
c.allreduce_sum(g)
rows = torch.randn(37, 5, device="cuda")
got = c.allgather_rows(rows)
work = c.allreduce_sum_async(g)
work.wait()
torch.cuda.synchronize()
assert torch.equal(g, torch.arange(1 << 18, device="cuda", dtype=torch.float32)) and torch.equal(got, rows)
assert c.allgather_rows(rows[:0]).shape == (0, 5)
c.close()
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_rccl_world1_smoke():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X: no CUDA/HIP device visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               CPG_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
