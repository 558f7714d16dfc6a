"""Data-parallel HIP path, two ranks sharing the one GPU of the test box (gloo carries the collectives; on a real node
the same code runs one rank per GPU over RCCL): all-reduced gradients of two half batches == gradients of the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads(rank, world, rows, out, bucketed=False, regu='mmdrf'):
    import sys
    for p in (ROOT, os.path.join(ROOT, "controlled-peptide-generation_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from conftest import load_golden, weights_of
    from helpers import build_model, cu
    import cfg
    import losses
    from cpg import dist as cdist
    from cpg.optim import FusedAdamClip
    g = load_golden("model_micro")
    m = build_model(weights_of(g))
    losses.rf.clear()
    losses.rf['gaussian'] = (cu(g["rf_w"]), cu(g["rf_b"]))
    cfg.losses.wae_mmd.sigma = 7.0
    reduce_fn = None
    if world > 1:
        reduce_fn = cdist.allreduce_sum
        losses.set_distributed(reduce_fn, world, gather_fn=cdist.allgather_equal, rank=rank)
    if bucketed:   # the trainer's form: gradient buckets all-reduced from the boundaries inside backward()
        import train_vae as tv
        opt = tv.make_optimizer(cfg.Bunch(lr=1e-3, clip_grad=5.0), m, reduce_fn, world)
        assert set(opt.bucket_range) == {'decoder', 'encoder_heads'} and 0 < opt.tail_end < opt.flat_g.numel()
    else:
        opt = FusedAdamClip(m.vae_params(), lr=1e-3, max_norm=5.0, reduce_fn=reduce_fn, world=world)
    lo, hi = rows
    ids = cu(g["ids"][lo:hi])
    rnd = dict(eps=cu(g["eps"][lo:hi]), c=cu(g["c"][lo:hi]), wd_mask=cu(g["wd_mask"][lo:hi]), out_mask=cu(g["out_mask"][lo:hi]))
    (mu, lv), (z, c), logits = m(ids, rnd=rnd)
    if regu == 'mmd':   # the full-kernel MMD as the regulariser: evaluated on the all-gathered global batch
        reg = losses.wae_mmd_gaussianprior(z, method='full_kernel', z_prior=cu(g["z_prior_full"][lo:hi]), global_batch=True)
    else:
        reg = losses.wae_mmd_gaussianprior(z, method='rf', z_prior=cu(g["z_prior_rf"][lo:hi]))
    loss = losses.recon_dec(ids, logits) + 1.25 * reg + 1e-3 * losses.kl_gaussian_sharedmu(mu, lv)
    opt.zero_grad()
    from cpg import ops
    if bucketed:
        opt.backward(loss)
        if world > 1:
            assert opt._reduced == {'decoder', 'encoder_heads'}, opt._reduced     # both boundaries fired inside backward()
        opt._finish_reduce()
    else:
        loss.backward()
        ops.join_deferred()
        if reduce_fn is not None:
            reduce_fn(opt.flat_g)
    # report in parameter-name order: the flat layout differs between the bucketed and the plain optimiser
    names = dict(m.named_parameters())
    flat = torch.cat([names[k].grad.reshape(-1) for k in sorted(names) if names[k].grad is not None and not k.startswith('classifier')])
    out[rank] = ((flat / world).cpu().numpy(), float(loss.item()))


def _worker(rank, world, port, out, bucketed=False, regu='mmdrf'):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CPG_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
    from cpg import dist as cdist
    cdist.init()
    _grads(rank, world, (0, 3) if rank == 0 else (3, 6), out, bucketed, regu)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("bucketed,regu", [(False, 'mmdrf'), (True, 'mmdrf'), (True, 'mmd')])
def test_two_rank_gradients_equal_single_rank(bucketed, regu):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, bucketed, regu), nprocs=2, join=True)
    single = mgr.dict()
    p = mp.get_context("spawn").Process(target=_grads, args=(0, 1, (0, 6), single, False, regu))
    p.start()
    p.join()
    assert p.exitcode == 0
    ref, ref_loss = single[0]
    for r in (0, 1):
        got, _ = out[r]
        np.testing.assert_allclose(got, ref, atol=2e-6 + 2e-4 * np.abs(ref).max())
