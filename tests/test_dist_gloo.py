"""The N>1 path on CPU: world_size-2 gloo processes (one process per rank, rendezvous on 127.0.0.1).

  * cpg.dist plumbing: SUM all-reduce, variable-length all-gather of accepted CLaSS rows, parameter broadcast;
  * the data-parallel FORMULATION the HIP path uses (losses.set_distributed + FusedAdamClip(reduce_fn)): with the global
    non-PAD count and the global random-feature sums exchanged, SUM-all-reduce / world of the per-rank gradients equals
    the single-device gradient of the full batch.  Checked here with the numpy oracle standing in for the kernels.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, out):
    import sys
    for p in (ROOT, os.path.join(ROOT, "controlled-peptide-generation_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cpg import dist as cdist
    w, r, _ = cdist.init(backend="gloo")
    assert (w, r) == (world, rank)
    try:
        out[rank] = fn(rank, world)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def _plumbing(rank, world):
    from cpg import dist as cdist
    t = torch.full((3,), float(rank + 1))
    cdist.allreduce_sum(t)
    rows = torch.arange((rank + 1) * 2 * 3, dtype=torch.float32).reshape((rank + 1) * 2, 3) + 100 * rank
    g = cdist.allgather_rows(rows)
    p = torch.nn.Parameter(torch.full((4,), float(rank)))
    cdist.broadcast_params([p])
    cdist.barrier()
    return t.tolist(), g.tolist(), p.data.tolist()


def test_gloo_plumbing():
    res = _run(_plumbing)
    for t, g, p in res:
        assert t == [3.0, 3.0, 3.0]
        assert len(g) == 2 + 4 and g[0] == [0.0, 1.0, 2.0] and g[2] == [100.0, 101.0, 102.0]
        assert p == [0.0] * 4


def test_dp_formulation_equals_single_device():
    """Equal shards: recon (global count) + beta*mmd_rf (global feature means) are exactly shard-decomposable."""
    from conftest import load_golden, weights_of
    from oracle import wae
    g = load_golden("model_micro")
    P = weights_of(g)
    # single device on the first 6 rows vs two ranks with 3 rows each
    rnd6 = {k: g[k][:6] for k in ("eps", "c", "wd_mask", "out_mask", "z_prior_full", "z_prior_rf")}
    rnd6["rf_w"], rnd6["rf_b"] = g["rf_w"], g["rf_b"]
    terms, G, _ = wae.train_loss_and_grads(P, g["ids"][:6], rnd6, 1.25, 0.0, 0.0, "mmdrf")
    ref = np.concatenate([G[k].reshape(-1) for k in sorted(G)])
    res = _run(_dp_equal)
    for flat, mmdrf, _ in res:
        assert abs(mmdrf - float(terms["mmdrf"])) < 1e-6          # every rank sees the GLOBAL mmd value
        np.testing.assert_allclose(flat, ref, atol=2e-6, rtol=1e-4)


def _dp_equal(rank, world):
    from conftest import load_golden, weights_of
    from oracle import wae
    g = load_golden("model_micro")
    P = weights_of(g)
    lo, hi = rank * 3, rank * 3 + 3
    rnd = {k: g[k][lo:hi] for k in ("eps", "c", "wd_mask", "out_mask", "z_prior_full", "z_prior_rf")}
    rnd["rf_w"], rnd["rf_b"] = g["rf_w"], g["rf_b"]

    def allreduce(a):
        t = torch.from_numpy(np.asarray(a, np.float64).copy())
        dist.all_reduce(t)
        return t.numpy()
    terms, G, _ = wae.train_loss_and_grads(P, g["ids"][lo:hi], rnd, 1.25, 0.0, 0.0, "mmdrf", dp=(allreduce, world))
    t = torch.from_numpy(np.concatenate([G[k].reshape(-1) for k in sorted(G)]).astype(np.float64))
    dist.all_reduce(t)
    return (t.numpy() / world), float(terms["mmdrf"]), float(terms["recon"])


# ---------------------------------------------------------------------------------------------- CLaSS rounds, sharded
class _FakeQ:
    rng = 'device'


def _fake_round_factory():
    """Stands in for the device part of a sampling round (draw + score + decode: GPU kernels, covered by tests/test_gpu_pipeline.py)
    with a deterministic function of (round, global row): the test is about the sharding / gather / de-duplication / stop-rule
    logic of sample_pipeline.run_rounds, which must give the same table for 1 and 2 ranks."""
    state = {"round": 0}

    def fake(model, dataset, Q, n, sample_mode='beam', decode_accepted_only=False, shard=(0, 1)):
        import numpy as np
        rank, world = shard
        rnd = state["round"]
        state["round"] += 1
        n_local = n // world
        rows = np.arange(rank * n_local, (rank + 1) * n_local)
        rs = np.random.RandomState(1000 + rnd)
        table = rs.randint(4, 10, size=(n, 6))            # few distinct residue rows -> plenty of duplicates
        acc_all = rs.rand(n) < 0.3
        zs = rs.randn(n, 5).astype(np.float32)
        ids = np.full((n_local, 26), -1, np.int16)
        ids[:, 0] = 2
        ids[:, 1:7] = table[rows]
        ids[:, 7] = 3
        import torch
        import sample_pipeline as sp
        letters, n_res = sp.residue_rows(torch.from_numpy(ids), dataset.n_vocab)
        frame = {'letters': letters, 'n_res': n_res, 'z': torch.from_numpy(zs[rows]), 'accept_z': torch.from_numpy(acc_all[rows]),
                 'clfZ_prob_accum': torch.from_numpy(rs.rand(n)[rows])}
        return frame, dict(proposed=n_local, decoded=n_local, decoder_evals=25 * n_local)
    return fake


def _class_rounds(rank, world):
    import logging
    import sample_pipeline as sp
    from cpg.synth import SyntheticPeptideLoader
    logging.getLogger('GenerationAPI').setLevel(logging.WARNING)
    sp.sample_round_arrays = _fake_round_factory()
    ds = SyntheticPeptideLoader(4, 25, 'cpu', size=8)
    df, st = sp.run_rounds(None, ds, _FakeQ(), 64, 40, max_rounds=30, return_stats=True)
    return list(df['peptide']), [bool(a) for a in df['accept']], [z.tolist() for z in df['z']], st


def test_class_rounds_sharded_equal_single_rank():
    """Reference main loop sample_pipeline.py:299-322 under sharding: the union of the ranks' rows, gathered, de-duplicated
    within the round and against earlier rounds, with the stop rule on the gathered set, is the single-rank table."""
    two = _run(_class_rounds, world=2)
    one = _run(_class_rounds, world=1)[0]
    assert two[0][:3] == two[1][:3]            # identical on every rank
    assert two[0][:3] == one[:3]
    pep, acc, _, st = one
    assert len(set(pep)) == len(pep) and sum(acc) >= 40
    assert st['rounds'] == two[0][3]['rounds'] and st['kept'] == len(pep)


def test_bench_self_spawns_ranks_without_world_size():
    """`python bench.py --gpus 2` started WITHOUT torch.distributed.run (the form the driver uses for N=1) re-executes itself as
    two ranks; --dist-selftest keeps the run on the CPU (gloo) and returns the `rccl` object of the collectives probe."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-selftest"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["ranks"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["rccl"]["allreduce_ms"] > 0 and line["rccl"]["allgather_ms"] > 0
