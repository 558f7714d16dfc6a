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
    return list(df['peptide']), [bool(a) for a in df['accept']], [np.asarray(z).tolist() for z in df['z']], st


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


# ---------------------------------------------------------------------------------------------- world 8 (the run nobody can make here)
def _class_rounds_big(rank, world):
    """One 1 M-proposal round (BASELINE.json configs[3]: 125 000 rows per rank at world 8) + a second round forced by the stop rule."""
    import logging
    import sample_pipeline as sp
    from cpg.synth import SyntheticPeptideLoader
    logging.getLogger('GenerationAPI').setLevel(logging.WARNING)
    state = {"round": 0}

    def fake(model, dataset, Q, n, sample_mode='beam', decode_accepted_only=False, shard=(0, 1)):
        r, w = shard
        rnd = state["round"]
        state["round"] += 1
        n_local = n // w
        rows = np.arange(r * n_local, (r + 1) * n_local)
        rs = np.random.RandomState(2000 + rnd)            # the ROUND's stream; a rank keeps its rows of it
        table = rs.randint(4, 24, size=(n, 5))            # 20^5 = 3.2 M residue rows: duplicates within and across rounds
        acc_all = rs.rand(n) < 0.05
        prob = rs.rand(n)
        ids = np.full((n_local, 26), -1, np.int16)
        ids[:, 0] = 2
        ids[:, 1:6] = table[rows]
        ids[:, 6] = 3
        letters, n_res = sp.residue_rows(torch.from_numpy(ids), dataset.n_vocab)
        z = torch.from_numpy(rows.astype(np.float32)).unsqueeze(1) + 1e7 * rnd      # z = global row id: rows are traceable
        frame = {'letters': letters, 'n_res': n_res, 'z': z, 'accept_z': torch.from_numpy(acc_all[rows]),
                 'clfZ_prob_accum': torch.from_numpy(prob[rows])}
        return frame, dict(proposed=n_local, decoded=n_local, decoder_evals=25 * n_local)
    sp.sample_round_arrays = fake
    ds = SyntheticPeptideLoader(4, 25, 'cpu', size=8)
    df, st = sp.run_rounds(None, ds, _FakeQ(), 1000000 // (4 * world) * (4 * world), 60000, max_rounds=4, return_stats=True)
    import hashlib
    h = hashlib.sha256()
    h.update("".join(df['peptide']).encode())
    h.update(np.asarray(df['accept'], bool).tobytes())
    h.update(np.concatenate([np.asarray(z, np.float32).reshape(-1) for z in df['z']]).tobytes())
    return h.hexdigest(), len(df), int(df['accept'].sum()), st


@pytest.mark.timeout(600)
def test_class_rounds_world8_equal_single_rank():
    """BASELINE.json configs[3] sharding at the world size of the node the driver benches on: 8 gloo ranks x 125 000 rows per
    round.  The gathered, de-duplicated table (peptides, accept flags, z rows - hashed) and the stop rule (two rounds needed for
    60 000 accepted) are identical on all eight ranks and equal to the single-rank table of the same streams."""
    eight = _run(_class_rounds_big, world=8)
    one = _run(_class_rounds_big, world=1)[0]
    assert len({r[0] for r in eight}) == 1, "ranks disagree on the gathered table"
    assert eight[0][:3] == one[:3]
    assert one[3]['rounds'] == eight[0][3]['rounds'] >= 2 and one[2] >= 60000
    assert eight[0][3]['proposed'] == one[3]['proposed'] == 1000000 * one[3]['rounds']


def _bucket_overlap(rank, world):
    """The optimiser's bucket scheduler (cpg.optim.BucketReducer: pure host logic) at the bench model's real layout: 19.6 MB of
    config-B gradients, 'decoder' and 'encoder_heads' buckets all-reduced asynchronously from their boundaries - one of them fired
    TWICE, as a module used twice in the graph does - the tail reduced in finish().  CPU tensors, gloo collectives."""
    from cpg.optim import BucketReducer, bucket_layout
    from cpg import dist as cdist
    pad = lambda k: -(-k // 16) * 16
    H, Z, E, V = 512, 510, 150, 24
    groups = [(None, [V * E]),                                                      # shared embedding (listed twice, F6)
              (None, [3 * H * E, 3 * H * H, 3 * H, 3 * H] * 2),                     # encoder recurrence, two directions
              ('decoder', [3 * H * (E + H), 3 * H * H, 3 * H, 3 * H, V * H, V]),
              ('encoder_heads', [Z * 2 * H, Z, Z * 2 * H, Z])]
    sizes = [k for _, ks in groups for k in ks]
    tags = [t for t, ks in groups for _ in ks]
    offs, rng, tail_end, total = bucket_layout(sizes, tags, pad)
    assert set(rng) == {'decoder', 'encoder_heads'} and 0 < tail_end == rng['decoder'][0] and rng['encoder_heads'][1] == total
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(total, generator=g)
    mine = flat.clone()
    red = BucketReducer(flat, rng, tail_end, reduce_fn=cdist.allreduce_sum, async_reduce_fn=cdist.allreduce_sum_async, world=world)
    assert red.overlapped
    red.on_boundary('decoder')
    red.on_boundary('decoder')            # a second firing of a started bucket must not reduce it again
    red.on_boundary('encoder_heads')
    assert red.reduced == {'decoder', 'encoder_heads'} and len(red.inflight) == 2
    red.finish()
    assert not red.inflight and not red.reduced
    # expected: SUM over ranks of every rank's buffer
    want = torch.zeros(total)
    for r in range(world):
        want += torch.randn(total, generator=torch.Generator().manual_seed(100 + r))
    err = float((flat - want).abs().max())
    # plain path (no boundaries fired): one all-reduce of the whole buffer
    red2 = BucketReducer(mine, rng, tail_end, reduce_fn=cdist.allreduce_sum, async_reduce_fn=None, world=world)
    assert not red2.overlapped
    red2.finish()
    return err, float((mine - want).abs().max()), total


@pytest.mark.parametrize("world", [2, 8])
def test_bucketed_gradient_reduction_world_n(world):
    for err, err_plain, total in _run(_bucket_overlap, world=world):
        assert total > 4_900_000        # the config-B model (4 907 556 unique parameters + padding)
        assert err < 1e-4 and err_plain < 1e-4


class _GlooCommApi:
    """Stand-in for the library's RCCL entry points (cpg.dist._CApi) with the SAME contract on CPU tensors over the gloo group:
    the unique id is an opaque 128-byte token every rank must present identically, cpg_allgatherv takes the byte count of EVERY
    rank - identical tables on all ranks or the collective is broken (here: asserted) - and moves raw bytes."""

    def unique_id(self):
        return bytes(range(128))

    def init(self, uid, rank, world):
        assert uid == bytes(range(128)), "the id broadcast from rank 0 did not arrive intact"
        return ("comm", rank, world)

    def allreduce_f32(self, comm, t):
        dist.all_reduce(t)

    def allgatherv(self, comm, send, counts, rank, world, out):
        tables = [None] * world
        dist.all_gather_object(tables, [int(c) for c in counts])
        assert all(tb == tables[0] for tb in tables), f"ranks disagree on the byte-count table: {tables}"
        flat = out.view(-1).view(torch.uint8) if out.numel() else out.reshape(0).view(torch.uint8)
        assert flat.numel() == sum(counts)
        off = 0
        for r in range(world):
            if counts[r]:
                piece = flat[off:off + counts[r]]
                if r == rank:
                    piece.copy_(send.contiguous().view(-1).view(torch.uint8))
                dist.broadcast(piece, r)          # in place: `piece` is a view of `out`
            off += counts[r]

    def destroy(self, comm):
        pass

    def record(self):
        return None


def _libcomm_logic(rank, world):
    from cpg import dist as cdist
    c = cdist.LibComm(rank, world, api=_GlooCommApi())
    t = torch.full((5,), float(rank + 1))
    c.allreduce_sum(t)
    w = c.allreduce_sum_async(torch.ones(3))
    w.wait()
    # rank 1 contributes NO rows (an empty CLaSS shard) - [0, 7] - the others rank+1 rows
    n = 0 if rank == 1 else rank + 1
    rows = (torch.arange(n * 7, dtype=torch.float32).reshape(n, 7) + 1000 * rank)
    out = c.allgather_rows(rows)
    ids = c.allgather_rows(torch.full((2 if rank == 0 else 0, 26), rank, dtype=torch.int16))   # only rank 0 has rows
    none = c.allgather_rows(torch.zeros(0, 3))                                                   # nobody has rows
    c.close()
    return t.tolist(), out.tolist(), list(ids.shape), list(none.shape)


@pytest.mark.parametrize("world", [2, 4])
def test_libcomm_id_exchange_and_row_counts_with_empty_ranks(world):
    """cpg.dist.LibComm (CPG_COMM=lib) at world > 1 - RCCL refuses two ranks on the 1-GPU box, so its HOST logic runs here over a
    stand-in with the C entry points' contract: id broadcast, count exchange, and the byte table of cpg_allgatherv when a rank has
    zero rows (round-3 advisor finding: the row size used to come from the local rows and was 0 on an empty rank)."""
    res = _run(_libcomm_logic, world=world)
    want_rows = []
    for r in range(world):
        n = 0 if r == 1 else r + 1
        want_rows += (np.arange(n * 7, dtype=np.float32).reshape(n, 7) + 1000 * r).tolist()
    for t, out, ids_shape, none_shape in res:
        assert t == [float(sum(range(1, world + 1)))] * 5
        assert out == want_rows
        assert ids_shape == [2, 26] and none_shape == [0, 3]


def test_gradient_boundary_fires_after_last_use_in_the_graph():
    """cpg.ops.grad_boundary when a module runs twice in the differentiated graph: the bucket callback must fire once, after the
    LAST of the tag's boundaries IN THAT GRAPH has been reached (round-3 advisor finding: it used to fire at the first).  The count is
    taken on the graph being differentiated (round-4 advisor finding: a process-global count of forward passes made the decision depend
    on host history): a grad-enabled forward that is never differentiated changes NOTHING."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "controlled-peptide-generation_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cpg import ops
    w = torch.nn.Parameter(torch.ones(4))
    x = torch.ones(4, requires_grad=True)
    events = []

    class Probe(torch.autograd.Function):    # records when the gradient of one use of `w` is produced
        @staticmethod
        def forward(ctx, t, name):
            ctx.name = name
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            events.append(ctx.name)
            return g, None

    def use(inp, name):                       # a "decoder" applied to inp: boundary on its input, parameter use behind it
        inp = ops.grad_boundary('decoder', inp)
        return (Probe.apply(inp, name) * w).sum()
    loss = use(x, 'first') + use(x * 2.0, 'second')
    assert ops._count_boundaries(loss) == {'decoder': 2}
    with ops.backward_scope(lambda tag: events.append('FIRE:' + tag), root=loss):
        loss.backward()
    assert events.count('FIRE:decoder') == 1
    assert events.index('FIRE:decoder') > max(events.index('first'), events.index('second')), events
    # a grad-enabled forward that is never differentiated (a validation / logging forward without no_grad on ONE rank) must not
    # change when - or whether - the next backward pass fires the bucket
    _ = use(x, 'dangling')
    events.clear()
    loss = use(x, 'only')
    with ops.backward_scope(lambda tag: events.append('FIRE:' + tag), root=loss):
        loss.backward()
    assert events == ['only', 'FIRE:decoder']
    with pytest.raises(ValueError):
        ops.backward_scope(lambda tag: None)           # a callback without the graph it applies to is refused


def _asym_worker(rank, world, port, q):
    """Rank 0 runs an EXTRA grad-enabled forward that it never differentiates; both ranks then run the same training backward with
    bucketed all-reduces.  The sequence of collectives each rank issues (sizes, in order) must be identical."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "controlled-peptide-generation_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpg import ops
    from cpg.optim import BucketReducer
    torch.manual_seed(0)
    w_dec, w_head, w_enc = (torch.nn.Parameter(torch.randn(8)) for _ in range(3))
    flat = torch.zeros(24)
    issued = []

    def areduce(t):
        issued.append(int(t.numel()))
        return dist.all_reduce(t, async_op=True)
    red = BucketReducer(flat, {'encoder_heads': (8, 16), 'decoder': (16, 24)}, 8, None, areduce, world)

    def model(x):
        h = (x * w_enc).tanh()
        h = ops.grad_boundary('encoder_heads', h)
        zc = h * w_head
        zc = ops.grad_boundary('decoder', zc)
        return (zc * w_dec).sum()
    x = torch.full((8,), float(rank + 1), requires_grad=True)
    if rank == 0:
        _ = model(x)                                   # validation-style forward WITH grad enabled, never differentiated
    for _step in range(2):
        loss = model(x)
        for p_ in (w_dec, w_head, w_enc):
            p_.grad = None
        with ops.backward_scope(red.on_boundary, root=loss):
            loss.backward()
        flat[0:8], flat[8:16], flat[16:24] = w_enc.grad, w_head.grad, w_dec.grad
        red.finish()
    q.put((rank, issued, flat.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_asymmetric_forward_keeps_the_collective_sequence():
    """Round-4 advisor finding (medium): with a process-global boundary count, a rank that ran one extra grad-enabled forward skipped
    its boundary all-reduces and reduced those buckets synchronously in finish() - a different collective sequence from its peers
    (hang / silent corruption on RCCL).  Two gloo ranks, rank 0 with the extra forward: both must issue the same sequence."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 17
    procs = [ctx.Process(target=_asym_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert res[0][1] == res[1][1], (res[0][1], res[1][1])
    assert res[0][1][:2] == [8, 8], res[0][1]          # both buckets were started from their boundaries, on both ranks
