"""Two ranks (gloo rendezvous, both on cuda:0 -- RCCL itself refuses two ranks on one device, the
driver's multi-GPU runs use it) through the row-block sharded full-graph SpMM of
stochastic_gcn_amd/parallel.py: nnz-balanced vertex ranges, operand-resident and operand-all-gathered
forward and backward, against SciPy on the whole matrix."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parallel_gloo as tg        # noqa: E402  (spawn helper)

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _worker(rank, world, port, kernel, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGCN_DIST_BACKEND="gloo")
    from stochastic_gcn_amd.parallel import DataParallel, ShardedSpMM
    from stochastic_gcn_amd import synthetic
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    par = DataParallel(backend="gloo", device=dev)
    try:
        n, d = 3000, 70
        a = synthetic.rmat_like(n, 30 * n, seed=5)          # same matrix on every rank
        rng = np.random.RandomState(1)
        B = rng.standard_normal((n, d)).astype(np.float32)
        dC = rng.standard_normal((n, d)).astype(np.float32)
        sh = ShardedSpMM(par, a, dev, kernel=kernel)
        pad = lambda x: torch.nn.functional.pad(torch.from_numpy(x), (0, 2)).to(dev)[:, :d]   # noqa: E731  pitch 72
        Bd, dCd = pad(B), pad(dC)
        c1 = sh.forward(Bd)
        c2 = sh.forward_allgather(Bd[sh.lo:sh.hi].contiguous())
        db = sh.backward(dCd)
        db2 = sh.backward_allgather(dCd[sh.lo:sh.hi].contiguous())
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), c1=c1.cpu().numpy(), c2=c2.cpu().numpy(),
                 db=db.cpu().numpy(), db2=db2.cpu().numpy(), lo=np.array([sh.lo]), hi=np.array([sh.hi]), nnz=np.array([sh.local_nnz]))
    finally:
        par.shutdown()


@pytest.mark.parametrize("kernel", ["cs", "rows"])
def test_sharded_spmm_two_ranks(tmp_path, kernel):
    import torch.multiprocessing as mp
    from stochastic_gcn_amd import synthetic
    from oracle import oracle_np as onp
    world, port = 2, tg._free_port()
    mp.spawn(_worker, args=(world, port, kernel, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "r%d.npz" % k)) for k in range(world)]
    n, d = 3000, 70
    a = synthetic.rmat_like(n, 30 * n, seed=5)
    rng = np.random.RandomState(1)
    B = rng.standard_normal((n, d)).astype(np.float32)
    dC = rng.standard_normal((n, d)).astype(np.float32)
    want_c, want_db = a.dot(B.astype(np.float64)), a.T.dot(dC.astype(np.float64))
    assert r[0]["lo"][0] == 0 and r[0]["hi"][0] == r[1]["lo"][0] and r[1]["hi"][0] == n
    for key, want in (("c1", want_c), ("c2", want_c), ("db", want_db), ("db2", want_db)):
        got = np.concatenate([r[0][key], r[1][key]], axis=0)      # rank order = vertex order
        assert got.shape == want.shape
        assert onp.rel_err(got, want) <= TOL, key


def test_sharded_spmm_blocks_tile_the_full_product(dev=None):
    """Four nnz-balanced row blocks of S-Reddit/10 (built one after the other on this GPU, no
    collectives: the operand is resident) reproduce the unsharded product, forward and backward."""
    import types
    from stochastic_gcn_amd import ops, synthetic
    from stochastic_gcn_amd.parallel import ShardedSpMM
    from oracle import oracle_np as onp
    dev = torch.device("cuda:0")
    n, _, full_adj, *_ = synthetic.reddit_like(n=23296, m=1160000, splits=(15241, 2369, 5533), seed=3,
                                               with_features=False)
    d = 96
    g = torch.Generator(device=dev); g.manual_seed(0)
    B = torch.randn((n, d), device=dev, generator=g)
    dC = torch.randn((n, d), device=dev, generator=g)
    whole = ops.DeviceCSR.from_scipy(full_adj, dev, with_transpose=True)
    want_c, want_db = ops.spmm(whole, B), ops.spmm(whole.transpose, dC)
    world, rows, nnz = 4, 0, []
    cs, dbs = [], []
    for r in range(world):
        sh = ShardedSpMM(types.SimpleNamespace(rank=r, world=world, active=False), full_adj, dev)
        assert sh.lo == rows
        rows = sh.hi
        nnz.append(sh.local_nnz)
        cs.append(sh.forward(B))
        dbs.append(sh.backward(dC))
    assert rows == n and sum(nnz) == full_adj.nnz
    assert max(nnz) - min(nnz) <= 2 * int(np.diff(full_adj.indptr).max())           # balanced by nonzeros
    assert onp.rel_err(torch.cat(cs).cpu().numpy(), want_c.cpu().numpy()) <= TOL
    assert onp.rel_err(torch.cat(dbs).cpu().numpy(), want_db.cpu().numpy()) <= TOL


def test_sharded_spmm_rmat_eight_blocks_d256():
    """BASELINE config 5 in miniature: an R-MAT graph (2^18 vertices, 5 M edges, the generator's skewed
    rows and columns), d = 256, eight nnz-balanced row blocks with the operand resident -- each block built
    with the lane-group count ShardedSpMM picks for the width and the block's density (two groups per wavefront at
    d = 256, one for the block of the heaviest rows); the
    blocks tile the unsharded product of the row-gather kernel, forward and backward."""
    import types
    from stochastic_gcn_amd import ops, synthetic
    from stochastic_gcn_amd.parallel import ShardedSpMM
    from oracle import oracle_np as onp
    dev = torch.device("cuda:0")
    n = 1 << 18
    adj = synthetic.rmat_like(n, 5_000_000, seed=2)
    d = 256
    g = torch.Generator(device=dev); g.manual_seed(1)
    B = torch.randn((n, d), device=dev, generator=g)
    dC = torch.randn((n, d), device=dev, generator=g)
    # the reference product is the CPU ORACLE's (oracle_np.spmm on the whole matrix), not another HIP kernel
    adj_t = adj.T.tocsr()
    Bh, dCh = B.cpu().numpy(), dC.cpu().numpy()
    want_c = onp.spmm(adj.indptr, adj.indices, adj.data, Bh)
    want_db = onp.spmm(adj_t.indptr, adj_t.indices, adj_t.data, dCh)
    rows, cs, dbs, groups = 0, [], [], set()
    for r in range(8):
        sh = ShardedSpMM(types.SimpleNamespace(rank=r, world=8, active=False), adj, dev, d=d)
        assert sh.lo == rows
        rows = sh.hi
        groups.add(sh.A.G)
        cs.append(sh.forward(B))
        dbs.append(sh.backward(dC))
    # (every block of this small graph fits the rounds of two-group tiles: two groups, or one where that saves passes)
    assert rows == n and 2 in groups and groups <= {1, 2}
    assert onp.rel_err(torch.cat(cs).cpu().numpy(), want_c) <= TOL
    assert onp.rel_err(torch.cat(dbs).cpu().numpy(), want_db) <= TOL


def _sampled_rows_vs_oracle(blk, rows, X, got, dev):
    """max-norm relative error of ``got[rows]`` against the oracle's product of the rows ``rows`` of the CSR block
    ``blk`` with the device operand X (only the operand rows those nonzeros reference leave the device)."""
    from oracle import oracle_np as onp
    sub = blk[rows].tocsr()
    cols = np.unique(sub.indices)
    remap = np.searchsorted(cols, sub.indices).astype(np.int32)
    Xh = X[torch.from_numpy(cols.astype(np.int64)).to(dev)].cpu().numpy()
    ref = onp.spmm(sub.indptr, remap, sub.data, Xh)
    return onp.rel_err(got[torch.from_numpy(rows.astype(np.int64)).to(dev)].cpu().numpy(), ref), ref


def test_rmat_10m_one_block_of_eight_full_size_vs_oracle():
    """BASELINE config 5 at its REAL size (SURVEY.md 8d S-RMAT: 10 M vertices, 200 M R-MAT edges, d = 256): one GPU's
    block of the 8-way load-balanced sharding -- the block an 8-GPU job's rank 3 owns, built by the same ShardedSpMM
    code path (emulated rank 3 of 8), the 10.24 GB dense operand resident -- forward C[lo:hi] = A[lo:hi,:] . B and
    backward dB[lo:hi] = A^T[lo:hi,:] . dC on the autotuned column sweep, against the CPU oracle on 1,540 sampled rows
    of the block including its heaviest, plus size-independent properties on ALL rows (A . 1 = row sums, linearity)."""
    import types
    from stochastic_gcn_amd import ops, synthetic
    from stochastic_gcn_amd.parallel import ShardedSpMM
    dev = torch.device("cuda:0")
    n, d = 10_000_000, 256
    adj = synthetic.cached_graph("rmat_10m_200m_seed1", lambda: synthetic.rmat_like(n, 200_000_000, seed=1))
    assert adj.shape == (n, n) and adj.nnz == 196_949_452
    sh = ShardedSpMM(types.SimpleNamespace(rank=3, world=8, active=False), adj, dev, d=d)
    adj_t = adj.T.tocsr()
    blk, blk_t = adj[sh.lo:sh.hi].tocsr(), adj_t[sh.lo:sh.hi].tocsr()
    # the blocks are balanced by LOAD: nonzeros of both directions + ShardedSpMM.ROW_WEIGHT per row and direction; a sparse
    # block with more rows than a round of two-group tiles takes four lane groups and the clock in work coordinates
    load, total = blk.nnz + blk_t.nnz + 2 * sh.row_weight * (sh.hi - sh.lo), 2 * adj.nnz + 2 * sh.row_weight * n
    assert sh.row_weight == ShardedSpMM.ROW_WEIGHT > 0 and abs(load - total / 8) <= 0.02 * total / 8
    assert sh.A.G == 4 and sh.A.warp is not None and sh.AT.warp is not None
    del adj, adj_t
    g = torch.Generator(device=dev); g.manual_seed(5)
    B = torch.empty((n, d), device=dev).uniform_(-1, 1, generator=g)
    dC = torch.empty((n, d), device=dev).uniform_(-1, 1, generator=g)
    sh.autotune(B, dC)
    c, db = sh.forward(B), sh.backward(dC)
    assert c.shape == (sh.hi - sh.lo, d) and db.shape == (sh.hi - sh.lo, d)
    rng = np.random.RandomState(11)
    for name, m, X, got in (("fwd", blk, B, c), ("bwd", blk_t, dC, db)):
        deg = np.diff(m.indptr)
        rows = np.unique(np.concatenate([np.argsort(deg)[-20:], np.argsort(deg)[:20], rng.choice(m.shape[0], 1500, replace=False)]))
        e, _ = _sampled_rows_vs_oracle(m, rows, X, got, dev)
        assert e <= TOL, (name, e)
        print("S-RMAT 10M block 3/8 %s: %d sampled rows (heaviest %d nnz), rel err %.1e" % (name, rows.shape[0], deg.max(), e))
    # every row: A . 1 = row sums (fp64 on the host), and linearity  A.(2B + dC) = 2 A.B + A.dC
    ones = torch.ones((n, 4), device=dev)
    s1 = sh.forward(ones)[:, 0].cpu().numpy().astype(np.float64)
    want = np.asarray(blk.astype(np.float64).sum(axis=1)).ravel()
    assert np.abs(s1 - want).max() <= TOL * max(1.0, np.abs(want).max())
    lin = sh.forward(2 * B + dC)
    assert float((lin - (2 * c + sh.forward(dC))).abs().max() / lin.abs().max()) <= TOL


def test_reddit_eighth_block_on_a_column_range_plan_full_size_vs_oracle():
    """BASELINE config 4's strong scaling at 8 GPUs, ONE rank's block at full size (S-Reddit N = 232,965, d = 602): the block
    ShardedSpMM gives rank 3 of 8 is small enough for a column-range plan (round 6: rows split by column range, a range per half
    of the XCDs) in both directions; forward and backward on the autotuned plans against the CPU oracle on sampled rows
    including the heaviest, A . 1 = row sums and linearity on ALL rows, the 1-D plan of the same block as a second opinion,
    bit-identical reruns."""
    import types
    from stochastic_gcn_amd import ops, synthetic
    from stochastic_gcn_amd.parallel import ShardedSpMM
    dev = torch.device("cuda:0")
    n, _, adj, *_ = synthetic.reddit_like(with_features=False)
    d = 602
    sh = ShardedSpMM(types.SimpleNamespace(rank=3, world=8, active=False), adj, dev, d=d)
    assert sh.A.ranged == 2 and sh.AT.ranged == 2 and sh.A.G == 1 and sh.A.nfix > 0.9 * (sh.hi - sh.lo)
    assert sh.A.ntiles <= 4096 and sh.AT.ntiles <= 4096                        # one round of resident tiles
    adj_t = adj.T.tocsr()
    blk, blk_t = adj[sh.lo:sh.hi].tocsr(), adj_t[sh.lo:sh.hi].tocsr()
    g = torch.Generator(device=dev); g.manual_seed(5)
    Bp = torch.zeros((n, 608), device=dev); Bp[:, :d] = torch.randn((n, d), device=dev, generator=g)
    dCp = torch.zeros((n, 608), device=dev); dCp[:, :d] = torch.randn((n, d), device=dev, generator=g)
    B, dC = Bp[:, :d], dCp[:, :d]
    sh.autotune(B, dC)
    assert sh.A.pace[d] > 0 and sh.AT.pace[d] > 0
    c, db = sh.forward(B), sh.backward(dC)
    assert torch.equal(sh.forward(B), c) and torch.equal(sh.backward(dC), db)
    rng = np.random.RandomState(11)
    for name, m, X, got in (("fwd", blk, B, c), ("bwd", blk_t, dC, db)):
        deg = np.diff(m.indptr)
        rows = np.unique(np.concatenate([np.argsort(deg)[-20:], np.argsort(deg)[:20], rng.choice(m.shape[0], 600, replace=False)]))
        e, _ = _sampled_rows_vs_oracle(m, rows, X, got, dev)
        assert e <= TOL, (name, e)
        print("S-Reddit block 3/8 (column-range plan) %s: %d sampled rows (heaviest %d nnz), rel err %.1e" % (name, rows.shape[0], deg.max(), e))
    one_d = ops.ColumnSweepCSR(blk, dev, G=1)
    assert not one_d.ranged
    c1 = ops.spmm_cs(one_d, B)
    assert float((c1 - c).abs().max() / c1.abs().max()) <= 1e-5
    ones = torch.ones((n, 4), device=dev)
    s1 = sh.forward(ones)[:, 0].cpu().numpy().astype(np.float64)
    want = np.asarray(blk.astype(np.float64).sum(axis=1)).ravel()
    assert np.abs(s1 - want).max() <= TOL * max(1.0, np.abs(want).max())
    mixp = 2 * Bp + dCp                                                        # (rows stay 16-byte aligned: pitch 608)
    lin = sh.forward(mixp[:, :d])
    assert float((lin - (2 * c + sh.forward(dC))).abs().max() / lin.abs().max()) <= TOL


def _train_worker(rank, world, port, native, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGCN_DIST_BACKEND="gloo")
    import contextlib
    import io
    from stochastic_gcn_amd import synthetic
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    torch.cuda.set_device(0)
    data = synthetic.reddit_like(n=6000, m=60000, f=32, classes=6, splits=(3600, 800, 1600), seed=5,
                                 with_features=True, planted=True)
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.1, layer_norm=True,
                 hidden1=64, num_fc_layers=2, batch_size=256, test_batch_size=512, cv=True, cvd=True, test_cv=True,
                 degree=1, test_degree=1, seed=1, native_step=native, max_steps=5)
    with contextlib.redirect_stdout(io.StringIO()):
        trn = Trainer(data=data, verbose=False)
        for _ in range(2):
            trn.train_epoch()
    m = trn.train_model
    m.join_history()                              # the last step's exchange (asynchronous: joined by the history's next reader)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "t%d_%d.npz" % (int(native), rank)), theta=m.theta.cpu().numpy(),
             hist=m.history[0][0].cpu().numpy(), used_program=np.array([bool(getattr(m, '_programs', {}))]))
    trn.par.shutdown()


def test_two_rank_training_program_equals_eager_and_replicas_agree(tmp_path):
    """Data-parallel minibatch training over two ranks (gradient all-reduce + history all-gather between the
    phases of the step program): the compiled step program and the eager path give bit-identical weights
    and history, and the two replicas stay identical."""
    import torch.multiprocessing as mp
    res = {}
    for native in (False, True):
        port = tg._free_port()
        mp.spawn(_train_worker, args=(2, port, native, str(tmp_path)), nprocs=2, join=True)
        res[native] = [np.load(os.path.join(str(tmp_path), "t%d_%d.npz" % (int(native), r))) for r in range(2)]
    assert res[True][0]["used_program"][0] and not res[False][0]["used_program"][0]
    for native in (False, True):
        np.testing.assert_array_equal(res[native][0]["theta"], res[native][1]["theta"])      # replicas in lock-step
        np.testing.assert_array_equal(res[native][0]["hist"], res[native][1]["hist"])
    np.testing.assert_array_equal(res[True][0]["theta"], res[False][0]["theta"])
    np.testing.assert_array_equal(res[True][0]["hist"], res[False][0]["hist"])
    assert np.abs(res[True][0]["hist"]).sum() > 0


def _oracle_pair_worker(rank, world, port, steps, out_dir):
    """One rank of BASELINE config 4 in miniature: the HIP training step with the data-parallel hooks (gradient
    all-reduce, rank-ordered history all-gather; gloo rendezvous, both ranks on cuda:0) next to the 2-rank NumPy
    oracle of tests/test_parallel_gloo.py:w_train_step, which exchanges ITS gradients and history rows over the
    same process group on CPU tensors.  Each side keeps its own weights, Adam moments and history throughout."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGCN_DIST_BACKEND="gloo")
    import model_cases as mc
    import test_model_gpu as tm
    from oracle import oracle_np as onp
    from stochastic_gcn_amd.parallel import DataParallel
    from stochastic_gcn_amd.scheduler import PyScheduler
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    par = DataParallel(backend="gloo", device=dev)
    try:
        case = mc.build_case(mc.REDDIT_MID)
        fl, c, ph = case['flags'], case['cfg'], case['ph']
        om = mc.make_oracle_model(case, seed=3)
        dm = tm._make_device_model(case, {k: v.copy() for k, v in om.params.items()})
        par.attach(dm)                                    # hooks, per-rank dropout seed, weights from rank 0
        par.history_cap = c['batch'] * 2
        names = sorted(om.params)
        train = par.shard_ids(np.sort(case['train']), c['n']).astype(np.int32)
        sch = PyScheduler(case['adj'], case['labels'], 1, [1], ph, par.sampler_seed(1), data=train, cv=True)
        out = dict(n_train=[len(train)], err_act=[], err_grad=[], err_param=[], err_hist=[], loss=[], oloss=[], acc=[], oacc=[])
        well = {}
        for step in range(steps):
            if sch.start >= sch.data.shape[0]:
                sch.start = 0
            feed = sch.minibatch(c['batch'])
            feed[ph['dropout']] = fl['dropout']
            masks = tm._masks(dm, 1.0 - fl['dropout'])
            # device: forward, backward, all-reduce(mean) of the flat gradient, Adam, history all-gather + scatter
            outs = dm.run_one_step(None, feed)
            dm.join_history()                             # (otherwise joined in front of the next step's aggregator)
            d_acts, dg = [tm._np(a) for a in dm.activations[1:]], dm.get_grads()
            # oracle: the same step, its own collectives on CPU tensors
            logits, o_acts = om.forward(feed, ph, fl['dropout'], masks)
            o_loss, o_acc, _, dlogits = om.loss_and_grad(logits, feed[ph['labels']])
            grads = om.backward(dlogits)
            flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in names]))
            par.allreduce_mean_(flat)
            off = 0
            for k in names:
                sz = grads[k].size
                grads[k] = flat.numpy()[off:off + sz].reshape(grads[k].shape).copy()
                off += sz
            om.adam_step(grads)
            hist = torch.from_numpy(om.history[0])
            par.sync_history(hist, torch.from_numpy(feed[ph['fields'][0]]), torch.from_numpy(om._new_hist[0]),
                             lambda h, i, r: onp.scatter_rows(h.numpy(), i.numpy(), r.numpy()))
            par.join_history()
            worst = 0.0
            for da, oa in zip(d_acts, o_acts):
                for dd, oo in (zip(da, oa) if isinstance(oa, tuple) else [(da, oa)]):
                    worst = max(worst, onp.rel_err(dd, oo))
            out['err_act'].append(worst)
            out['err_grad'].append(max(onp.rel_err(dg[k], grads[k]) for k in names))        # the AVERAGED gradient
            dp = dm.get_params()
            e = 0.0
            for k in names:
                well[k] = well.get(k, True) & (np.abs(grads[k]) > 1e-6)
                e = max(e, float(np.abs(dp[k] - om.params[k])[well[k]].max() / np.abs(om.params[k]).max()))
            out['err_param'].append(e)
            out['err_hist'].append(onp.rel_err(dm.history[0][0].cpu().numpy(), om.history[0]))
            out['loss'].append(outs[1]); out['oloss'].append(float(o_loss))
            out['acc'].append(outs[2]); out['oacc'].append(float(o_acc))
        torch.cuda.synchronize()
        out['theta'] = dm.theta.cpu().numpy()
        out['hist'] = dm.history[0][0].cpu().numpy()
        out['otheta'] = np.concatenate([om.params[k].ravel() for k in names])
        out['ohist'] = om.history[0]
        np.savez(os.path.join(out_dir, "o%d.npz" % rank), **{k: np.asarray(v) for k, v in out.items()})
    finally:
        par.shutdown()


def test_two_rank_cvd_pp_training_steps_match_the_two_rank_oracle(tmp_path):
    """BASELINE config 4 (Reddit CVD+PP, vertex-range sharding, gradient all-reduce, H-a history exchange) on the GPU
    against the ORACLE, not against itself: two ranks x three consecutive steps of the Reddit recipe on a 12 k-vertex
    S-Reddit-shaped graph, HIP path vs the 2-rank NumPy oracle -- every layer activation, loss, accuracy, the averaged
    gradient, the Adam-updated weights and the rank-ordered history; the device replicas (and the oracle replicas)
    stay identical across the ranks."""
    import torch.multiprocessing as mp
    steps, port = 3, tg._free_port()
    mp.spawn(_oracle_pair_worker, args=(2, port, steps, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), "o%d.npz" % k)) for k in range(2)]
    assert r[0]["n_train"][0] + r[1]["n_train"][0] == 256 * 3 and min(r[0]["n_train"][0], r[1]["n_train"][0]) > 256
    for k in range(2):
        assert r[k]["err_act"].max() <= TOL, ("activations", k, r[k]["err_act"])
        assert r[k]["err_grad"].max() <= 1e-4, ("mean gradient", k, r[k]["err_grad"])
        assert r[k]["err_param"].max() <= 5e-4, ("weights", k, r[k]["err_param"])
        assert r[k]["err_hist"].max() <= TOL, ("history", k, r[k]["err_hist"])
        assert np.abs(r[k]["loss"] - r[k]["oloss"]).max() <= 1e-4 * max(1.0, np.abs(r[k]["oloss"]).max())
        assert np.abs(r[k]["acc"] - r[k]["oacc"]).max() <= 1e-6
    np.testing.assert_array_equal(r[0]["theta"], r[1]["theta"])          # device replicas in lock-step
    np.testing.assert_array_equal(r[0]["hist"], r[1]["hist"])
    np.testing.assert_array_equal(r[0]["otheta"], r[1]["otheta"])        # and the oracle's
    np.testing.assert_array_equal(r[0]["ohist"], r[1]["ohist"])
    assert np.abs(r[0]["hist"]).sum() > 0 and not np.array_equal(r[0]["loss"], r[1]["loss"])   # two different shards
    print("2-rank CVD+PP vs 2-rank oracle, 3 steps: activations %.1e  mean gradient %.1e  weights %.1e  history %.1e"
          % (max(x["err_act"].max() for x in r), max(x["err_grad"].max() for x in r),
             max(x["err_param"].max() for x in r), max(x["err_hist"].max() for x in r)))


# ---- RCCL itself: one rank (SGCN_FORCE_PG=1) ---------------------------------------------------------------------------
def _rccl_worker(rank, world, port, force, out_dir, native=True, program=True, overlap=True):
    """Three training steps + an all-gathered sharded product, with a REAL one-rank RCCL process group (force; the step's
    collectives on the library's own communicator -- native -- or as torch.distributed calls) or without any process group
    (not force): the same numbers, bit for bit."""
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SGCN_FORCE_PG="1" if force else "0", SGCN_NATIVE_COLL="1" if native else "0",
                      SGCN_EXCHANGE_OVERLAP="1" if overlap else "0")
    os.environ.pop("SGCN_DIST_BACKEND", None)
    import contextlib
    import io
    import torch.distributed as dist
    from stochastic_gcn_amd import synthetic
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.parallel import ShardedSpMM
    from stochastic_gcn_amd.train import Trainer
    torch.cuda.set_device(0)
    data = synthetic.reddit_like(n=6000, m=60000, f=32, classes=6, splits=(3600, 800, 1600), seed=5,
                                 with_features=True, planted=True)
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.1, layer_norm=True,
                 hidden1=64, num_fc_layers=2, batch_size=256, test_batch_size=512, cv=True, cvd=True, test_cv=True,
                 degree=1, test_degree=1, seed=1, native_step=program, max_steps=3)
    with contextlib.redirect_stdout(io.StringIO()):
        trn = Trainer(data=data, verbose=False)
        trn.train_epoch()
    m, par = trn.train_model, trn.par
    assert par.active == force and dist.is_initialized() == force
    if force:
        from stochastic_gcn_amd._ffi import lib
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and par.backend == "nccl"
        assert par.native == native and lib.sgcn_coll_world() == (1 if native else 0) and m.native_coll == (1 if native else 0)
        if native and program:                 # the collectives were ops of the step program: one foreign call per step
            progs = [p for p in m._programs.values() if p is not None]
            assert progs and all(p.native_world == 1 for p in progs)
            # the history exchange: on the library's exchange stream right behind the aggregator (a second communicator,
            # round 6) -- ops of the forward / backward phase, their last argument 2 -- or behind the optimizer on the step's own
            assert par.exchange_overlap == overlap == bool(lib.sgcn_coll_has_exchange())
            where = (lambda p: p.ops_fb) if overlap else (lambda p: p.ops_hist)
            assert all(any(op == 29 for op, _ in p.ops_fb) and all(any(op == o for op, _ in where(p)) for o in (30, 31, 32)) for p in progs)
            assert all(args[-1][2] == (2 if overlap else 0) for p in progs for op, args in where(p) if op in (30, 31, 32))
            if overlap:                        # ... in front of the loss: beside the backward pass, not behind it
                for p in progs:
                    codes = [op for op, _ in p.ops_fb]
                    assert max(codes.index(o) for o in (30, 31, 32)) < min(codes.index(o) for o in (5, 13) if o in codes)
                    assert not any(op in (30, 31, 32) for op, _ in p.ops_hist)
        assert m.grad_hook is not None and m.history_hook is not None and len(par._pending) == (0 if native else 1)
        # the last step's exchange is still in flight -- and READING the history is what lands it (the public
        # names join; nothing can see a replica that lacks this step's rows, its own included)
        hist = m.history[0][0]
        assert len(par._pending) == 0 and hist is m._history[0][0]
    m.join_history()
    # the collectives on their own: mean all-reduce (identity on one rank), padded row all-gather
    g = torch.Generator(device=trn.device); g.manual_seed(3)
    flat = torch.randn(1000, device=trn.device, generator=g)
    ref = flat.clone()
    par.allreduce_mean_(flat)
    sh = ShardedSpMM(par, data[2], trn.device, kernel="cs", d=30)
    X = torch.randn((data[0], 30), device=trn.device, generator=g)
    full = sh.allgather_rows(X[sh.lo:sh.hi].contiguous())
    c = sh.forward_allgather(X[sh.lo:sh.hi].contiguous())
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rccl%d%d%d%d.npz" % (int(force), int(native), int(program), int(overlap))), theta=m.theta.cpu().numpy(), hist=m.history[0][0].cpu().numpy(),
             used_program=np.array([bool([p for p in getattr(m, '_programs', {}).values() if p is not None])]), steps=np.array([m.adam_t]),
             allreduce_ok=np.array([bool(torch.equal(flat, ref))]), gathered_ok=np.array([bool(torch.equal(full, X))]),
             c=c.cpu().numpy())
    par.shutdown()


def test_rccl_one_rank_training_steps_and_allgather_equal_the_single_process_path(tmp_path):
    """First contact with RCCL: init_process_group("nccl", world_size=1, device_id=...),
    DataParallel.attach, three program-path training steps -- prog.run('fb') -> all-reduce (ReduceOp.AVG) ->
    prog.run('opt') -> asynchronous history all-gather, joined in front of the next step's aggregator -- and
    ShardedSpMM.allgather_rows, against the same run without a process group: bit-identical weights, history and
    products (gloo stages through the host and synchronises; only RCCL can show a stream-ordering bug)."""
    import torch.multiprocessing as mp
    res = {}
    # (process group, the library's own communicator, step programs): native and torch collectives, compiled and eager
    # steps, against the run without a process group
    # (+ round 6: the history exchange on the exchange stream with its own communicator -- the default -- or behind the optimizer)
    modes = [(True, True, True, True), (True, True, True, False), (True, False, True, True), (True, True, False, True),
             (False, True, True, True)]
    for mode in modes:
        mp.spawn(_rccl_worker, args=(1, tg._free_port(), mode[0], str(tmp_path), mode[1], mode[2], mode[3]), nprocs=1, join=True)
        res[mode] = np.load(os.path.join(str(tmp_path), "rccl%d%d%d%d.npz" % tuple(int(x) for x in mode)))
    ref = res[(False, True, True, True)]
    for mode in modes:
        assert res[mode]["used_program"][0] == mode[2] and res[mode]["steps"][0] == 3
        assert res[mode]["allreduce_ok"][0] and res[mode]["gathered_ok"][0]
        np.testing.assert_array_equal(res[mode]["c"], ref["c"])
        if mode[2]:                                # (the eager path sums in another order than the program: its own check below)
            np.testing.assert_array_equal(res[mode]["theta"], ref["theta"])
            np.testing.assert_array_equal(res[mode]["hist"], ref["hist"])
    eager = res[(True, True, False, True)]
    assert np.abs(eager["theta"] - ref["theta"]).max() <= 1e-5 and np.abs(eager["hist"] - ref["hist"]).max() <= 1e-4
    assert np.abs(ref["hist"]).sum() > 0


@pytest.mark.parametrize("world,sizes", [(3, [64, 17, 0]), (8, [64, 17, 0, 64, 33, 1, 64, 50])])
def test_history_exchange_pack_and_apply_in_rank_order(world, sizes):
    """sgcn_hist_pack_f32 / sgcn_hist_apply_f32 with THREE and with EIGHT ranks' blocks in the gathered buffer (the
    all-gather itself is the one call that needs that many GPUs): padded ids are skipped, and a vertex that several ranks
    updated -- with eight writers on a quarter of the vertices most are -- keeps the highest rank's row: what
    DataParallel.join_history does with one scatter call per rank."""
    from stochastic_gcn_amd._ffi import check, lib
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    N, d, cap = 500, 37, 64
    H0 = rng.standard_normal((N, 40)).astype(np.float32)          # history with a pitch
    H = torch.from_numpy(H0).to(dev)
    want = H0.copy()
    recv = torch.empty(world * cap * (d + 1), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        n = sizes[r]
        ids = rng.choice(N // 4, n, replace=False).astype(np.int32)              # a quarter of the vertices: ranks collide
        rows = rng.standard_normal((max(n, 1), 48)).astype(np.float32)           # rows with a pitch
        send = recv[r * cap * (d + 1):(r + 1) * cap * (d + 1)]
        idt, rt = torch.from_numpy(ids).to(dev), torch.from_numpy(rows).to(dev)
        check(lib.sgcn_hist_pack_f32(idt.data_ptr(), n, rt.data_ptr(), 48, d, cap, send.data_ptr(), st))
        got = send.cpu().numpy()
        np.testing.assert_array_equal(got[:n], ids)
        assert (got[n:cap] == -1).all()
        np.testing.assert_array_equal(got[cap:].view(np.float32).reshape(cap, d)[:n], rows[:n, :d])
        want[ids, :d] = rows[:n, :d]
    H2 = H.clone()
    check(lib.sgcn_hist_apply_f32(H.data_ptr(), 40, recv.data_ptr(), world, cap, d, None, st))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(H.cpu().numpy(), want)
    # ... and the two-launch form (a claim table: one zeroed word per history row, zero again afterwards), twice
    owner = torch.zeros(N, dtype=torch.int32, device=dev)
    for _ in range(2):
        H3 = H2.clone()
        check(lib.sgcn_hist_apply_f32(H3.data_ptr(), 40, recv.data_ptr(), world, cap, d, owner.data_ptr(), st))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(H3.cpu().numpy(), want)
        assert int(owner.abs().sum()) == 0
