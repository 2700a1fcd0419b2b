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
    whole = ops.DeviceCSR.from_scipy(adj, dev, with_transpose=True)
    want_c, want_db = ops.spmm(whole, B), ops.spmm(whole.transpose, dC)
    rows, cs, dbs, groups = 0, [], [], set()
    for r in range(8):
        sh = ShardedSpMM(types.SimpleNamespace(rank=r, world=8, active=False), adj, dev, d=d)
        assert sh.lo == rows
        rows = sh.hi
        groups.add(sh.A.G)
        cs.append(sh.forward(B))
        dbs.append(sh.backward(dC))
    # the block of the heaviest rows is dense enough (average degree > 300) to keep one group; the others take two
    assert rows == n and 2 in groups and groups <= {1, 2}
    assert onp.rel_err(torch.cat(cs).cpu().numpy(), want_c.cpu().numpy()) <= TOL
    assert onp.rel_err(torch.cat(dbs).cpu().numpy(), want_db.cpu().numpy()) <= TOL


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
    torch.cuda.synchronize()
    m = trn.train_model
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
