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
