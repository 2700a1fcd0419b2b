"""Randomised parity of the column sweep (one / two / four lane groups per wave; round 6: + column-range plans) against the oracle: shapes
around the bin / tile boundaries, split rows, gathered operand rows, row / column scales, beta, pitch padding, paced and unpaced.
usage: python profiles/cs_fuzz.py [seed] [cases]"""
import sys
import numpy as np
import scipy.sparse as sp
import torch
R = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))); sys.path[:0] = [R, R + "/tests", R + "/oracle"]
from stochastic_gcn_amd import ops
import oracle_np as onp
dev = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    M = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 700, 4097, 9000])); K = int(rng.choice([1, 5, 130, 1000, 5000]))
    dens = float(rng.choice([0.002, 0.02, 0.1, 0.4]))
    a = sp.random(M, K, density=dens, format='csr', random_state=rng, dtype=np.float32)
    a.data[:] = rng.standard_normal(a.nnz).astype(np.float32)
    if rng.rand() < 0.4 and a.nnz:            # skewed columns (the clock in work coordinates: a warp table by itself)
        coo = a.tocoo()
        a = sp.coo_matrix((coo.data, (coo.row, (coo.col.astype(np.int64) ** 2 // max(K, 1)).astype(np.int32))), shape=(M, K)).tocsr()
        a.sum_duplicates()
    a.sort_indices()
    warp = [True, False, 'auto'][int(rng.randint(3))]
    G = int(rng.choice([1, 2, 4]))
    d = int(rng.choice([4, 30, 64, 66, 128, 130, 320, 602])); pad = (-d) % 4 + int(rng.choice([0, 4]))
    T = int(rng.choice([0, 8, 64])); align = int(rng.choice([0, 64, 2048]))
    B = rng.standard_normal((K + 50, d + pad)).astype(np.float32)
    gidx = rng.choice(K + 50, K, replace=False).astype(np.int32) if rng.rand() < 0.4 else None
    rs = rng.rand(M).astype(np.float32) if rng.rand() < 0.5 else None
    cs = rng.rand(K).astype(np.float32) if rng.rand() < 0.5 else None
    beta = float(rng.choice([0.0, 0.5, 1.0]))
    c0 = rng.standard_normal((M, d + pad)).astype(np.float32)
    ranged = G == 1 and rng.rand() < 0.5            # rows split by column range (ops.ColumnSweepCSR col_ranges)
    try:
        if ranged:
            A = ops.ColumnSweepCSR(a, dev, T=T, col_ranges=int(rng.choice([2, 2, 3, 4])))
            assert A.ranged >= 2
        else:
            A = ops.ColumnSweepCSR(a, dev, T=T, G=G, warp=warp, **({} if G == 1 else {"align": align if rng.rand() < 0.7 else 'auto'}))
        A.pace[d] = int(rng.choice([-1, 100, 300]))
        t = lambda x: None if x is None else torch.from_numpy(x).to(dev)     # noqa: E731
        out = t(c0.copy())
        Bd = t(B)[:K if gidx is None else K + 50, :d]
        ops.spmm_cs(A, Bd, out=out[:, :d], gidx=t(gidx), rscale=t(rs), cscale=t(cs), beta=beta)
        ref = onp.spmm(a.indptr, a.indices, a.data, B[:K if gidx is None else K + 50, :d], gidx=gidx, rscale=rs, cscale=cs, C_in=c0[:, :d], beta=beta)
        err = onp.rel_err(out[:, :d].cpu().numpy(), ref)
        okpad = np.array_equal(out[:, d:].cpu().numpy(), c0[:, d:])
        if not (err <= 1e-4 and okpad):
            bad += 1
            print("FAIL", case, M, K, dens, G, "ranged" if ranged else "", d, pad, T, align, beta, "err", err, "pad", okpad)
    except Exception as e:
        bad += 1
        print("EXC", case, M, K, dens, G, "ranged" if ranged else "", d, pad, T, align, beta, repr(e)[:200])
print("done, failures:", bad)
