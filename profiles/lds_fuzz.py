"""Randomised parity of the LDS sweep against the oracle: shapes around the tile / wave / ring boundaries, row-constant, column-
constant, arbitrary and all-ones values, labels, min_reuse, split rows, both rings, row scale, beta, pitch padding.
usage: python profiles/lds_fuzz.py [seed] [cases]"""
import sys, numpy as np, scipy.sparse as sp, torch
R = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))); sys.path[:0] = [R, R + "/tests", R + "/oracle"]
from stochastic_gcn_amd import ops
import oracle_np as onp
dev = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    M = int(rng.choice([1, 7, 95, 96, 97, 700, 769, 1500, 3000])); K = int(rng.choice([1, 5, 130, 129, 1000, 2500]))
    dens = float(rng.choice([0.002, 0.02, 0.1, 0.4]))
    a = sp.random(M, K, density=dens, format='csr', random_state=rng, dtype=np.float32)
    kind = rng.choice(["row", "col", "gen", "ones"])
    a.data[:] = rng.standard_normal(a.nnz).astype(np.float32)
    if kind == "row":
        a = sp.diags(rng.rand(M).astype(np.float32) + 0.5).dot((a != 0).astype(np.float32)).tocsr().astype(np.float32)
    elif kind == "col":
        a = (a != 0).astype(np.float32).dot(sp.diags(rng.rand(K).astype(np.float32) + 0.5)).tocsr().astype(np.float32)
    elif kind == "ones":
        a.data[:] = 1.0
    a.sort_indices()
    d = int(rng.choice([2, 30, 128, 130, 256, 602])); pad = (-d) % 4 + int(rng.choice([0, 4, 8]))     # (rows 16-byte aligned: the kernels' contract)
    lab = None if rng.rand() < 0.4 else (rng.randint(0, 4, M).astype(np.int32), rng.randint(0, 4, K).astype(np.int32))
    mr = int(rng.choice([1, 2, 3])); T = int(rng.choice([0, 8, 64])); ring = int(rng.choice([0, 80]))
    B = rng.standard_normal((K, d + pad)).astype(np.float32)
    rs = rng.rand(M).astype(np.float32) if rng.rand() < 0.5 else None
    beta = float(rng.choice([0.0, 0.5, 1.0]))
    c0 = rng.standard_normal((M, d + pad)).astype(np.float32)
    try:
        A = ops.LdsSweepCSR(a, dev, labels=lab, min_reuse=mr, T=T, ring_slots=ring, general=bool(kind == "gen" and rng.rand() < 0.5))
        out = torch.from_numpy(c0.copy()).to(dev)
        Bd = torch.from_numpy(B).to(dev)[:, :d]
        ops.spmm_lds(A, Bd, out=out[:, :d], rscale=None if rs is None else torch.from_numpy(rs).to(dev), beta=beta)
        ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d], rscale=rs, C_in=c0[:, :d], beta=beta)
        err = onp.rel_err(out[:, :d].cpu().numpy(), ref)
        okpad = np.array_equal(out[:, d:].cpu().numpy(), c0[:, d:])
        if not (err <= 1e-4 and okpad):
            bad += 1
            print("FAIL", case, M, K, dens, kind, d, pad, mr, T, ring, beta, "err", err, "pad", okpad, "unit", A.unit, A.col_fold is not None)
    except Exception as e:
        bad += 1
        print("EXC", case, M, K, dens, kind, d, pad, mr, T, ring, beta, repr(e)[:200])
print("done, failures:", bad)
