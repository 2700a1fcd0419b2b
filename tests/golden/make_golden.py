"""Generate the golden fixtures of tests/golden/ from the REAL reference C++.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

It drives oracle/_ref/libsgcn_ref.so (= gcn/scheduler.cpp + gcn/mult.cpp + gcn/history.cpp
compiled from the reference tree, see oracle/Makefile) through oracle/ref_binding.py and
stores INPUTS AND OUTPUTS ONLY (numbers), never reference source:

  sampler_small.npz  G2/G3/G4: full feed-dicts for the 11-node tree of gcn/test_scheduler.py
                     and a 50-node random graph x seeds x {NS, CV, IS} x L x degrees x 3
                     consecutive batch() calls (captures the in-place permutation state).
  sampler_big.npz    G3/G4 on a 2k-node power-law graph with isolated vertices: the graph,
                     the batches and a SHA-256 digest of every output array (+ sizes).
  mult.npz           G1: the known answers of gcn/test_mult.cpp + random trees.
  slice.npz          G5: history.slice / dense_slice inputs and outputs incl. empty rows
                     and the nnz == 0 case.
"""
import hashlib
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_binding as rb  # noqa: E402


def placeholders(L):
    return {'adj': ['adj_%d' % i for i in range(L)], 'madj': ['madj_%d' % i for i in range(L)],
            'fadj': ['fadj_%d' % i for i in range(L)],
            'fields': ['fields_%d' % i for i in range(L + 1)],
            'ffields': ['ffields_%d' % i for i in range(L + 1)],
            'scales': ['scales_%d' % i for i in range(L)], 'labels': 'labels'}


def tree_graph():
    """The graph of gcn/test_scheduler.py:10-21 (11-node tree, row-normalised)."""
    edges = np.array([(0, 1), (0, 2), (0, 3), (1, 4), (1, 5), (1, 6), (2, 7), (2, 8), (2, 9), (3, 10)])
    adj = sp.csr_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(11, 11),
                        dtype=np.float32)
    adj = adj + adj.transpose()
    deg = np.array(adj.sum(axis=0)).flatten()
    adj = sp.diags(1.0 / deg, 0).dot(adj).tocsr().astype(np.float32)
    # NOT sorted: the stored column order is whatever the script's scipy expressions yield
    # (the sampler permutes rows in storage order); the arrays are saved in the fixture.
    return adj


def random_graph(n, avg, seed, zipf=False, isolated=False):
    rng = np.random.RandomState(seed)
    m = n * avg // 2
    src = ((rng.zipf(1.6, m) - 1) % n) if zipf else rng.randint(0, n, m)
    dst = rng.randint(0, n, m)
    a = sp.csr_matrix((np.ones(m, np.float32), (src, dst)), shape=(n, n))
    a = a + a.T
    a.data[:] = 1
    if isolated:
        keep = np.ones(n, bool)
        keep[rng.choice(n, n // 20, replace=False)] = False
        d = sp.diags(keep.astype(np.float32))
        a = (d @ a @ d).tocsr()
        a.eliminate_zeros()
    rs = np.array(a.sum(1)).ravel()
    a = sp.diags(1.0 / (rs + 1e-20)).dot(a).tocsr().astype(np.float32)
    a.sort_indices()
    return a


def flatten_feed(fd, prefix, out):
    for k, v in fd.items():
        if isinstance(v, tuple):
            out["%s/%s/idx" % (prefix, k)] = np.asarray(v[0], dtype=np.int32).reshape(-1, 2)
            out["%s/%s/w" % (prefix, k)] = np.asarray(v[1], dtype=np.float32)
            out["%s/%s/shape" % (prefix, k)] = np.asarray(v[2], dtype=np.int64)
        else:
            out["%s/%s" % (prefix, k)] = np.asarray(v)


def digest(a):
    a = np.ascontiguousarray(a)
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)


def sampler_cases(n, big):
    seeds = [0, 1, 123]
    modes = [(False, False), (True, False), (False, True)]      # (cv, importance)
    degs = [1, 2, 20, 10000]
    for seed in seeds:
        for cv, imp in modes:
            for L in (1, 2):
                for deg in degs:
                    if big and deg == 10000 and (L == 2 or seed != 0):
                        continue
                    yield seed, cv, imp, L, deg


def make_sampler(path, graphs, big):
    out = {}
    for gname, adj in graphs.items():
        n = adj.shape[0]
        out["%s/indptr" % gname] = adj.indptr.astype(np.int32)
        out["%s/indices" % gname] = adj.indices.astype(np.int32)
        out["%s/data" % gname] = adj.data.astype(np.float32)
        labels = np.zeros((n, 2), np.float32)
        for seed, cv, imp, L, deg in sampler_cases(n, big):
            sch = rb.RefPyScheduler(adj, labels, L, [deg] * L, placeholders(L), seed, cv=cv,
                                    importance=imp)
            rng = np.random.RandomState(1000 + seed)
            for it in range(3):
                ids = rng.choice(n, min(n, 1 if gname == 'tree' else 17), replace=False).astype(np.int32)
                if gname == 'tree' and it == 0:
                    ids = np.array([0], dtype=np.int32)          # gcn/test_scheduler.py:36
                fd = sch.batch(ids)
                prefix = "%s/s%d_cv%d_is%d_L%d_d%d/it%d" % (gname, seed, cv, imp, L, deg, it)
                out[prefix + "/ids"] = ids
                flat = {}
                flatten_feed(fd, prefix, flat)
                if big:
                    for k, v in flat.items():
                        out[k + "#sha"] = digest(v)
                        out[k + "#shape"] = np.asarray(v.shape, dtype=np.int64)
                else:
                    out.update(flat)
            # the private CSR copy after the 3 batches (statefulness probe)
            key = "%s/s%d_cv%d_is%d_L%d_d%d/adj_i_after" % (gname, seed, cv, imp, L, deg)
            perm = sch.c_sch.ivec(rb.I_ADJ_I)
            out[key + ("#sha" if big else "")] = digest(perm) if big else perm
    np.savez_compressed(path, **out)
    print(path, len(out), "arrays", os.path.getsize(path), "bytes")


def make_mult(path):
    out = {}
    cases = {"a": [3, 2, 1, 3], "b": [3, 2, 1, 3, 4], "c": [1, 0.1, 100, 10000, 1000]}
    rng = np.random.RandomState(7)
    cases["r37"] = rng.rand(37).tolist()
    cases["r64"] = rng.rand(64).tolist()
    cases["one"] = [2.5]
    for name, p in cases.items():
        m = rb.RefMult(p)
        out[name + "/prob"] = np.asarray(p, np.float32)
        out[name + "/bit"] = m.bit
        us = np.array([0, 2, 4, 5.5, 7, 10, 14, 1e9, 0.3, 0.999], dtype=np.float32)
        out[name + "/u"] = us
        out[name + "/query_u"] = np.array([m.query_u(u) for u in us], dtype=np.int32)
        out[name + "/draws"] = np.array([m.query() for _ in range(len(p))], dtype=np.int32)
    np.savez_compressed(path, **out)
    print(path, len(out), "arrays", os.path.getsize(path), "bytes")


def make_slice(path):
    out = {}
    rng = np.random.RandomState(3)
    a = sp.random(40, 23, density=0.15, format='lil', random_state=rng, dtype=np.float32)
    for r in (5, 17, 18, 33):           # guaranteed empty rows (the nnz == 0 edge case)
        a[r, :] = 0
    a = a.tocsr()
    a.eliminate_zeros()
    a.sort_indices()
    out["a/indptr"], out["a/indices"], out["a/data"] = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    out["a/shape"] = np.asarray(a.shape, np.int64)
    empty_rows = np.where(np.diff(a.indptr) == 0)[0].astype(np.int32)
    cases = {"mixed": rng.choice(40, 12, replace=False).astype(np.int32),
             "dups": np.array([3, 3, 7, 3], dtype=np.int32),
             "all": np.arange(40, dtype=np.int32)}
    assert len(empty_rows) >= 3
    cases["empty_only"] = empty_rows[:3]
    cases["with_empty"] = np.array([5, 0, 17, 1, 33], dtype=np.int32)
    for name, r in cases.items():
        res = rb.ref_slice(a, r)
        out["slice/%s/r" % name] = r
        if sp.issparse(res):
            out["slice/%s/is_empty_csr" % name] = np.asarray(res.shape, np.int64)
        else:
            out["slice/%s/indices" % name], out["slice/%s/data" % name], out["slice/%s/shape" % name] = res
    d = rng.standard_normal((40, 19)).astype(np.float32)
    out["dense/a"] = d
    for name, r in cases.items():
        out["dense/%s/out" % name] = rb.ref_dense_slice(d, r)
    np.savez_compressed(path, **out)
    print(path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if not rb.available():
        sys.exit("oracle/_ref/libsgcn_ref.so missing: run `make -C oracle ref` first")
    make_sampler(os.path.join(HERE, "sampler_small.npz"),
                 {"tree": tree_graph(), "rand50": random_graph(50, 6, 1)}, big=False)
    make_sampler(os.path.join(HERE, "sampler_big.npz"),
                 {"pl2k": random_graph(2000, 12, 2, zipf=True, isolated=True)}, big=True)
    make_mult(os.path.join(HERE, "mult.npz"))
    make_slice(os.path.join(HERE, "slice.npz"))
