"""Deterministic synthetic stand-ins for the reference's datasets (no datasets and no network
on the box).  Shapes and statistics follow SURVEY.md §8d; normalisations follow the
reference loaders (gcn/utils.py:120-136 GCN sets, :299-309 GraphSAGE sets).  The return
value of ``load_data`` has the reference's 10-tuple layout (gcn/utils.py:183,335):

    num_data, train_adj, full_adj, feats, train_feats, test_feats, labels,
    train_data, val_data, test_data

``train_feats`` / ``test_feats`` (the PP products  train_adj . feats, full_adj . feats,
gcn/utils.py:169-170,321-322) are left as ``None`` here: the training driver computes them
on the GPU with the SpMM kernel (that product is K11 of the hot path).
"""
import os

import numpy as np
import scipy.sparse as sp


def _row_normalize(adj):
    """D^-1 A  (gcn/utils.py:120-126, :304-309)."""
    rowsum = np.array(adj.sum(1)).flatten()
    d_inv = 1.0 / (rowsum + 1e-20)
    out = sp.diags(d_inv, 0).dot(adj).tocsr().astype(np.float32)
    out.sort_indices()
    return out


def _gcn_normalize(adj):
    """D^-1/2 (A + I) D^-1/2  (gcn/utils.py:127-136)."""
    adj = adj + sp.eye(adj.shape[0], dtype=np.float32)
    rowsum = np.array(adj.sum(1)).flatten() + 1e-20
    d = np.power(rowsum, -0.5)
    d[np.isinf(d)] = 0.0
    dm = sp.diags(d, 0)
    out = adj.dot(dm).transpose().dot(dm).tocsr().astype(np.float32)
    out.sort_indices()
    return out


def zipf_uniform_edges(n, m, s, rng):
    """m undirected edges: source ~ bounded Zipf(s) over a random vertex order, destination
    uniform (SURVEY.md §8d S-Reddit generator).  Returns a symmetric 0/1 CSR without
    duplicates."""
    w = np.arange(1, n + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    rank = np.searchsorted(cdf, rng.random_sample(m)).astype(np.int64)
    np.minimum(rank, n - 1, out=rank)
    perm = rng.permutation(n)              # hubs get arbitrary vertex ids, as in real data
    src = perm[rank]
    dst = rng.randint(0, n, m)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    a = sp.coo_matrix((np.ones(src.shape[0], dtype=np.float32), (src, dst)), shape=(n, n)).tocsr()
    a = a + a.T
    a.data[:] = 1.0
    a.sort_indices()
    return a


def sbm_zipf_edges(n, m, s, comm, p_in, rng):
    """m undirected edges with planted communities: source ~ bounded Zipf(s) as in
    ``zipf_uniform_edges``; the destination is uniform INSIDE the source's community with
    probability p_in, uniform over all vertices otherwise.  Same degree law as S-Reddit, plus the
    block structure real Reddit has (posts of one subreddit are mostly linked to each other)."""
    w = np.arange(1, n + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    rank = np.searchsorted(cdf, rng.random_sample(m)).astype(np.int64)
    np.minimum(rank, n - 1, out=rank)
    perm = rng.permutation(n)
    src = perm[rank]
    members = np.argsort(comm, kind='stable')                 # vertices grouped by community
    start = np.concatenate([[0], np.cumsum(np.bincount(comm))])
    cs = comm[src]
    inside = rng.random_sample(m) < p_in
    pick = start[cs] + (rng.random_sample(m) * (start[cs + 1] - start[cs])).astype(np.int64)
    dst = np.where(inside, members[np.minimum(pick, start[cs + 1] - 1)], rng.randint(0, n, m))
    keep = src != dst
    src, dst = src[keep], dst[keep]
    a = sp.coo_matrix((np.ones(src.shape[0], dtype=np.float32), (src, dst)), shape=(n, n)).tocsr()
    a = a + a.T
    a.data[:] = 1.0
    a.sort_indices()
    return a


def er_edges(n, m, rng):
    src = rng.randint(0, n, m)
    dst = rng.randint(0, n, m)
    keep = src != dst
    a = sp.coo_matrix((np.ones(int(keep.sum()), dtype=np.float32), (src[keep], dst[keep])),
                      shape=(n, n)).tocsr()
    a = a + a.T
    a.data[:] = 1.0
    a.sort_indices()
    return a


def _splits(n, n_train, n_val, n_test, rng):
    perm = rng.permutation(n)
    tr = np.sort(perm[:n_train]).astype(np.int32)
    va = np.sort(perm[n_train:n_train + n_val]).astype(np.int32)
    te = np.sort(perm[n_train + n_val:n_train + n_val + n_test]).astype(np.int32)
    return tr, va, te


def _onehot(n, c, rng):
    y = np.zeros((n, c), dtype=np.float32)
    y[np.arange(n), rng.randint(0, c, n)] = 1.0
    return y


def _sparse_features(n, f, per_row, rng):
    """Row-normalised sparse bag-of-words like the Planetoid sets (gcn/utils.py:138-143)."""
    cols = rng.randint(0, f, (n, per_row))
    rows = np.repeat(np.arange(n), per_row)
    x = sp.coo_matrix((np.ones(n * per_row, dtype=np.float32), (rows, cols.ravel())),
                      shape=(n, f)).tocsr()
    x.data[:] = 1.0
    rowsum = np.array(x.sum(1)).flatten() + 1e-9
    x = sp.diags(1.0 / rowsum, 0).dot(x).tocsr().astype(np.float32)
    x.sort_indices()
    return x


def planted_labels(full_adj, feats, classes, rng):
    """One-hot labels from a random linear teacher on the 1-hop aggregate, argmax((A X) W): a
    learnable target, so end-to-end tests can check that training converges."""
    w = rng.standard_normal((feats.shape[1], classes)).astype(np.float32)
    z = full_adj.dot(feats if not sp.issparse(feats) else feats.tocsr()).dot(w) \
        if not sp.issparse(feats) else np.asarray(full_adj.dot(feats).dot(w))
    y = np.zeros((feats.shape[0], classes), dtype=np.float32)
    y[np.arange(feats.shape[0]), np.asarray(z).argmax(axis=1)] = 1.0
    return y


def reddit_like(n=232965, m=11600000, f=602, classes=41, splits=(152410, 23699, 55334),
                seed=1, with_features=True, planted=False):
    """S-Reddit: Zipf(0.6)-source x uniform-destination graph, D^-1 A,
    dense N(0,1) features.  Defaults give nnz ~= 23.17 M, avg degree ~= 99.5."""
    rng = np.random.RandomState(seed)
    a = zipf_uniform_edges(n, m, 0.6, rng)
    full_adj = _row_normalize(a)
    tr, va, te = _splits(n, splits[0], splits[1], splits[2], rng)
    # training graph = edges among non-val/test vertices (GraphSAGE "train_removed")
    is_train = np.ones(n, dtype=bool)
    is_train[va] = False
    is_train[te] = False
    dm = sp.diags(is_train.astype(np.float32), 0)
    at = dm.dot(a).dot(dm).tocsr()
    at.eliminate_zeros()
    train_adj = _row_normalize(at)
    labels = _onehot(n, classes, rng)
    feats = rng.standard_normal((n, f)).astype(np.float32) if with_features else None
    if planted and with_features:
        labels = planted_labels(full_adj, feats, classes, rng)
    return n, train_adj, full_adj, feats, None, None, labels, tr, va, te


def reddit_sbm(n=232965, m=11600000, f=602, classes=41, splits=(152410, 23699, 55334), p_in=0.8,
               seed=1, with_features=False):
    """S-Reddit-SBM: the S-Reddit degree law with ``classes`` planted communities (sizes ~
    rank^-0.5, vertex ids scattered at random -- the raw matrix shows no block structure), a
    fraction p_in of every vertex's edges inside its community, labels = the community.  The
    locality-bearing companion of ``reddit_like`` (whose destinations are uniform, i.e. which has
    no structure any reordering could find).  Same 10-tuple."""
    rng = np.random.RandomState(seed)
    w = np.arange(1, classes + 1, dtype=np.float64) ** (-0.5)
    comm = np.searchsorted(np.cumsum(w) / w.sum(), rng.random_sample(n)).astype(np.int64)
    np.minimum(comm, classes - 1, out=comm)
    a = sbm_zipf_edges(n, m, 0.6, comm, p_in, rng)
    full_adj = _row_normalize(a)
    tr, va, te = _splits(n, splits[0], splits[1], splits[2], rng)
    is_train = np.ones(n, dtype=bool)
    is_train[va] = False
    is_train[te] = False
    dm = sp.diags(is_train.astype(np.float32), 0)
    at = dm.dot(a).dot(dm).tocsr()
    at.eliminate_zeros()
    train_adj = _row_normalize(at)
    labels = np.zeros((n, classes), dtype=np.float32)
    labels[np.arange(n), comm] = 1.0
    feats = rng.standard_normal((n, f)).astype(np.float32) if with_features else None
    return n, train_adj, full_adj, feats, None, None, labels, tr, va, te


def planetoid_like(n, m, f, per_row, classes, n_train, n_val, n_test, normalization, seed, planted=False):
    rng = np.random.RandomState(seed)
    a = er_edges(n, m, rng)
    adj = _gcn_normalize(a) if normalization == 'gcn' else _row_normalize(a)
    feats = _sparse_features(n, f, per_row, rng)
    labels = planted_labels(adj, feats, classes, rng) if planted else _onehot(n, classes, rng)
    tr = np.arange(n_train, dtype=np.int32)
    va = np.arange(n_train, n_train + n_val, dtype=np.int32)
    te = np.arange(n - n_test, n, dtype=np.int32)
    return n, adj, adj.copy(), feats, None, None, labels, tr, va, te


def cora_like(normalization='gcn', seed=123, planted=False):
    """S-Cora: N=2,708, 5,278 undirected edges, 1,433 sparse features (~18/row), 7 classes."""
    return planetoid_like(2708, 5278, 1433, 18, 7, 140, 500, 1000, normalization, seed, planted)


def pubmed_like(normalization='gcn', seed=123, planted=False):
    """S-PubMed: N=19,717, 44,324 undirected edges, 500 sparse features (~50/row), 3 classes."""
    return planetoid_like(19717, 44324, 500, 50, 3, 60, 500, 1000, normalization, seed, planted)


def rmat_edges(scale_log2, m, rng, abcd=(0.57, 0.19, 0.19, 0.05)):
    """R-MAT directed edge list on 2^scale vertices (S-RMAT): per bit level one uniform draw per
    edge picks the quadrant.  The draws are taken level by level from ``rng`` (that fixes the graph of a seed);
    the per-edge bit arithmetic runs in place on int32/int64 id arrays, in slices on a few host threads (NumPy
    releases the interpreter lock), which is what makes the 10 M / 200 M graph a matter of a minute."""
    from concurrent.futures import ThreadPoolExecutor
    a, b, c, _ = abcd
    dt = np.int32 if scale_log2 <= 30 else np.int64
    src = np.zeros(m, dtype=dt)
    dst = np.zeros(m, dtype=dt)
    nthr = max(1, min(8, os.cpu_count() or 1, m // (1 << 20) + 1))
    cuts = [(m * k) // nthr for k in range(nthr + 1)]

    def work(k, r):
        lo, hi = cuts[k], cuts[k + 1]
        rr, s_, d_ = r[lo:hi], src[lo:hi], dst[lo:hi]
        down = rr >= a + b                                       # quadrants c, d
        right = (rr >= a) & ((rr < a + b) | (rr >= a + b + c))   # quadrants b, d
        np.left_shift(s_, 1, out=s_)
        np.bitwise_or(s_, down, out=s_, casting='unsafe')
        np.left_shift(d_, 1, out=d_)
        np.bitwise_or(d_, right, out=d_, casting='unsafe')

    with ThreadPoolExecutor(nthr) as pool:
        for _bit in range(scale_log2):
            r = rng.random_sample(m)
            list(pool.map(lambda k: work(k, r), range(nthr)))
    return src.astype(np.int64, copy=False), dst.astype(np.int64, copy=False)


def rmat_like(n, m, seed=1):
    """S-RMAT: n vertices (ids folded from the next power of two), m directed edges, values
    U(0,1) row-normalised."""
    rng = np.random.RandomState(seed)
    scale = int(np.ceil(np.log2(n)))
    src, dst = rmat_edges(scale, m, rng)
    src %= n
    dst %= n
    val = rng.random_sample(m).astype(np.float32)
    a = sp.coo_matrix((val, (src, dst)), shape=(n, n)).tocsr()   # sums duplicates
    a.sort_indices()
    return _row_normalize(a)


# SHA-256 of (shape, indptr, indices, data) of the generators' big graphs, computed on the GPU box from a fresh
# ``build()`` (profiles/README.md).  A cache file is builder-writable state outside the repository: a file whose
# arrays do not hash to the digest it carries -- or, for a graph named here, to THIS digest -- is thrown away and rebuilt.
KNOWN_DIGESTS = {
    # rmat_like(10_000_000, 200_000_000, seed=1): 196,949,452 nonzeros after duplicate merging (BASELINE config 5)
    "rmat_10m_200m_seed1": "e47dfff761e3e989122a4e09983d2caebc0a5346aa5097672e02033302cfb773",
}


def graph_digest(a):
    """SHA-256 over a CSR's shape and its three arrays (native byte order, C layout)."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.asarray(a.shape, np.int64).tobytes())
    for x in (a.indptr, a.indices, a.data):
        x = np.ascontiguousarray(x)
        h.update(("%s:%d;" % (x.dtype.str, x.size)).encode())
        h.update(memoryview(x).cast("B"))
    return h.hexdigest()


def cached_graph(name, build):
    """``build()`` (a SciPy CSR), kept as raw arrays under ``$TMPDIR/sgcn_graphs/<name>.npz`` so that the big
    generators (S-RMAT 10 M / 200 M: a minute or two of host time) run once per box, not once per test / bench leg.
    The file carries the SHA-256 of its arrays (``graph_digest``); a file that does not load, whose arrays do not
    hash to that digest, or -- for the graphs listed in ``KNOWN_DIGESTS`` -- to the committed digest, is rebuilt;
    a fresh build of a listed graph that hashes differently raises (the generator has changed)."""
    import tempfile
    d = os.path.join(os.environ.get("TMPDIR") or tempfile.gettempdir(), "sgcn_graphs")
    path = os.path.join(d, name + ".npz")
    want = KNOWN_DIGESTS.get(name)
    try:
        z = np.load(path)
        a = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(int(x) for x in z["shape"]))
        got = graph_digest(a)
        if got == str(z["sha256"]) and want in (None, got):
            return a
    except Exception:
        pass
    a = build().tocsr()
    got = graph_digest(a)
    if want is not None and got != want:
        raise RuntimeError("graph '%s': a fresh build hashes to %s, the committed digest is %s" % (name, got, want))
    try:
        os.makedirs(d, exist_ok=True)
        fd, tmp = tempfile.mkstemp(suffix=".tmp", dir=d)
        with os.fdopen(fd, "wb") as f:
            np.savez(f, data=a.data, indices=a.indices, indptr=a.indptr, shape=np.array(a.shape, np.int64),
                     sha256=np.array(got))
        os.replace(tmp, path)
    except OSError:
        pass
    return a


def load_data(dataset, normalization='gcn', scale=1.0, seed=None):
    """Synthetic counterpart of gcn/utils.py:466-473 ``load_data(dataset)``."""
    # the command-line datasets carry PLANTED labels (a random linear teacher on the 1-hop
    # aggregate) so that a training run has something to learn; the generators' default is random
    # labels, which is all the benchmarks and parity tests need
    if dataset in ('cora', 's-cora'):
        return cora_like(normalization, 123 if seed is None else seed, planted=True)
    if dataset in ('pubmed', 's-pubmed'):
        return pubmed_like(normalization, 123 if seed is None else seed, planted=True)
    if dataset in ('reddit', 's-reddit'):
        n = int(232965 * scale)
        m = int(11600000 * scale)
        sp_ = tuple(int(x * scale) for x in (152410, 23699, 55334))
        return reddit_like(n, m, 602, 41, sp_, 1 if seed is None else seed, planted=True)
    raise ValueError("no synthetic generator for dataset '%s' (real datasets are not on this box)"
                     % dataset)
