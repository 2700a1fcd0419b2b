"""The CPU oracle (oracle/oracle_c.c, oracle/oracle_np.py) against independent references:
scipy.sparse (what the reference itself uses for the PP product, gcn/utils.py:321-322),
float64 NumPy, the algebraic identities of SURVEY.md §8c, and the golden vectors of the
reference's row slicers (tests/golden/slice.npz)."""
import numpy as np
import scipy.sparse as sp

import golden_util as gu
from oracle import oracle_np as onp


def _rand_csr(m, k, density, seed):
    rng = np.random.RandomState(seed)
    a = sp.random(m, k, density=density, format='csr', random_state=rng, dtype=np.float32)
    a.sort_indices()
    return a


def test_spmm_c_vs_scipy_and_f64():
    for (m, k, d, dens) in [(50, 40, 7, 0.2), (300, 500, 128, 0.05), (64, 64, 602, 0.1), (5, 9, 1, 0.5)]:
        a = _rand_csr(m, k, dens, m)
        b = np.random.RandomState(1).standard_normal((k, d)).astype(np.float32)
        got = onp.spmm(a.indptr, a.indices, a.data, b)
        assert onp.rel_err(got, a.dot(b)) < 1e-5
        assert onp.rel_err(got, onp.spmm_f64(a.indptr, a.indices, a.data, b)) < 1e-5


def test_spmm_options():
    a = _rand_csr(30, 20, 0.3, 0)
    rng = np.random.RandomState(2)
    H = rng.standard_normal((100, 16)).astype(np.float32)
    g = rng.choice(100, 20, replace=False).astype(np.int32)
    rs = rng.rand(30).astype(np.float32)
    cs = rng.rand(20).astype(np.float32)
    c0 = rng.standard_normal((30, 16)).astype(np.float32)
    got = onp.spmm(a.indptr, a.indices, a.data, H, gidx=g, rscale=rs, cscale=cs, C_in=c0, beta=0.5)
    ref = (sp.diags(rs) @ a @ sp.diags(cs)).dot(H[g].astype(np.float64)) + 0.5 * c0
    assert onp.rel_err(got, ref) < 1e-5


def test_empty_rows_and_empty_matrix():
    a = sp.csr_matrix((4, 6), dtype=np.float32)
    b = np.ones((6, 3), np.float32)
    np.testing.assert_array_equal(onp.spmm(a.indptr, a.indices, a.data, b), np.zeros((4, 3), np.float32))


def test_coo_to_csr_keeps_stored_order():
    idx = np.array([[0, 3], [0, 1], [2, 2], [2, 0]], dtype=np.int32)
    w = np.array([1, 2, 3, 4], dtype=np.float32)
    m = onp.coo_to_csr((idx, w, (3, 4)))
    assert m.indptr.tolist() == [0, 2, 2, 4]
    assert m.indices.tolist() == [3, 1, 2, 0] and m.data.tolist() == [1, 2, 3, 4]


def _vr_case(seed, cvd, concat, n=400, d=16):
    from oracle import sampler as osamp  # noqa: F401  (pure-python sampler is slow; use product-free path)
    rng = np.random.RandomState(seed)
    adj = _rand_csr(n, n, 0.02, seed)
    adj = sp.diags(1.0 / (np.array(adj.sum(1)).ravel() + 1e-20)).dot(adj).tocsr().astype(np.float32)
    n1 = 20
    rows = rng.choice(n, n1, replace=False)
    full = adj[rows].tocsr()
    ffield, fcol = np.unique(full.indices, return_inverse=True)
    fadj = sp.csr_matrix((full.data, fcol.astype(np.int32), full.indptr), shape=(n1, len(ffield)))
    return rng, adj, rows, fadj, ffield.astype(np.int32)


def test_vr_identities():
    """with a fresh history (Hbar[ifield] == mu) the sampled term
    vanishes and mu_nbr is the exact full-neighbour aggregate; with dropout off (h == mu)
    also h_nbr == mu_nbr; degree >= max degree makes NS == CV == exact."""
    rng, adj, rows, fadj, ffield = _vr_case(0, True, False)
    n, d = adj.shape[0], 16
    act = rng.standard_normal((n, d)).astype(np.float32)        # true activations of all vertices
    Hbar = act.copy()                                            # fresh history
    # sampled adjacency: 1 neighbour per row, amplified by degree
    ifield = list(rows)
    pos = {v: i for i, v in enumerate(ifield)}
    er, ec, ew = [], [], []
    for i, r in enumerate(rows):
        lo, hi = adj.indptr[r], adj.indptr[r + 1]
        if hi > lo:
            c = adj.indices[lo]
            if c not in pos:
                pos[c] = len(ifield)
                ifield.append(c)
            er.append(i); ec.append(pos[c]); ew.append(adj.data[lo] * (hi - lo))
    ifield = np.array(ifield, dtype=np.int32)
    A = onp.coo_to_csr((np.stack([er, ec], 1).astype(np.int32), np.float32(ew), (len(rows), len(ifield))))
    mu = act[ifield]
    scale = np.ones(len(rows), np.float32)
    h_nbr, mu_nbr, new_hist = onp.vr_aggregate(A, fadj, mu, mu, Hbar, ifield, ffield, scale, True, False)
    exact = adj[rows].dot(act.astype(np.float64))
    assert onp.rel_err(mu_nbr, exact) < 1e-5
    assert onp.rel_err(h_nbr, mu_nbr) < 1e-6
    assert new_hist[0] is mu or np.array_equal(new_hist[0], mu)
    # plain CV with fresh history: A x - A Hbar[ifield] + P Hbar[ffield] == exact
    out, _, _ = onp.vr_aggregate(A, fadj, mu, None, Hbar, ifield, ffield, None, False, False)
    assert onp.rel_err(out, exact) < 1e-5
    # concat puts the self rows first
    out2, _, _ = onp.vr_aggregate(A, fadj, mu, None, Hbar, ifield, ffield, None, False, True)
    np.testing.assert_array_equal(out2[:, :d], mu[:len(rows)])
    np.testing.assert_array_equal(out2[:, d:], out)


def test_slicers_golden():
    z = gu.load("slice.npz")
    a = sp.csr_matrix((z["a/data"], z["a/indices"], z["a/indptr"]), shape=tuple(z["a/shape"]))
    names = sorted({k.split("/")[1] for k in z.files if k.startswith("slice/")})
    assert "empty_only" in names or any(("slice/%s/is_empty_csr" % n) in z.files for n in names)
    for n in names:
        r = z["slice/%s/r" % n]
        res = onp.csr_slice(a, r)
        if ("slice/%s/is_empty_csr" % n) in z.files:
            assert sp.issparse(res) and res.shape == tuple(z["slice/%s/is_empty_csr" % n])
        else:
            assert gu.bits_equal(res[0], z["slice/%s/indices" % n])
            assert gu.bits_equal(res[1], z["slice/%s/data" % n])
            assert gu.bits_equal(res[2], z["slice/%s/shape" % n])
        assert gu.bits_equal(onp.gather_rows(z["dense/a"], r), z["dense/%s/out" % n])


def test_host_slice_indptr_golden():
    """sgcn_csr_slice_indptr (host half of the product's row slice) vs the reference c_indptr."""
    import ctypes as C  # noqa: F401
    from stochastic_gcn_amd._ffi import lib, check
    z = gu.load("slice.npz")
    ap = np.ascontiguousarray(z["a/indptr"], dtype=np.int32)
    for n in sorted({k.split("/")[1] for k in z.files if k.startswith("slice/")}):
        r = np.ascontiguousarray(z["slice/%s/r" % n], dtype=np.int32)
        op = np.empty(len(r) + 1, dtype=np.int32)
        check(lib.sgcn_csr_slice_indptr(len(r), r.ctypes.data, ap.ctypes.data, op.ctypes.data))
        want = np.concatenate([[0], np.cumsum(np.diff(ap)[r])]).astype(np.int32)
        np.testing.assert_array_equal(op, want)
        if ("slice/%s/indices" % n) in z.files:
            rows = z["slice/%s/indices" % n][:, 0]
            np.testing.assert_array_equal(np.repeat(np.arange(len(r)), np.diff(op)), rows)
