"""The LDS-sweep plan (sgcn_ldsplan_*, host arithmetic) without a GPU: the kernel's own operands, decoded the way the
kernel reads them (ring slot of an entry's LDS address -> chunk_cols, register offset -> tile_rows), plus the residual
CSR, are exactly the matrix the plan was built from."""
import numpy as np
import pytest
import scipy.sparse as sp


def _matrix(m, k, dens, seed, long_rows=()):
    rng = np.random.RandomState(seed)
    a = sp.random(m, k, density=dens, format='lil', random_state=rng, dtype=np.float32)
    for r, n in long_rows:
        cols = rng.choice(k, min(n, k), replace=False)
        a[r, cols] = rng.rand(len(cols)).astype(np.float32) + 0.1
    a = a.tocsr()
    a.data[:] = rng.standard_normal(a.nnz).astype(np.float32)
    a.sort_indices()
    return a


def _check(a, labels, min_reuse, T=0, general=True, ring_slots=0):
    from stochastic_gcn_amd import ops
    h = ops.LdsPlanHost(a, labels=labels, min_reuse=min_reuse, T=T, general=general, ring_slots=ring_slots)
    assert (h.S, h.nparts) == ((80, 3) if ring_slots == 80 else (128, 2))
    assert not (general and h.unit)
    r, c, v, s = h.decode()
    if h.unit:                                   # the values live once per row
        assert np.all(v == h.row_fold[r])
    assert r.shape[0] + h.residual.nnz == a.nnz                       # nothing lost, nothing doubled
    assert np.all(r >= 0)
    loc = sp.coo_matrix((v.astype(np.float64), (r, c)), shape=a.shape).tocsr()
    diff = abs((loc + h.residual.astype(np.float64)) - a.astype(np.float64))
    assert diff.nnz == 0 or diff.max() == 0.0
    # a split row's pieces land in consecutive workspace slots of ITS fix record
    fix = {int(f[0]): (int(f[1]), int(f[2])) for f in h.fix}
    for row, slot in zip(r[s >= 0], s[s >= 0]):
        f0, n = fix[int(row)]
        assert f0 <= slot < f0 + n
    assert not np.any(np.isin(r[s < 0], list(fix.keys())))
    return h


@pytest.mark.parametrize("m,k,dens", [(5, 7, 0.5), (100, 300, 0.1), (1000, 1000, 0.02), (3000, 2000, 0.01)])
def test_plan_is_the_matrix(m, k, dens):
    a = _matrix(m, k, dens, m + k, long_rows=[(0, 200)])
    rng = np.random.RandomState(1)
    h1 = _check(a, None, 1)
    assert h1.residual.nnz == 0 and h1.local_nnz == a.nnz
    _check(a, None, 2)
    lab = (rng.randint(0, 3, m).astype(np.int32), rng.randint(0, 3, k).astype(np.int32))
    h = _check(a, lab, 2, T=16)
    assert h.nfix >= 1 and h.ntiles >= 3 or m < 10
    h3 = _check(a, lab, 2, T=16, ring_slots=80)             # three ring parts of 80 slots: more, smaller chunks
    assert h3.nchunks >= h.nchunks and h3.local_nnz == h.local_nnz


def test_one_value_per_column_becomes_a_unit_plan_and_a_row_scale_of_the_operand():
    """The transpose of a row-normalised adjacency (the backward of a mean aggregation) carries one value per COLUMN: the
    plan is built on the pattern (a unit plan, pair words and all), the values become ``col_fold``; planned() + residual
    (x col_fold) is the matrix, exactly."""
    from stochastic_gcn_amd import ops
    a = _matrix(600, 600, 0.05, 11, long_rows=[(3, 300)])
    a.data[:] = 1.0
    deg = np.maximum(np.diff(a.indptr), 1).astype(np.float32)
    a = sp.diags((1.0 / deg).astype(np.float32)).dot(a).tocsr().astype(np.float32)      # D^-1 A: one value per row
    at = a.T.tocsr().astype(np.float32)
    at.sort_indices()
    for lab in (None, np.random.RandomState(2).randint(0, 4, 600).astype(np.int32)):
        h = ops.LdsPlanHost(at, labels=lab, min_reuse=2, T=16)
        assert h.unit and h.col_fold is not None and h.nfix >= 1
        r, c, v, _ = h.planned()
        loc = sp.coo_matrix((v.astype(np.float64), (r, c)), shape=at.shape).tocsr()
        res = h.residual.astype(np.float64).dot(sp.diags(h.col_fold.astype(np.float64)))
        diff = abs(loc + res - at.astype(np.float64))
        assert r.shape[0] + h.residual.nnz == at.nnz and (diff.nnz == 0 or diff.max() == 0.0)
        assert ops.LdsPlanHost(a, labels=lab, min_reuse=2).col_fold is None          # one value per ROW: the ordinary unit plan
    g = ops.LdsPlanHost(_matrix(200, 200, 0.1, 5), min_reuse=2)
    assert g.col_fold is None and not g.unit                                          # arbitrary values: a general plan


def test_plan_edge_cases():
    _check(sp.csr_matrix((10, 10), dtype=np.float32), None, 1)        # no nonzeros at all
    _check(sp.csr_matrix((0, 5), dtype=np.float32), None, 1)          # no rows
    a = _matrix(50, 40, 0.3, 3)
    a[7] = 0                                                          # an empty row in the middle
    a = a.tocsr()
    a.eliminate_zeros()
    _check(a, None, 3)


def test_plan_rejects_bad_input():
    from stochastic_gcn_amd import ops, _ffi
    a = _matrix(20, 20, 0.3, 5)
    with pytest.raises(_ffi.SgcnError):
        ops.LdsPlanHost(a, labels=(None, np.zeros(20, np.int32) - 1), min_reuse=1) if False else \
            ops.LdsPlanHost(a, labels=(np.full(20, -1, np.int32), None), min_reuse=1)
    with pytest.raises(ValueError):
        ops.LdsPlanHost(a, labels=(None, np.zeros(19, np.int32)))


def test_unit_plans_fold_row_constant_values():
    """A row-normalised adjacency (the reference's D^-1 A, gcn/utils.py:299-309) gives a unit plan: one value per row;
    a matrix whose rows mix values does not, and `general=True` never folds."""
    rng = np.random.RandomState(3)
    a = _matrix(400, 300, 0.1, 9)
    a.data[:] = 1.0
    deg = np.maximum(np.diff(a.indptr), 1)
    a = sp.diags((1.0 / deg).astype(np.float32)).dot(a).tocsr().astype(np.float32)
    h = _check(a, None, 2, general=False)
    assert h.unit == 1
    assert _check(a, None, 2, general=True).unit == 0
    b = _matrix(400, 300, 0.1, 9)
    assert _check(b, None, 2, general=False).unit == 0
    assert np.all(_check(b, None, 2, general=False).row_fold == 1.0)


def test_plan_on_a_graph_with_communities_stages_shared_columns():
    from stochastic_gcn_amd import synthetic
    data = synthetic.reddit_sbm(n=20000, m=800000, classes=5, splits=(15000, 2000, 3000), p_in=0.8, seed=2)
    a = data[2]
    comm = data[6].argmax(1).astype(np.int32)
    h = _check(a, comm, 2, general=False)
    assert h.unit == 1                               # row-normalised: values fold into the row scale
    assert h.local_nnz / a.nnz > 0.75                # the in-community nonzeros ride the ring ...
    assert h.local_nnz / h.staged > 4.0              # ... with real reuse per staged piece
    flat = _check(a, None, 2)                        # without labels the same matrix shares far less
    assert flat.local_nnz / max(flat.staged, 1) < h.local_nnz / h.staged


def test_a_small_residual_is_folded_into_the_plan():
    """LdsSweepCSR.auto_host: min_reuse 3 leaves the columns a tile uses once or twice to the residual sweep; when that is under
    16 % of the nonzeros every column is staged instead (min_reuse 1: no residual at all, one kernel) -- and not otherwise."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(0)
    n, nb = 3000, 3
    lab = np.repeat(np.arange(nb), n // nb).astype(np.int32)
    dense_blocks = sp.block_diag([sp.random(n // nb, n // nb, density=0.05, format='csr', random_state=rng, dtype=np.float32)
                                  for _ in range(nb)]).tocsr()
    for wide_nnz_per_row, want_all in ((0.5, True), (20, False)):
        # out-of-block nonzeros over 200,000 extra columns: nearly all of them the only one of their column in a tile
        m = int(wide_nnz_per_row * n)
        wide = sp.coo_matrix((np.ones(m, np.float32), (rng.randint(0, n, m), rng.randint(0, 40000, m))), shape=(n, 40000)).tocsr()
        a = sp.hstack((dense_blocks, wide)).tocsr()
        a.data[:] = 1.0
        a.sort_indices()
        labels = (lab, np.concatenate([lab, np.full(40000, nb, np.int32)]))
        h3 = ops.LdsPlanHost(a, labels=labels, min_reuse=3)
        h = ops.LdsSweepCSR.auto_host(a, labels)
        small = 0 < h3.residual.nnz < ops.LdsSweepCSR.ALL_STAGED_BELOW * a.nnz
        assert small == want_all, (h3.residual.nnz, a.nnz)
        assert (h.residual.nnz == 0 and h.local_nnz == a.nnz) if want_all else (h.residual.nnz == h3.residual.nnz > 0)
