"""Host-side column-sweep plan and graph reordering (csrc/sgcn_csplan.cpp): every nonzero lands in
exactly one tile slot with its row and value, grouped plans keep tiles inside their row group,
and label propagation finds planted communities (and nothing in a graph without structure)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from stochastic_gcn_amd import synthetic
from stochastic_gcn_amd._ffi import check, lib


def _plan(a, R=16, T=0, row_group=None):
    a = a.tocsr()
    rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
    col = np.ascontiguousarray(a.indices, dtype=np.int32)
    val = np.ascontiguousarray(a.data, dtype=np.float32)
    M = a.shape[0]
    rg = None if row_group is None else np.ascontiguousarray(row_group, dtype=np.int32)
    rgp = None if rg is None else rg.ctypes.data
    nt, nf, ns = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.sgcn_csplan_count(rowptr.ctypes.data, M, R, T, rgp, C.byref(nt), C.byref(nf), C.byref(ns)))
    tp = np.empty(nt.value + 1, np.int64)
    cr = np.empty(a.nnz, np.int32)
    vo = np.empty(a.nnz, np.float32)
    tr = np.empty(nt.value * R, np.int32)
    ts = np.empty(nt.value * R, np.int32)
    fx = np.empty((nf.value, 3), np.int32)
    check(lib.sgcn_csplan_fill(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, R, T, rgp, tp.ctypes.data,
                               cr.ctypes.data, vo.ctypes.data, tr.ctypes.data, ts.ctypes.data,
                               fx.ctypes.data if nf.value else None))
    return dict(nt=nt.value, nfix=nf.value, nslots=ns.value, tile_ptr=tp, colrow=cr, val=vo, rows=tr.reshape(-1, R),
                slots=ts.reshape(-1, R), fix=fx, R=R)


def _rebuild(p, shape):
    """The matrix a plan encodes (rows summed over their virtual pieces)."""
    shift = 28 if p['R'] <= 16 else 27
    cr = p['colrow'].view(np.uint32)
    col = (cr & ((1 << shift) - 1)).astype(np.int64)
    lr = (cr >> shift).astype(np.int64)
    tile = np.repeat(np.arange(p['nt']), np.diff(p['tile_ptr']))
    row = p['rows'][tile, lr]
    assert (row >= 0).all()
    return sp.coo_matrix((p['val'], (row, col)), shape=shape).tocsr()


@pytest.mark.parametrize("grouped", [False, True])
def test_plan_encodes_the_matrix_exactly(grouped):
    rng = np.random.RandomState(3)
    a = sp.random(700, 500, density=0.05, random_state=rng, format='csr', dtype=np.float32)
    a = sp.vstack([a, sp.csr_matrix(np.ones((1, 500), np.float32))]).tocsr()     # one very long row -> split
    a.sort_indices()
    g = rng.randint(0, 7, a.shape[0]) if grouped else None
    p = _plan(a, T=64, row_group=g)
    assert p['nfix'] >= 1 and p['tile_ptr'][-1] == a.nnz
    b = _rebuild(p, a.shape)
    assert (abs(a - b)).nnz == 0
    # column-sorted inside every tile
    shift = 28
    for t in range(p['nt']):
        c = p['colrow'].view(np.uint32)[p['tile_ptr'][t]:p['tile_ptr'][t + 1]] & ((1 << shift) - 1)
        assert (np.diff(c.astype(np.int64)) >= 0).all()
    if grouped:     # a tile's rows belong to ONE group, groups appear in label order
        tg = []
        for t in range(p['nt']):
            rows = p['rows'][t][p['rows'][t] >= 0]
            assert len(set(g[rows].tolist())) == 1
            tg.append(int(g[rows[0]]))
        assert tg == sorted(tg)
    # split rows: slots are consecutive per row, in row order
    for r, first, cnt in p['fix']:
        got = sorted(p['slots'][p['rows'] == r].tolist())
        assert got == list(range(first, first + cnt))


def test_negative_group_label_is_rejected():
    a = sp.identity(8, format='csr', dtype=np.float32)
    g = np.array([0, 1, -1, 0, 0, 0, 0, 0], np.int32)
    nt, nf, ns = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib.sgcn_csplan_count(np.ascontiguousarray(a.indptr, np.int32).ctypes.data, 8, 16, 0, g.ctypes.data,
                               C.byref(nt), C.byref(nf), C.byref(ns))
    assert rc == -1 and b"negative group" in lib.sgcn_last_error()


def test_label_propagation_finds_planted_communities_and_nothing_else():
    from stochastic_gcn_amd import ops
    d = synthetic.reddit_sbm(n=12000, m=500000, classes=12, splits=(8000, 1000, 3000), p_in=0.85, seed=4)
    truth = d[6].argmax(1)
    comm, nc = ops.reorder_labels(d[2], seed=1)
    assert 6 <= nc <= 40
    agree = sum(np.bincount(truth[comm == c]).max() for c in range(nc) if (comm == c).any())
    assert agree / len(truth) > 0.8, agree / len(truth)
    assert (np.diff(np.bincount(comm, minlength=nc)[:nc - 1]) <= 0).all()      # numbered by decreasing size
    comm2, nc2 = ops.reorder_labels(d[2], seed=1)
    assert nc2 == nc and np.array_equal(comm, comm2)                            # deterministic
    u = synthetic.reddit_like(n=12000, m=500000, splits=(8000, 1000, 3000), seed=4, with_features=False)
    _, ncu = ops.reorder_labels(u[2])
    assert ncu == 1          # uniform destinations: no structure, the reordering degenerates to a no-op


@pytest.mark.parametrize("align", [0, 40])
def test_two_lane_group_plan_encodes_the_matrix_exactly(align):
    """sgcn_csplang_* (two lane groups): interleaved entries (2*step + bin), pads marked by the value bits 0x80000000, 32 row slots
    per tile, whole launches; with align > 0 the two bins of a tile stay within `align` columns of each other."""
    rng = np.random.RandomState(5)
    a = sp.random(900, 700, density=0.04, random_state=rng, format='csr', dtype=np.float32)
    a = sp.vstack([a, sp.csr_matrix(np.ones((1, 700), np.float32))]).tocsr()
    a.sort_indices()
    a.data[3] = -0.0                                      # a real -0.0f must not be taken for a pad
    rowptr = np.ascontiguousarray(a.indptr, np.int32)
    col, val = np.ascontiguousarray(a.indices, np.int32), np.ascontiguousarray(a.data, np.float32)
    M = a.shape[0]
    for rnd in (0, 8):
        nt, ne, nf, ns = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.sgcn_csplang_count(rowptr.ctypes.data, col.ctypes.data, M, 64, rnd, align, 2, None, 0, C.byref(nt), C.byref(ne),
                                     C.byref(nf), C.byref(ns)))
        if rnd:
            assert nt.value % rnd == 0
        tp = np.empty(nt.value + 1, np.int64)
        cr, vo = np.empty(ne.value, np.int32), np.empty(ne.value, np.float32)
        tr, ts = np.empty(nt.value * 32, np.int32), np.empty(nt.value * 32, np.int32)
        fx = np.empty((nf.value, 3), np.int32)
        check(lib.sgcn_csplang_fill(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, 64, rnd, align, 2, None, 0, tp.ctypes.data,
                                    cr.ctypes.data, vo.ctypes.data, tr.ctypes.data, ts.ctypes.data, fx.ctypes.data))
        assert tp[-1] == ne.value and nf.value >= 1 and (np.diff(tp) % 2 == 0).all()
        u = cr.view(np.uint32)
        real = vo.view(np.uint32) != 0x80000000
        assert real.sum() == a.nnz
        e = np.arange(ne.value)
        tile = np.repeat(np.arange(nt.value), np.diff(tp))
        g = (e - tp[tile]) % 2
        cols = (u & 0x0FFFFFFF).astype(np.int64)
        rows = tr[tile * 32 + g * 16 + (u >> 28).astype(np.int64)]
        assert (rows[real] >= 0).all()
        b = sp.coo_matrix((vo[real], (rows[real], cols[real])), shape=a.shape).tocsr()
        assert (abs(a - b)).nnz == 0
        for t in range(nt.value):
            for gg in range(2):
                sel = (tile == t) & (g == gg) & real
                assert (np.diff(cols[sel]) >= 0).all()                   # column-sorted inside a bin
            if align:                                                     # the halves of a step stay close
                s0, s1 = (tile == t) & (g == 0), (tile == t) & (g == 1)
                both = real[s0] & real[s1]
                assert (np.abs(cols[s0][both] - cols[s1][both]) <= align).all()


def _skewed_csr(M, K, nnz, seed):
    """columns drawn with an R-MAT-like skew: a third of the nonzeros in the first sixteenth of the ids"""
    rng = np.random.RandomState(seed)
    bits = max(int(np.ceil(np.log2(K))), 1)
    c = np.zeros(nnz, np.int64)
    for _ in range(bits):
        c = (c << 1) | (rng.random_sample(nnz) >= 0.76)
    c %= K
    r = rng.randint(0, M, nnz)
    a = sp.coo_matrix((rng.rand(nnz).astype(np.float32) + 0.1, (r, c)), shape=(M, K)).tocsr()
    a.sort_indices()
    return a


def test_warp_table_is_the_cumulative_share_of_the_work():
    """ops.ColumnSweepCSR.make_warp: entry b = share of the nonzeros in columns < b << shift, scaled to [0, K); 'auto' keeps
    the linear clock for evenly spread columns and builds a table for skewed ones; the table never exceeds its bound."""
    from stochastic_gcn_amd import ops
    a = _skewed_csr(3000, 70000, 400000, 1)
    tab, sh = ops.ColumnSweepCSR.make_warp(a.indices, a.shape[1])
    assert tab is not None and tab.dtype == np.uint32
    nb = ((a.shape[1] - 1) >> sh) + 1
    assert tab.shape[0] == nb <= ops.ColumnSweepCSR.WARP_BUCKETS < 2 * nb
    assert tab[0] == 0 and (np.diff(tab.astype(np.int64)) >= 0).all() and tab[-1] < a.shape[1]
    hist = np.bincount(a.indices >> sh, minlength=nb)
    before = np.concatenate([[0], np.cumsum(hist)[:-1]])
    np.testing.assert_array_equal(tab, np.minimum(np.floor(before / a.nnz * a.shape[1]), a.shape[1] - 1).astype(np.uint32))
    assert tab[nb // 16] > 0.2 * a.shape[1]                   # the skew: the first sixteenth of the ids is > 20 % of the sweep
    rng = np.random.RandomState(0)
    even = rng.randint(0, 70000, 400000)
    assert ops.ColumnSweepCSR.make_warp(even, 70000) == (None, 0)
    assert ops.ColumnSweepCSR.make_warp(even, 70000, True)[0] is not None
    assert ops.ColumnSweepCSR.make_warp(a.indices, a.shape[1], False) == (None, 0)
    assert ops.ColumnSweepCSR.make_warp(np.zeros(0, np.int32), 10) == (None, 0)


@pytest.mark.parametrize("G", [2, 4])
def test_lane_group_plan_aligns_its_bins_in_sweep_positions(G):
    """sgcn_csplang_* with a warp table: the plan still encodes the matrix exactly (every nonzero once, in its row, columns
    ascending per bin), pads gather from a column of the tile, and the bins of a tile stay within `align` POSITIONS -- with
    far fewer pads than the same bound in column ids costs on a skewed matrix."""
    from stochastic_gcn_amd import ops
    a = _skewed_csr(2000, 60000, 60000, 2)
    rowptr = np.ascontiguousarray(a.indptr, np.int32)
    col, val = np.ascontiguousarray(a.indices, np.int32), np.ascontiguousarray(a.data, np.float32)
    M, K = a.shape
    tab, sh = ops.ColumnSweepCSR.make_warp(col, K, True)
    align = 3000

    def build(warp):
        wp = tab.ctypes.data if warp else None
        nt, ne, nf, ns = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.sgcn_csplang_count(rowptr.ctypes.data, col.ctypes.data, M, 0, 0, align, G, wp, sh, C.byref(nt), C.byref(ne),
                                     C.byref(nf), C.byref(ns)))
        tp = np.empty(nt.value + 1, np.int64)
        cr, vo = np.empty(ne.value, np.int32), np.empty(ne.value, np.float32)
        tr, ts = np.empty(nt.value * 16 * G, np.int32), np.empty(nt.value * 16 * G, np.int32)
        fx = np.empty((max(nf.value, 1), 3), np.int32)
        check(lib.sgcn_csplang_fill(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, 0, 0, align, G, wp, sh,
                                    tp.ctypes.data, cr.ctypes.data, vo.ctypes.data, tr.ctypes.data, ts.ctypes.data, fx.ctypes.data))
        return nt.value, ne.value, tp, cr.view(np.uint32), vo, tr

    nt, ne, tp, u, vo, tr = build(True)
    real = vo.view(np.uint32) != 0x80000000
    assert real.sum() == a.nnz
    e = np.arange(ne)
    tile = np.repeat(np.arange(nt), np.diff(tp))
    g = (e - tp[tile]) % G
    cols = (u & 0x0FFFFFFF).astype(np.int64)
    rows = tr[tile * 16 * G + g * 16 + (u >> 28).astype(np.int64)]
    got = sp.coo_matrix((vo[real], (rows[real], cols[real])), shape=a.shape).tocsr()
    got.sort_indices()
    assert (got != a).nnz == 0
    pos = tab[cols >> sh].astype(np.int64)
    for t in range(nt):
        sl = slice(tp[t], tp[t + 1])
        steps = (e[sl] - tp[t]) // G
        for b in range(G):
            m = real[sl] & (g[sl] == b)
            assert (np.diff(cols[sl][m]) >= 0).all()
        # within a step, the entries that are applied lie within `align` positions of the slowest bin's
        p_t, r_t = pos[sl], real[sl]
        lo = np.full(steps.max() + 1, np.iinfo(np.int64).max)
        np.minimum.at(lo, steps[r_t], p_t[r_t])
        assert (p_t[r_t] <= lo[steps[r_t]] + align).all()
    ne_cols = build(False)[1]
    assert ne < ne_cols                                       # the same bound in column ids pads more


def test_auto_align_is_a_third_of_the_l2_window():
    from stochastic_gcn_amd import ops
    f = ops.ColumnSweepCSR.auto_align
    assert f(232965, 23173306, 232965, 2, 4096) == 2048       # S-Reddit: the measured default stays
    assert 60000 < f(10_000_000, 24_613_381, 1_053_273, 2, 4096) < 120000      # one GPU's block of S-RMAT 10 M
    assert f(1000, 10, 50, 2, 4096) <= max(2048, 1000 // 8) and f(1000, 10, 50, 2, 4096) >= 2048


def test_auto_t_fills_the_rounds_a_small_block_needs_and_choose_g_counts_rounds():
    """ops.ColumnSweepCSR.auto_t: a block with few rows splits them further (T down from the library's 4 x mean degree) until
    its virtual rows fill 70 % of the round it occupies anyway; a matrix that fills its rounds keeps the default (0).
    choose_g compares passes x ROUNDS: a block that fits one round of single-group tiles takes one group."""
    from stochastic_gcn_amd import ops
    Cs = ops.ColumnSweepCSR
    rng = np.random.RandomState(0)
    deg = rng.poisson(100, 29000)
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    for G in (1, 2):
        T = Cs.auto_t(rowptr, G, 4096)
        cap = 4096 * 16 * G
        v = lambda t: int(np.maximum(1, -(-deg // t)).sum())       # noqa: E731
        assert 24 <= T < 400 and v(T) <= 0.7 * cap and (T == 24 or v(T - 1) > 0.7 * cap)
    big = np.concatenate([[0], np.cumsum(rng.poisson(100, 60000))])
    assert Cs.auto_t(big, 1, 4096) == 0                            # 60 k of 65 k slots: the library default stays
    assert Cs.auto_t(np.zeros(1, np.int64), 1, 4096) == 0
    assert Cs.choose_g(602, 100, 29000) == 1 and Cs.choose_g(602, 100, 58000) == 1
    assert Cs.choose_g(602, 100, 116000) == 2 and Cs.choose_g(602, 100, 232965) == 2
    assert Cs.choose_g(256, 23, 1053273) == 4 and Cs.choose_g(256, 334, 73793) == 2 and Cs.choose_g(602, 490, 232965) == 2


def test_warp_table_edge_cases():
    """one column, every nonzero in one column, fewer columns than buckets, the largest column id: the table has one entry
    per bucket, stays below K and never decreases; the plans built with it still encode the matrix."""
    from stochastic_gcn_amd import ops
    Cs = ops.ColumnSweepCSR
    tab, sh = Cs.make_warp(np.zeros(5, np.int32), 1, True)
    assert sh == 0 and tab.tolist() == [0]
    tab, sh = Cs.make_warp(np.full(1000, 7, np.int32), 50, True)
    assert sh == 0 and tab.shape[0] == 50 and tab[7] == 0 and (tab[8:] == 49).all()       # everything in front of column 8
    K = 3 * Cs.WARP_BUCKETS + 5
    cols = np.array([0, K - 1, K - 1, 17], np.int32)
    tab, sh = Cs.make_warp(cols, K, True)
    assert sh == 2 and tab.shape[0] == ((K - 1) >> 2) + 1 and tab[-1] < K and (np.diff(tab.astype(np.int64)) >= 0).all()
    a = sp.csr_matrix((np.ones(4, np.float32), (np.array([0, 1, 2, 3]), cols)), shape=(4, K))
    for G in (1, 2, 4):
        A = Cs(a, 'cpu', G=G, warp=True)
        assert A.warp is not None and A.warp.numel() == tab.shape[0] and A.nnz >= 4


def _build(a, G, threads, T=0, rnd=0, align=0, R=16, row_group=None, warp=None, shift=0):
    """sgcn_csplan_build + export on `threads` host threads -> the plan's arrays"""
    a = a.tocsr()
    rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
    col = np.ascontiguousarray(a.indices, dtype=np.int32)
    val = np.ascontiguousarray(a.data, dtype=np.float32)
    rg = None if row_group is None else np.ascontiguousarray(row_group, dtype=np.int32)
    h = C.c_void_p()
    check(lib.sgcn_csplan_build(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, a.shape[0], G, R, T, rnd, align,
                                None if rg is None else rg.ctypes.data, None if warp is None else warp.ctypes.data, shift,
                                threads, C.byref(h)))
    try:
        nt, ne, nf, ns, tu, th = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        check(lib.sgcn_csbuild_sizes(h, C.byref(nt), C.byref(ne), C.byref(nf), C.byref(ns), C.byref(tu), C.byref(th)))
        assert th.value == threads and tu.value > 0
        out = [np.empty(nt.value + 1, np.int64), np.empty(ne.value, np.int32), np.empty(ne.value, np.float32),
               np.empty(nt.value * R * G, np.int32), np.empty(nt.value * R * G, np.int32), np.empty((nf.value, 3), np.int32)]
        check(lib.sgcn_csbuild_export(h, *[x.ctypes.data if x.size else None for x in out]))
    finally:
        lib.sgcn_csbuild_free(h)
    return out + [np.array([ns.value])]


def _unsorted_with_duplicates(seed):
    """a CSR whose rows are NOT column-sorted and hold repeated (row, column) entries"""
    rng = np.random.RandomState(seed)
    coo = sp.random(3000, 2000, density=0.02, random_state=rng, format='coo', dtype=np.float32)
    r = np.concatenate([coo.row, coo.row[:500]])
    c = np.concatenate([coo.col, coo.col[:500]])
    v = np.concatenate([coo.data, coo.data[:500] * 2])
    perm = rng.permutation(len(r))
    order = np.argsort(r[perm], kind='stable')
    r, c, v = r[perm][order], c[perm][order], v[perm][order]
    ip = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=3000))]).astype(np.int32)
    return sp.csr_matrix((v, c.astype(np.int32), ip), shape=(3000, 2000))


def test_parallel_plan_is_bit_identical_to_the_serial_one_and_to_the_two_call_form():
    """sgcn_csplan_build (ABI v14): the plan does not depend on the number of building threads, equals what the count / fill
    pair returns, handles rows that are not column-sorted (stable: duplicates keep their order), and a hub row's pieces."""
    from stochastic_gcn_amd import ops
    mats = [_skewed_csr(20000, 30000, 400000, 5), _unsorted_with_duplicates(2)]
    a = sp.random(700, 500, density=0.05, random_state=np.random.RandomState(3), format='csr', dtype=np.float32)
    a = sp.vstack([a, sp.csr_matrix(np.ones((1, 500), np.float32))]).tocsr()
    a.sort_indices()
    mats.append(a)
    for a in mats:
        rg = np.random.RandomState(1).randint(0, 5, a.shape[0])
        for T in (0, 64):
            for kw in (dict(G=1), dict(G=1, row_group=rg), dict(G=1, R=32), dict(G=2, rnd=64, align=500), dict(G=4, rnd=64, align=2048),
                       dict(G=2, align=0)):
                one = _build(a, threads=1, T=T, **kw)
                for threads in (3, 8):
                    many = _build(a, threads=threads, T=T, **kw)
                    assert all(np.array_equal(x.view(np.int32) if x.dtype == np.float32 else x,
                                              y.view(np.int32) if y.dtype == np.float32 else y) for x, y in zip(one, many)), (T, kw)
            # the two-call form of the header is the same plan
            p = _plan(a, T=T, row_group=rg)
            one = _build(a, G=1, threads=2, T=T, row_group=rg)
            assert np.array_equal(p['tile_ptr'], one[0]) and np.array_equal(p['colrow'], one[1])
            assert np.array_equal(p['val'].view(np.int32), one[2].view(np.int32)) and np.array_equal(p['rows'].ravel(), one[3])
        wt, sh = ops.ColumnSweepCSR.make_warp(a.indices, a.shape[1], True)
        one = _build(a, G=2, threads=1, rnd=64, align=300, warp=wt, shift=sh)
        many = _build(a, G=2, threads=5, rnd=64, align=300, warp=wt, shift=sh)
        assert all(np.array_equal(x, y) for x, y in zip(one[:2], many[:2]))
    # the encoded matrix of an unsorted input: exactly the input (duplicates summed by both sides)
    a = mats[1]
    one = _build(a, G=1, threads=4, T=64)
    p = dict(nt=len(one[0]) - 1, tile_ptr=one[0], colrow=one[1], val=one[2], rows=one[3].reshape(-1, 16), R=16)
    b = _rebuild(p, a.shape)
    ac = a.copy()
    ac.sum_duplicates()
    assert abs(ac - b).max() < 1e-6


def test_build_rejects_bad_arguments():
    a = sp.identity(8, format='csr', dtype=np.float32)
    rp, col, val = (np.ascontiguousarray(x) for x in (a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data))
    h = C.c_void_p()
    assert lib.sgcn_csplan_build(rp.ctypes.data, col.ctypes.data, val.ctypes.data, 8, 3, 16, 0, 0, 0, None, None, 0, 1, C.byref(h)) == -1
    g = np.zeros(8, np.int32)
    assert lib.sgcn_csplan_build(rp.ctypes.data, col.ctypes.data, val.ctypes.data, 8, 2, 16, 0, 0, 0, g.ctypes.data, None, 0, 1, C.byref(h)) == -1
    bad = col.copy()
    bad[3] = -1
    assert lib.sgcn_csplan_build(rp.ctypes.data, bad.ctypes.data, val.ctypes.data, 8, 1, 16, 0, 0, 0, None, None, 0, 2, C.byref(h)) == -1
    assert b"does not fit" in lib.sgcn_last_error() and not h.value
    assert lib.sgcn_host_threads() >= 1


def test_host_transpose_equals_scipy():
    """sgcn_csr_transpose_host: the stable parallel counting sort gives SciPy's csr -> csc arrays, whatever the thread count;
    rectangular matrices, empty rows / columns, duplicates."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(7)
    for a in (_skewed_csr(5000, 9000, 300000, 3), _unsorted_with_duplicates(4),
              sp.csr_matrix((50, 70), dtype=np.float32), sp.random(400, 30, density=0.3, random_state=rng, format='csr', dtype=np.float32)):
        want = a.T.tocsr()
        for threads in (1, 2, 7):
            got = ops.transpose_host(a, threads=threads)
            assert got.shape == want.shape
            assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
            assert np.array_equal(got.data.view(np.int32), want.data.astype(np.float32).view(np.int32))
    bad = sp.csr_matrix((np.ones(2, np.float32), np.array([0, 9], np.int32), np.array([0, 2], np.int32)), shape=(1, 10))
    bad.indices[1] = 11
    with pytest.raises(RuntimeError, match="outside"):
        ops.transpose_host(bad)


def test_warp_table_in_c_equals_the_numpy_expression():
    """sgcn_cs_warp_table against the NumPy form ops.ColumnSweepCSR.make_warp had until round 5 (float64, same order)."""
    from stochastic_gcn_amd import ops

    def numpy_form(cols, K, mode):
        shift = 0
        while ((K - 1) >> shift) + 1 > ops.ColumnSweepCSR.WARP_BUCKETS:
            shift += 1
        nb = ((K - 1) >> shift) + 1
        hist = np.bincount(np.asarray(cols, dtype=np.int64) >> shift, minlength=nb)[:nb]
        before = np.concatenate([[0], np.cumsum(hist)[:-1]]).astype(np.float64)
        share = before / float(hist.sum())
        if mode == 'auto':
            ids = (np.arange(nb, dtype=np.float64) * (1 << shift)) / K
            if np.abs(share - ids).max() <= ops.ColumnSweepCSR.WARP_AUTO_DEV:
                return None, 0
        return np.minimum(np.floor(share * K), K - 1).astype(np.uint32), shift

    for a in (_skewed_csr(3000, 100000, 200000, 1), _skewed_csr(2000, 777, 50000, 2),
              sp.random(3000, 50000, density=0.002, random_state=np.random.RandomState(5), format='csr', dtype=np.float32)):
        for mode in ('auto', True):
            w0, s0 = numpy_form(a.indices, a.shape[1], mode)
            w1, s1 = ops.ColumnSweepCSR.make_warp(a.indices, a.shape[1], mode)
            assert (w0 is None) == (w1 is None) and s0 == s1
            if w0 is not None:
                assert np.array_equal(w0, w1)


def _ranged_plan(a, **kw):
    import torch
    from stochastic_gcn_amd import ops
    A = ops.ColumnSweepCSR(a, torch.device("cpu"), col_ranges=2, **kw)
    tp = A.tile_ptr.numpy()
    return A, dict(nt=A.ntiles, tile_ptr=tp, colrow=A.colrow.numpy(), val=A.val.numpy(), rows=A.tile_rows.numpy().reshape(-1, 16),
                   slots=A.tile_slots.numpy().reshape(-1, 16), fix=A.fix.numpy() if A.fix is not None else np.zeros((0, 3), np.int32),
                   R=16)


@pytest.mark.parametrize("T", [0, 40])
def test_column_range_plan_encodes_the_matrix_and_keeps_ranges_on_their_tiles(T):
    """ops.ColumnSweepCSR(col_ranges=2) (round 6: a small row block whose rows are split by COLUMN RANGE, range j on XCDs
    4j .. 4j + 3): every nonzero sits in exactly one tile with its row and value; a tile's columns lie in ONE range and the
    tiles come in range order (what xcd_map turns into XCD ranges); a row's pieces take consecutive workspace slots, the
    lower range first; a row that lives in one range only writes its output directly; empty pieces take no slot; the clock's
    table restarts at every range."""
    rng = np.random.RandomState(11)
    M, K = 3000, 20000
    a = sp.random(M, K, density=0.004, random_state=rng, format='csr', dtype=np.float32)
    a = a.tolil()
    a[5, :] = 0                                                   # an empty row
    a[6, :] = 0
    a[6, 3] = 1.5                                                 # a row in the lower range only
    a[7, :] = 0
    a[7, K - 2] = -2.5                                            # ... in the upper range only
    a[8, ::7] = 1.0                                               # a long row: strided pieces inside both ranges
    a = a.tocsr().astype(np.float32)
    a.sort_indices()
    A, p = _ranged_plan(a, T=T)
    assert A.ranged == 2 and A.shape == (M, K) and A.nnz == a.nnz and p['tile_ptr'][-1] == a.nnz
    cr = p['colrow'].view(np.uint32)
    col, lr = (cr & ((1 << 28) - 1)).astype(np.int64), (cr >> 28).astype(np.int64)
    tile = np.repeat(np.arange(p['nt']), np.diff(p['tile_ptr']))
    row = p['rows'][tile, lr]
    assert (row >= 0).all() and row.max() < M
    b = sp.coo_matrix((p['val'], (row, col)), shape=(M, K)).tocsr()
    assert (abs(a - b)).nnz == 0
    cut = A.range_cuts
    assert cut[0] == 0 and cut[-1] == K and len(cut) == 3 and cut[1] % (1 << A.warp_shift) == 0
    nnz_lo = int((a.indices < cut[1]).sum())
    assert abs(nnz_lo - a.nnz / 2) <= 0.02 * a.nnz                # ranges of equal nonzeros
    rng_of_tile = []
    for t in range(p['nt']):
        c = col[p['tile_ptr'][t]:p['tile_ptr'][t + 1]]
        assert (np.diff(c) >= 0).all()
        if c.size:
            j = {int(x >= cut[1]) for x in (c.min(), c.max())}
            assert len(j) == 1
            rng_of_tile.append(j.pop())
    assert rng_of_tile == sorted(rng_of_tile) and set(rng_of_tile) == {0, 1}
    # slots: consecutive per split row, lower range first; unsplit rows write directly
    fixed = {int(r): (int(f), int(c)) for r, f, c in p['fix']}
    assert A.nfix == len(fixed) and A.nslots == sum(c for _, c in fixed.values())
    firsts = sorted(fixed.values())
    assert firsts[0][0] == 0 and all(f0 + c0 == f1 for (f0, c0), (f1, _) in zip(firsts, firsts[1:]))
    live = p['rows'] >= 0
    for r in (6, 7):
        assert r not in fixed and (p['slots'][p['rows'] == r] == -1).all() and (p['rows'] == r).sum() == 1
    assert (p['rows'] == 5).sum() == 1 and (p['slots'][p['rows'] == 5] == -1).all()     # the empty row: ONE writer (of zeros)
    assert 8 in fixed and fixed[8][1] >= (4 if T else 2)
    tile_of_slot = np.repeat(np.arange(p['nt']), 16).reshape(-1, 16)
    tile_rng = np.full(p['nt'], -1)
    for t in range(p['nt']):
        c = col[p['tile_ptr'][t]:p['tile_ptr'][t + 1]]
        tile_rng[t] = int(c[0] >= cut[1]) if c.size else -1
    for r, (first, cnt) in list(fixed.items())[:200] + [(8, fixed[8])]:
        m = p['rows'] == r
        got = p['slots'][m]
        assert sorted(got.tolist()) == list(range(first, first + cnt))
        order = np.argsort(got)
        rngs = tile_rng[tile_of_slot[m]][order]
        assert (np.diff(rngs) >= 0).all()                          # the fix-up adds range 0's pieces first
    split_rows = np.array(sorted(fixed))
    direct = np.setdiff1d(np.unique(p['rows'][live]), split_rows)
    assert all((p['slots'][p['rows'] == r] == -1).all() for r in direct[:200])
    # the clock in range coordinates: 0 at the first bucket of every range, non-decreasing inside a range, below K
    w = A.warp.numpy().view(np.uint32).astype(np.int64)
    b1 = cut[1] >> A.warp_shift
    assert w[0] == 0 and w[b1] == 0 and (np.diff(w[:b1]) >= 0).all() and (np.diff(w[b1:]) >= 0).all() and w.max() < K
    # and the plan's struct asks for XCD ranges and carries a pace like an ungrouped plan
    A.pace[64] = 250
    st = A.struct(64)
    assert st.xcd_map == 1 and st.pace_ns_per_nnz == 250 and st.dev_warp and st.nslots == A.nslots


def test_choose_ranges_takes_the_small_fabric_bound_block_only():
    from stochastic_gcn_amd.ops import ColumnSweepCSR as CS
    assert CS.choose_ranges(29211, 2896039, 232965, 1) == 2        # an eighth of S-Reddit
    assert CS.choose_ranges(58300, 5800000, 232965, 1) == 0        # a quarter: 2 x 58 k pieces do not fit one round
    assert CS.choose_ranges(29211, 2896039, 232965, 2) == 0        # two lane groups: not a one-group plan
    assert CS.choose_ranges(2000, 20000, 232965, 1) == 0           # so sparse that an XCD touches few columns either way
