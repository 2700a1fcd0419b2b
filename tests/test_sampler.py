"""Host neighbour sampler of libsgcn.so vs the golden vectors captured from the REAL
reference C++ (tests/golden/make_golden.py): bit-exact indices AND fp32 weights, across
seeds x {NS, CV, IS} x L x degrees x 3 consecutive batches (in-place permutation state)."""
import numpy as np
import pytest

import golden_util as gu
from stochastic_gcn_amd.scheduler import PyScheduler, Mult, build_plan, I_ADJ_I


def _run(z, gname, big):
    adj = gu.graph(z, gname)
    labels = np.zeros((adj.shape[0], 2), np.float32)
    n_checked = 0
    for case, nit in gu.cases(z, gname):
        c = gu.parse_case(case)
        sch = PyScheduler(adj, labels, c['L'], [c['deg']] * c['L'], gu.placeholders(c['L']),
                          c['seed'], cv=c['cv'], importance=c['imp'])
        for it in range(nit):
            prefix = "%s/%s/it%d" % (gname, case, it)
            fd = sch.batch(z[prefix + "/ids"])
            gu.check_feed_against_golden(z, prefix, fd, big)
            n_checked += 1
        after = sch.c_sch.ivec(I_ADJ_I)
        key = "%s/%s/adj_i_after" % (gname, case)
        if big:
            assert np.array_equal(z[key + "#sha"], gu.digest(after)), case
        else:
            assert np.array_equal(z[key], after), case
    return n_checked


def test_tree_and_rand50_bit_exact():
    z = gu.load("sampler_small.npz")
    assert _run(z, "tree", False) >= 200
    assert _run(z, "rand50", False) >= 200


def test_powerlaw_2k_bit_exact():
    z = gu.load("sampler_big.npz")
    assert _run(z, "pl2k", True) >= 150


def test_reference_test_scheduler_known_answer():
    """The printed output of gcn/test_scheduler.py (seed 0, cv, L=2, degrees [1,2], batch [0]);
    values recorded in SURVEY.md §4."""
    z = gu.load("sampler_small.npz")
    adj = gu.graph(z, "tree")
    ph = gu.placeholders(2)
    sch = PyScheduler(adj, np.zeros((11, 2)), 2, [1, 2], ph, 0, cv=True)
    fd = sch.batch(np.array([0], dtype=np.int32))
    assert fd['fields_2'].tolist() == [0]
    assert fd['fields_1'].tolist() == [0, 2, 1]
    assert fd['fields_0'].tolist() == [0, 2, 1, 3, 4]
    assert fd['adj_1'][0].tolist() == [[0, 1], [0, 2]] and fd['adj_1'][1].tolist() == [0.5, 0.5]
    assert fd['adj_0'][0].tolist() == [[0, 3], [1, 0], [2, 4]]
    assert fd['ffields_1'].tolist() == [2, 1, 3]
    assert fd['ffields_0'].tolist() == [3, 1, 2, 0, 8, 7, 9, 4, 5, 6]
    assert fd['fadj_0'][0][:, 0].tolist() == [0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2]
    assert fd['fadj_0'][0][:, 1].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 3]
    np.testing.assert_array_equal(fd['scales_1'], np.float32([0.81649655]))
    np.testing.assert_array_equal(fd['scales_0'], np.float32([0.57735026, 0.5, 0.5]))


def test_csr_and_transpose_bundles_match_coo():
    z = gu.load("sampler_small.npz")
    adj = gu.graph(z, "rand50")
    ph = gu.placeholders(2)
    sch = PyScheduler(adj, np.zeros((50, 2)), 2, [2, 3], ph, 5, cv=True)
    for _ in range(3):
        fd = sch.batch(np.arange(7, dtype=np.int32))
        for l in range(2):
            for kind in ('adj', 'fadj'):
                idx, w, shape = fd[ph[kind][l]]
                csr = fd[('csr', ph[kind][l])]
                assert csr.shape == tuple(shape)
                assert csr.rowptr.shape[0] == shape[0] + 1 and csr.rowptr[-1] == len(w)
                rows = np.repeat(np.arange(shape[0]), np.diff(csr.rowptr))
                np.testing.assert_array_equal(rows, idx[:, 0])
                np.testing.assert_array_equal(csr.col, idx[:, 1])
                np.testing.assert_array_equal(csr.val, w)
            idx, w, shape = fd[ph['adj'][l]]
            csr = fd[('csr', ph['adj'][l])]
            import scipy.sparse as sp
            a = sp.coo_matrix((w, (idx[:, 0], idx[:, 1])), shape=shape).toarray()
            at = sp.csr_matrix((csr.t_val, csr.t_col, csr.t_rowptr), shape=(shape[1], shape[0])).toarray()
            np.testing.assert_array_equal(a.T, at)
            # transposed rows keep ascending output-row order (deterministic backward sums)
            for r in range(shape[1]):
                seg = csr.t_col[csr.t_rowptr[r]:csr.t_rowptr[r + 1]]
                assert np.all(np.diff(seg) >= 0)


def test_sampler_does_not_touch_callers_csr_and_is_seed_sensitive():
    z = gu.load("sampler_small.npz")
    adj = gu.graph(z, "rand50")
    before = (adj.indices.copy(), adj.data.copy())
    ph = gu.placeholders(1)
    ids = np.arange(10, dtype=np.int32)
    a = PyScheduler(adj, np.zeros((50, 2)), 1, [2], ph, 0).batch(ids)
    b = PyScheduler(adj, np.zeros((50, 2)), 1, [2], ph, 1).batch(ids)
    c = PyScheduler(adj, np.zeros((50, 2)), 1, [2], ph, 0).batch(ids)
    np.testing.assert_array_equal(adj.indices, before[0])
    np.testing.assert_array_equal(adj.data, before[1])
    assert not np.array_equal(a['adj_0'][0], b['adj_0'][0])
    np.testing.assert_array_equal(a['adj_0'][0], c['adj_0'][0])


def test_minibatch_walks_data_and_returns_none_at_epoch_end():
    z = gu.load("sampler_small.npz")
    adj = gu.graph(z, "rand50")
    ph = gu.placeholders(1)
    data = np.arange(23, dtype=np.int32)
    sch = PyScheduler(adj, np.zeros((50, 2)), 1, [2], ph, 0, data=data)
    sizes = []
    while True:
        fd = sch.minibatch(10)
        if fd is None:
            break
        sizes.append(len(fd['fields_1']))
    assert sizes == [10, 10, 3]
    np.random.seed(3)
    sch.shuffle()
    assert sch.start == 0 and sorted(sch.data.tolist()) == list(range(23))


def test_empty_batch_and_isolated_rows():
    z = gu.load("sampler_big.npz")
    adj = gu.graph(z, "pl2k")
    iso = np.where(np.diff(adj.indptr) == 0)[0].astype(np.int32)
    assert len(iso) > 0
    ph = gu.placeholders(1)
    sch = PyScheduler(adj, np.zeros((2000, 2)), 1, [3], ph, 0, cv=True)
    fd = sch.batch(iso[:4])
    assert fd['adj_0'][0].shape == (0, 2) and fd['fadj_0'][0].shape == (0, 2)
    np.testing.assert_array_equal(fd['scales_0'], np.ones(4, np.float32))   # scheduler.cpp:133
    np.testing.assert_array_equal(fd['fields_0'], iso[:4])
    fd = sch.batch(np.zeros(0, dtype=np.int32))
    assert fd['fields_0'].shape == (0,) and fd['adj_0'][0].shape == (0, 2)


def test_mult_golden():
    z = gu.load("mult.npz")
    names = sorted({k.split("/")[0] for k in z.files})
    for n in names:
        m = Mult(z[n + "/prob"])
        assert gu.bits_equal(m.bit, z[n + "/bit"]), n
        got = np.array([m.query_u(u) for u in z[n + "/u"]], dtype=np.int32)
        np.testing.assert_array_equal(got, z[n + "/query_u"])
        draws = np.array([m.query() for _ in range(len(z[n + "/prob"]))], dtype=np.int32)
        np.testing.assert_array_equal(draws, z[n + "/draws"])
    # the known answers printed by gcn/test_mult.cpp
    assert Mult([3, 2, 1, 3]).bit.tolist() == [0, 3, 5, 1, 9]
    assert [Mult([3, 2, 1, 3, 4]).query_u(u) for u in (0, 2, 4, 5.5, 7, 10, 14)] == [0, 0, 1, 2, 3, 4, 8]
    m = Mult([1, 0.1, 100, 10000, 1000])
    assert [m.query() for _ in range(5)] == [3, 4, 2, 0, 1]


def test_mult_empty_prob_is_an_error_not_an_abort():
    from stochastic_gcn_amd._ffi import SgcnError
    with pytest.raises(SgcnError) as e:
        Mult([])
    assert "Prob is empty" in str(e.value)


def test_plan_covers_every_nonzero_once():
    rng = np.random.RandomState(0)
    deg = np.concatenate([rng.randint(0, 40, 300), [0, 1000, 257, 256, 5000]])
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    for T in (0, 64, 256):
        seg, fix, nslots = build_plan(rowptr, T)
        t = T or 256
        assert np.all(seg[:, 2] - seg[:, 1] <= t)
        cover = np.zeros(rowptr[-1], dtype=np.int32)
        for row, s, e, slot in seg:
            assert rowptr[row] <= s <= e <= rowptr[row + 1]
            cover[s:e] += 1
            assert (slot < 0) == (deg[row] <= t)
        assert np.all(cover == 1)
        assert sorted(seg[seg[:, 3] >= 0, 3].tolist()) == list(range(nslots))
        assert fix.shape[0] == int((deg > t).sum())
        for row, first, n in fix:
            assert n == -(-deg[row] // t)


def test_packed_batch_equals_feed_dict_path():
    """sgcn_sched_batch_packed (one C call, the training loop's fast path) must emit exactly the
    arrays PyScheduler.batch emits: same sampler state evolution, fields, ffields, scales,
    labels, COO/CSR of adj and fadj, madj weights, transposed CSR; plus a valid row plan."""
    z = gu.load("sampler_big.npz")
    adj = gu.graph(z, "pl2k")
    labels = np.random.RandomState(0).rand(2000, 3).astype(np.float32)
    for cv in (True, False):
        for L in (1, 2):
            ph = gu.placeholders(L)
            a = PyScheduler(adj, labels, L, [2, 3][:L], ph, 4, cv=cv)
            b = PyScheduler(adj, labels, L, [2, 3][:L], ph, 4, cv=cv)
            for it in range(3):
                ids = np.random.RandomState(it).choice(2000, 50, replace=False).astype(np.int32)
                fa = a.batch(ids)
                pb = b.batch_packed(ids, plan_T=8)
                fb = pb.feed_dict(ph)
                for k, v in fa.items():
                    if isinstance(k, tuple):
                        l = ph['adj'].index(k[1]) if k[1] in ph['adj'] else ph['fadj'].index(k[1])
                        h = pb.csr(l, 0 if k[1] in ph['adj'] else 2)
                        assert gu.bits_equal(h.rowptr, v.rowptr) and gu.bits_equal(h.col, v.col)
                        assert gu.bits_equal(h.val, v.val)
                        if v.t_rowptr is not None:
                            t = pb.csr(l, 1)
                            assert gu.bits_equal(t.rowptr, v.t_rowptr) and gu.bits_equal(t.col, v.t_col)
                            assert gu.bits_equal(t.val, v.t_val)
                        # the packed row plan covers every nonzero exactly once
                        d = pb._csr[l, 0 if k[1] in ph['adj'] else 2]
                        seg = pb._i(d[6], 4 * d[7]).reshape(-1, 4)
                        cover = np.zeros(int(d[2]), np.int32)
                        for row, s, e, slot in seg:
                            cover[s:e] += 1
                        assert np.all(cover == 1)
                        continue
                    w = fb[k]
                    if isinstance(v, tuple):
                        for x, y in zip(v, w):
                            assert np.array_equal(np.asarray(x), np.asarray(y)), (cv, L, it, k)
                    else:
                        assert np.array_equal(v, w), (cv, L, it, k)
                # ABI v16: the medg weights in the order of adj^T's nonzeros (the det-dropout aggregator's third matrix on
                # the way back) = SciPy's transpose of the adjacency pattern carrying them, bit for bit; none without --cv
                import scipy.sparse as sp
                for l in range(L):
                    h, t = pb.csr(l, 0), pb.csr(l, 1)
                    tm = pb._f(*pb._tmedg[l])
                    if not cv:
                        assert tm.shape[0] == 0
                        continue
                    m = sp.csr_matrix((pb._f(*pb._medg[l]), h.col, h.rowptr), shape=tuple(h.shape))
                    mt = m.T.tocsr()                                  # (stable: rows ascending inside a column)
                    assert gu.bits_equal(mt.indptr.astype(np.int32), t.rowptr) and gu.bits_equal(mt.indices.astype(np.int32), t.col)
                    assert gu.bits_equal(mt.data, tm)
            np.testing.assert_array_equal(a.c_sch.ivec(I_ADJ_I), b.c_sch.ivec(I_ADJ_I))


def test_csplan_covers_matrix_and_balances_tiles():
    """Column-sweep plan (sgcn_csplan_*): every nonzero appears exactly once with the right
    output row, nonzeros inside a tile are column-sorted, tiles carry near-equal weight, and
    split rows own consecutive workspace slots."""
    import ctypes as C
    import scipy.sparse as sp
    from stochastic_gcn_amd._ffi import lib, check
    z = gu.load("sampler_big.npz")
    a = gu.graph(z, "pl2k").tocsr()
    a.sort_indices()
    n = a.shape[0]
    rowptr, col, val = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float32)
    for R, T in ((16, 16), (32, 40), (16, 0)):
        nt, nf, ns = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.sgcn_csplan_count(rowptr.ctypes.data, n, R, T, None, C.byref(nt), C.byref(nf), C.byref(ns)))
        tp = np.empty(nt.value + 1, np.int64); cr = np.empty(a.nnz, np.int32); vo = np.empty(a.nnz, np.float32)
        tr = np.empty(nt.value * R, np.int32); ts = np.empty(nt.value * R, np.int32)
        fx = np.empty((max(nf.value, 1), 3), np.int32)
        check(lib.sgcn_csplan_fill(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, n, R, T, None,
                                   tp.ctypes.data, cr.ctypes.data, vo.ctypes.data, tr.ctypes.data,
                                   ts.ctypes.data, fx.ctypes.data))
        shift = 28 if R <= 16 else 27
        w = np.diff(tp)
        tile = np.repeat(np.arange(nt.value), w)
        lr = (cr.view(np.uint32) >> shift).astype(np.int64)
        c = (cr.view(np.uint32) & ((1 << shift) - 1)).astype(np.int64)
        orow = tr.reshape(-1, R)[tile, lr]
        assert orow.min() >= 0
        b = sp.coo_matrix((vo, (orow, c)), shape=a.shape).tocsr()
        assert abs(b - a).max() == 0
        for t in range(nt.value):                       # column-sorted inside every tile
            assert np.all(np.diff(c[tp[t]:tp[t + 1]]) >= 0)
        if nt.value > 4:                                # LPT dealing: within one (virtual) row of the mean
            assert w.max() <= w.mean() + max(T, 64 if T == 0 else T) + R
        slots = ts[ts >= 0]
        assert sorted(slots.tolist()) == list(range(ns.value))
        for row, first, cnt in fx[:nf.value]:
            got = sorted(ts[(tr == row) & (ts >= 0)].tolist())
            assert got == list(range(first, first + cnt))


def test_vector_and_scalar_full_neighbour_placement_agree(monkeypatch):
    """The AVX-512 placement of a row's full neighbour list (sgcn_sched.cpp fplace_row_avx512) and the
    one-by-one loop give the same fields and CSRs over consecutive batches -- on a graph with long rows, rows
    shorter than a vector, and a MULTIGRAPH row whose repeated neighbours fall into the same 16-entry vector
    (the conflict path)."""
    import scipy.sparse as sp
    rng = np.random.RandomState(5)
    n = 400
    rows, cols = [], []
    for r in range(n):
        deg = [0, 1, 3, 15, 16, 17, 40, 120][r % 8]
        c = rng.choice(n, deg, replace=False)
        if r % 16 == 6 and deg >= 3:
            c[2] = c[0]                       # repeated neighbour inside one vector
            c[-1] = c[1]                      # ... and across vectors
        rows += [r] * deg
        cols += c.tolist()
    ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    adj = sp.csr_matrix((rng.rand(len(cols)).astype(np.float32), np.array(cols, np.int32), ptr), shape=(n, n))
    labels = np.zeros((n, 2), np.float32)
    out = []
    for novec in ("", "1"):
        if novec:
            monkeypatch.setenv("SGCN_NO_AVX512", novec)
        else:
            monkeypatch.delenv("SGCN_NO_AVX512", raising=False)
        sch = PyScheduler(adj, labels, 2, [2, 2], gu.placeholders(2), 3, cv=True)
        rng2 = np.random.RandomState(9)
        fds = []
        for it in range(4):              # rows 6 + 16 i are the multigraph rows: two of them in every batch
            rest = np.setdiff1d(rng2.choice(n, 64, replace=False), [6 + 32 * it, 22 + 32 * it])
            fds.append(sch.batch(np.concatenate([[6 + 32 * it, 22 + 32 * it], rest]).astype(np.int32)))
        out.append(fds)
    from stochastic_gcn_amd.scheduler import HostCSR

    def same(va, vb, k):
        if isinstance(va, HostCSR):
            for f in HostCSR.__slots__:
                same(getattr(va, f), getattr(vb, f), (k, f))
        elif isinstance(va, (tuple, list)):
            assert len(va) == len(vb), k
            for xa, xb in zip(va, vb):
                same(xa, xb, k)
        elif va is None:
            assert vb is None, k
        else:
            np.testing.assert_array_equal(np.asarray(va), np.asarray(vb), err_msg=str(k))

    n_f = 0
    for fa, fb in zip(*out):
        assert fa.keys() == fb.keys()
        for k in fa:
            same(fa[k], fb[k], k)
            n_f += str(k).startswith("('csr', 'fadj") or "fadj" in str(k)
    assert n_f >= 8                    # the full-neighbour CSRs were among what was compared
