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
