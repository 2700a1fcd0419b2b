"""GPU parity tests: every HIP kernel of libsgcn.so, called through the C-ABI (ops.py ->
ctypes), against the CPU oracle on the same seeded inputs.  Tolerance: 1e-4 relative
(max|x-ref| / max|ref|) for fp32 sums; bit-exact for pure copies / indices."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import golden_util as gu
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def rand_csr(m, k, density, seed, long_rows=()):
    rng = np.random.RandomState(seed)
    a = sp.random(m, k, density=density, format='lil', random_state=rng, dtype=np.float32)
    for r, n in long_rows:
        cols = rng.choice(k, min(n, k), replace=False)
        a[r, cols] = rng.rand(len(cols)).astype(np.float32) + 0.1
    a = a.tocsr()
    a.sort_indices()
    return a


SPMM_SHAPES = [
    # (M, K, d, density, pitch_pad)
    (1, 1, 1, 1.0, 0), (37, 53, 7, 0.2, 0), (200, 300, 32, 0.05, 0), (300, 200, 128, 0.05, 0),
    (128, 400, 602, 0.05, 0), (128, 400, 602, 0.05, 6), (90, 90, 256, 0.1, 0),
    (64, 64, 1024, 0.2, 0), (50, 70, 33, 0.2, 3), (40, 40, 130, 0.3, 2), (33, 44, 2, 0.3, 0),
]


@pytest.mark.parametrize("M,K,d,dens,pad", SPMM_SHAPES)
@pytest.mark.parametrize("use_plan", [False, True])
def test_spmm_vs_oracle(dev, M, K, d, dens, pad, use_plan):
    from stochastic_gcn_amd import ops
    a = rand_csr(M, K, dens, M + d, long_rows=[(0, min(K, 300))] if M > 1 else [])
    rng = np.random.RandomState(d)
    B = rng.standard_normal((K, d + pad)).astype(np.float32)
    A = ops.DeviceCSR.from_scipy(a, dev, plan_T=16 if use_plan else 0, with_plan=use_plan)
    Bd = T(B, dev)[:, :d]                       # row pitch d+pad, logical width d
    out_full = torch.full((M, d + pad), 7.0, device=dev)
    out = ops.spmm(A, Bd, out=out_full[:, :d])
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    assert onp.rel_err(out.cpu().numpy(), ref) <= TOL
    if pad:                                     # the pitch padding of C is never written
        assert torch.all(out_full[:, d:] == 7.0)


@pytest.mark.parametrize("d", [32, 128, 602])
def test_spmm_fusions(dev, d):
    from stochastic_gcn_amd import ops
    a = rand_csr(150, 120, 0.1, 5, long_rows=[(3, 120), (77, 100)])
    rng = np.random.RandomState(1)
    H = rng.standard_normal((1000, d)).astype(np.float32)
    g = rng.choice(1000, 120, replace=False).astype(np.int32)
    rs, cs = rng.rand(150).astype(np.float32), rng.rand(120).astype(np.float32)
    c0 = rng.standard_normal((150, d)).astype(np.float32)
    for plan in (False, True):
        A = ops.DeviceCSR.from_scipy(a, dev, plan_T=32, with_plan=plan)
        out = T(c0, dev)
        ops.spmm(A, T(H, dev), out=out, gidx=T(g, dev), rscale=T(rs, dev), cscale=T(cs, dev), beta=0.5)
        ref = onp.spmm(a.indptr, a.indices, a.data, H, gidx=g, rscale=rs, cscale=cs, C_in=c0, beta=0.5)
        assert onp.rel_err(out.cpu().numpy(), ref) <= TOL


def test_spmm_tuning_variants_agree(dev):
    from stochastic_gcn_amd import ops, _ffi
    a = rand_csr(200, 300, 0.2, 9)
    B = np.random.RandomState(2).standard_normal((300, 608)).astype(np.float32)
    A = ops.DeviceCSR.from_scipy(a, dev, plan_T=64)
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :602])
    try:
        for nv in (0, 1, 2, 3):
            for un in (0, 2, 4, 8):
                for sm in (0, 1):
                    _ffi.tune("spmm_nv", nv); _ffi.tune("spmm_unroll", un); _ffi.tune("spmm_slabmajor", sm)
                    out = ops.spmm(A, T(B, dev)[:, :602])
                    assert onp.rel_err(out.cpu().numpy(), ref) <= TOL, (nv, un, sm)
    finally:
        _ffi.tune("spmm_nv", 0); _ffi.tune("spmm_unroll", 0); _ffi.tune("spmm_slabmajor", 1)


def test_spmm_edge_cases(dev):
    from stochastic_gcn_amd import ops
    # all-empty matrix -> zeros (and beta keeps C)
    a = sp.csr_matrix((5, 9), dtype=np.float32)
    A = ops.DeviceCSR.from_scipy(a, dev)
    out = ops.spmm(A, torch.ones(9, 12, device=dev))
    assert torch.all(out == 0)
    c = torch.full((5, 12), 2.0, device=dev)
    ops.spmm(A, torch.ones(9, 12, device=dev), out=c, beta=1.0)
    assert torch.all(c == 2.0)
    # zero rows: no launch, no error
    A0 = ops.DeviceCSR.from_scipy(sp.csr_matrix((0, 9), dtype=np.float32), dev)
    assert ops.spmm(A0, torch.ones(9, 4, device=dev)).shape == (0, 4)
    # transposed backward product equals the dense transpose
    a = rand_csr(60, 80, 0.1, 4)
    A = ops.DeviceCSR.from_scipy(a, dev, with_transpose=True)
    g = np.random.RandomState(0).standard_normal((60, 128)).astype(np.float32)
    dx = ops.spmm(A.transpose, T(g, dev))
    assert onp.rel_err(dx.cpu().numpy(), a.T.dot(g.astype(np.float64))) <= TOL


def _sampled_case(seed, n=3000, batch=200, degree=1, d=32, L=1):
    from stochastic_gcn_amd import synthetic
    from stochastic_gcn_amd.scheduler import PyScheduler
    _, train_adj, _, _, _, _, labels, tr, _, _ = synthetic.reddit_like(
        n=n, m=30000, f=4, classes=3, splits=(2000, 300, 700), seed=seed, with_features=False)
    ph = gu.placeholders(L)
    sch = PyScheduler(train_adj, labels, L, [degree] * L, ph, seed, data=tr.copy(), cv=True)
    fd = sch.minibatch(batch)
    return train_adj, ph, fd


@pytest.mark.parametrize("cvd", [True, False])
@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("d,degree", [(32, 1), (128, 1), (128, 5), (602, 2), (30, 3)])
def test_vr_aggregate_vs_oracle(dev, cvd, concat, d, degree):
    from stochastic_gcn_amd import ops
    n = 3000
    _, ph, fd = _sampled_case(11, n=n, degree=degree, d=d)
    rng = np.random.RandomState(d + degree)
    Hbar = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    f0, ff0, s0 = fd[ph['fields'][0]], fd[ph['ffields'][0]], fd[ph['scales'][0]]
    h = rng.standard_normal((len(f0), d)).astype(np.float32)
    mu = rng.standard_normal((len(f0), d)).astype(np.float32)
    adj, fadj = onp.coo_to_csr(fd[ph['adj'][0]]), onp.coo_to_csr(fd[ph['fadj'][0]])
    ref_h, ref_mu, _ = onp.vr_aggregate(adj, fadj, h, mu if cvd else None, Hbar, f0, ff0, s0, cvd, concat)
    for plan_T in (0, 8):                         # 8 forces split rows + the ordered fix-up pass
        A = ops.DeviceCSR.from_host(fd[('csr', ph['adj'][0])], dev)
        P = ops.DeviceCSR.from_host(fd[('csr', ph['fadj'][0])], dev, plan_T=plan_T)
        if plan_T:
            assert P.plan.nfix > 0
        oh, om = ops.vr_aggregate(A, P, T(h, dev), T(mu, dev) if cvd else None, T(Hbar, dev),
                                  T(f0, dev), T(ff0, dev), T(s0, dev), cvd, concat)
        assert onp.rel_err(oh.cpu().numpy(), ref_h) <= TOL
        if cvd:
            assert onp.rel_err(om.cpu().numpy(), ref_mu) <= TOL
        if concat:                                # self rows are exact copies
            np.testing.assert_array_equal(oh.cpu().numpy()[:, :d], h[:adj.shape[0]])


@pytest.mark.parametrize("cvd", [True, False])
@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("d,plan_t", [(128, 16), (128, 0), (32, 8), (30, 8)])
def test_vr_aggregate_two_phase_is_bit_identical_to_the_fused_pass(dev, cvd, concat, d, plan_t):
    """sgcn_vr_aggregate_pre_f32 (history-only sum, what the step program runs beside the dense layers) +
    sgcn_vr_aggregate_post_f32 == sgcn_vr_aggregate_f32, bit for bit, with and without split rows."""
    from stochastic_gcn_amd import ops, synthetic
    from stochastic_gcn_amd.scheduler import PyScheduler
    n = 3000
    _, train_adj, _, _, _, _, labels, tr, _, _ = synthetic.reddit_like(n=n, m=60000, f=8, classes=5, splits=(2000, 300, 700),
                                                                        seed=7, with_features=False)
    ph = {'adj': ['a'], 'madj': ['m'], 'fadj': ['f'], 'fields': ['f0', 'f1'], 'ffields': ['ff'], 'scales': ['s'], 'labels': 'l'}
    sch = PyScheduler(train_adj, labels, 1, [2], ph, 3, data=tr.copy(), cv=True)
    fd = sch.minibatch(200)
    A = ops.DeviceCSR.from_host(fd[('csr', 'a')], dev)
    P = ops.DeviceCSR.from_host(fd[('csr', 'f')], dev, plan_T=plan_t or 100000)
    assert (P.plan.nfix > 0) == bool(plan_t)
    rng = np.random.RandomState(d)
    n0 = fd['f0'].shape[0]
    H = T(rng.uniform(-1, 1, (n, d)).astype(np.float32), dev)
    h, mu = T(rng.standard_normal((n0, d)).astype(np.float32), dev), T(rng.standard_normal((n0, d)).astype(np.float32), dev)
    args = (A, P, h, mu if cvd else None, H, T(fd['f0'], dev), T(fd['ff'], dev), T(fd['s'], dev) if cvd else None, cvd, concat)
    f_h, f_mu = ops.vr_aggregate(*args)
    t_h, t_mu = ops.vr_aggregate_two_phase(*args)
    assert torch.equal(f_h, t_h)
    if cvd:
        assert torch.equal(f_mu, t_mu)


def test_vr_aggregate_fresh_history_identity(dev):
    """Hbar[ifield] == mu and h == mu  =>  h_nbr == mu_nbr == P Hbar[ffield]."""
    from stochastic_gcn_amd import ops
    n, d = 3000, 128
    train_adj, ph, fd = _sampled_case(5, n=n, degree=2, d=d)
    rng = np.random.RandomState(0)
    act = rng.standard_normal((n, d)).astype(np.float32)
    f0, f1, ff0, s0 = fd[ph['fields'][0]], fd[ph['fields'][1]], fd[ph['ffields'][0]], fd[ph['scales'][0]]
    A = ops.DeviceCSR.from_host(fd[('csr', ph['adj'][0])], dev)
    P = ops.DeviceCSR.from_host(fd[('csr', ph['fadj'][0])], dev)
    x = T(act[f0], dev)
    oh, om = ops.vr_aggregate(A, P, x, x.clone(), T(act, dev), T(f0, dev), T(ff0, dev), T(s0, dev), True, False)
    exact = train_adj[f1].dot(act.astype(np.float64))
    assert onp.rel_err(om.cpu().numpy(), exact) <= TOL
    assert onp.rel_err(oh.cpu().numpy(), exact) <= TOL


@pytest.mark.parametrize("d,pad", [(1, 0), (19, 0), (32, 0), (128, 0), (602, 0), (602, 6), (1204, 0), (33, 3)])
def test_gather_scatter_rows(dev, d, pad):
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(d)
    N, n = 500, 123
    a = rng.standard_normal((N, d + pad)).astype(np.float32)
    r = rng.choice(N, n, replace=False).astype(np.int32)
    ad = T(a, dev)[:, :d]
    out = ops.gather_rows(ad, T(r, dev))
    np.testing.assert_array_equal(out.cpu().numpy(), onp.gather_rows(a[:, :d], r))
    src = rng.standard_normal((n, d)).astype(np.float32)
    Hd = T(a, dev)
    ops.scatter_rows(Hd[:, :d], T(r, dev), T(src, dev))
    want = a.copy()
    onp.scatter_rows(want[:, :d], r, src)
    np.testing.assert_array_equal(Hd.cpu().numpy(), want)      # padding columns untouched
    # negative ids are padding slots of the fixed-capacity multi-GPU history exchange: skipped
    r2 = r.copy()
    r2[::3] = -1
    Hd2 = T(a, dev)
    ops.scatter_rows(Hd2[:, :d], T(r2, dev), T(src, dev))
    want2 = a.copy()
    onp.scatter_rows(want2[:, :d], r2, src)
    np.testing.assert_array_equal(Hd2.cpu().numpy(), want2)


def test_gather_rows_golden_dense_slice(dev):
    from stochastic_gcn_amd import ops
    z = gu.load("slice.npz")
    a = T(z["dense/a"], dev)
    for n in sorted({k.split("/")[1] for k in z.files if k.startswith("slice/")}):
        r = z["slice/%s/r" % n]
        out = ops.gather_rows(a, T(r, dev))
        assert gu.bits_equal(out.cpu().numpy(), z["dense/%s/out" % n]), n


def test_csr_slice_golden(dev):
    from stochastic_gcn_amd import ops
    z = gu.load("slice.npz")
    a = sp.csr_matrix((z["a/data"], z["a/indices"], z["a/indptr"]), shape=tuple(z["a/shape"]))
    A = ops.DeviceCSR.from_scipy(a, dev, with_plan=False)
    for n in sorted({k.split("/")[1] for k in z.files if k.startswith("slice/")}):
        r = z["slice/%s/r" % n]
        s = ops.csr_slice(A, r, with_coo_rows=True)
        if ("slice/%s/is_empty_csr" % n) in z.files:
            assert s.nnz == 0 and s.shape == tuple(z["slice/%s/is_empty_csr" % n])
            continue
        idx = np.stack([s.coo_rows.cpu().numpy(), s.col.cpu().numpy()], axis=1)
        assert gu.bits_equal(idx, z["slice/%s/indices" % n]), n
        assert gu.bits_equal(s.val.cpu().numpy(), z["slice/%s/data" % n]), n
        assert s.shape == tuple(z["slice/%s/shape" % n])
        # the slice is a usable CSR: SpMM with it equals the oracle on the sliced rows
        B = np.random.RandomState(1).standard_normal((a.shape[1], 32)).astype(np.float32)
        out = ops.spmm(s, T(B, dev))
        assert onp.rel_err(out.cpu().numpy(), a[r].dot(B)) <= TOL


def test_csr_slice_row_pointer_on_the_device_equals_the_host_pass(dev):
    """sgcn_csr_slice_indptr_dev (one workgroup, what the compiled step uses: SGCN_OP_CSR_SLICE) against
    sgcn_csr_slice_indptr (the reference's c_indptr, gcn/history.cpp:50-58) on the reference's golden slices and on ragged /
    empty / repeated / 20,000-row selections: bit-exact, and the slice built on it equals the host-prefixed one."""
    from stochastic_gcn_amd import ops
    from stochastic_gcn_amd._ffi import lib, check
    z = gu.load("slice.npz")
    a = sp.csr_matrix((z["a/data"], z["a/indices"], z["a/indptr"]), shape=tuple(z["a/shape"]))
    rng = np.random.RandomState(5)
    big = rand_csr(30000, 500, 0.01, 9, long_rows=[(0, 400), (17, 300)])
    cases = [(a, z["slice/%s/r" % n]) for n in sorted({k.split("/")[1] for k in z.files if k.startswith("slice/")})]
    cases += [(big, rng.randint(0, 30000, 20000).astype(np.int32)), (big, np.zeros(0, np.int32)), (big, np.array([0, 0, 17, 0], np.int32)),
              (big, np.arange(255, dtype=np.int32)), (big, np.arange(257, dtype=np.int32))]
    for m, r in cases:
        A = ops.DeviceCSR.from_scipy(m, dev, with_plan=False)
        r = np.ascontiguousarray(r, dtype=np.int32)
        n = int(r.shape[0])
        host = np.empty(n + 1, dtype=np.int32)
        check(lib.sgcn_csr_slice_indptr(n, r.ctypes.data, A.host_rowptr.ctypes.data, host.ctypes.data))
        rd = T(r, dev) if n else torch.zeros(1, dtype=torch.int32, device=dev)
        o_p = torch.full((n + 1,), -7, dtype=torch.int32, device=dev)
        check(lib.sgcn_csr_slice_indptr_dev(n, rd.data_ptr(), A.rowptr.data_ptr(), o_p.data_ptr(), None))
        assert np.array_equal(o_p.cpu().numpy(), host)
        if n:
            s = ops.csr_slice(A, r, with_coo_rows=True)
            assert np.array_equal(s.rowptr.cpu().numpy(), host) and s.nnz == int(host[-1])


def test_full_size_reddit_shape_properties(dev):
    """BASELINE config 3 at full size (N=232,965, nnz~23.2 M, d=602): size-independent
    properties + oracle comparison on a row sample."""
    from stochastic_gcn_amd import ops, synthetic
    n, _, full_adj, _, _, _, _, _, _, _ = synthetic.reddit_like(with_features=False)
    d, ld = 602, 608
    A = ops.DeviceCSR.from_scipy(full_adj, dev)
    assert A.plan.nfix > 1000                    # power-law rows really are split
    g = torch.Generator(device=dev); g.manual_seed(1)
    Bfull = torch.zeros((n, ld), device=dev)
    Bfull[:, :d] = torch.randn((n, d), device=dev, generator=g)
    B = Bfull[:, :d]
    C = ops.spmm(A, B)
    # (1) oracle on 1500 sampled rows incl. the longest rows
    deg = np.diff(full_adj.indptr)
    rows = np.unique(np.concatenate([np.argsort(deg)[-20:], np.random.RandomState(0).choice(n, 1500)]))
    sub = full_adj[rows].tocsr()
    ref = onp.spmm(sub.indptr, sub.indices, sub.data, B.cpu().numpy())
    assert onp.rel_err(C[torch.from_numpy(rows).to(dev)].cpu().numpy(), ref) <= TOL
    # (2) constant columns: A (row-normalised) times ones == 1 on every non-isolated row
    ones = torch.ones((n, 8), device=dev)
    r1 = ops.spmm(A, ones).cpu().numpy()
    want = (deg > 0).astype(np.float32)[:, None] * np.ones((1, 8), np.float32)
    assert np.abs(r1 - want).max() <= 1e-4
    # (3) linearity: A (2x + y) == 2 A x + A y
    y = torch.randn((n, 128), device=dev, generator=g)
    x = torch.randn((n, 128), device=dev, generator=g)
    lhs = ops.spmm(A, 2 * x + y)
    rhs = 2 * ops.spmm(A, x) + ops.spmm(A, y)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= TOL
    # (4) checksum of checksums: sum_i C[i,:] == (column sums of A) . B   (fp64 on host)
    colsum = np.asarray(full_adj.sum(axis=0)).ravel()
    want = colsum.astype(np.float64) @ x.cpu().numpy().astype(np.float64)
    got = ops.spmm(A, x).double().sum(dim=0).cpu().numpy()
    assert np.abs(got - want).max() / np.abs(want).max() <= 1e-4
    # (5) determinism: the split-row fix-up is ordered, two runs are bit-identical
    assert torch.equal(ops.spmm(A, B), C)


@pytest.mark.parametrize("M,K,d,pad", [(37, 53, 8, 0), (300, 200, 128, 0), (128, 400, 602, 6),
                                         (500, 500, 256, 0), (64, 64, 1024, 0), (90, 70, 30, 2)])
def test_spmm_column_sweep_vs_oracle(dev, M, K, d, pad):
    from stochastic_gcn_amd import ops
    a = rand_csr(M, K, 0.08, M + d, long_rows=[(0, min(K, 300)), (M // 2, min(K, 150))])
    rng = np.random.RandomState(d)
    B = rng.standard_normal((K, d + pad)).astype(np.float32)
    A = ops.ColumnSweepCSR(a, dev, T=32)
    assert A.nfix >= 1
    Bd = T(B, dev)[:, :d]
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    out = ops.spmm_cs(A, Bd)
    assert onp.rel_err(out.cpu().numpy(), ref) <= TOL
    # fusions + beta on the same plan
    H = rng.standard_normal((1000, d + pad)).astype(np.float32)
    g = rng.choice(1000, K, replace=False).astype(np.int32)
    rs, cs = rng.rand(M).astype(np.float32), rng.rand(K).astype(np.float32)
    c0 = rng.standard_normal((M, d + pad)).astype(np.float32)
    o2 = T(c0, dev)
    ops.spmm_cs(A, T(H, dev)[:, :d], out=o2[:, :d], gidx=T(g, dev), rscale=T(rs, dev), cscale=T(cs, dev), beta=0.5)
    ref2 = onp.spmm(a.indptr, a.indices, a.data, H[:, :d], gidx=g, rscale=rs, cscale=cs, C_in=c0[:, :d], beta=0.5)
    assert onp.rel_err(o2[:, :d].cpu().numpy(), ref2) <= TOL
    if pad:
        np.testing.assert_array_equal(o2[:, d:].cpu().numpy(), c0[:, d:])


def test_column_sweep_full_size_matches_row_gather(dev):
    from stochastic_gcn_amd import ops, synthetic
    n, _, full_adj, *_ = synthetic.reddit_like(with_features=False)
    A = ops.DeviceCSR.from_scipy(full_adj, dev)
    Acs = ops.ColumnSweepCSR(full_adj, dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    Bfull = torch.zeros((n, 608), device=dev)
    Bfull[:, :602] = torch.randn((n, 602), device=dev, generator=g)
    c1 = ops.spmm(A, Bfull[:, :602])
    c2 = ops.spmm_cs(Acs, Bfull[:, :602])
    assert float((c1 - c2).abs().max() / c1.abs().max()) <= TOL
    assert torch.equal(ops.spmm_cs(Acs, Bfull[:, :602]), c2)       # deterministic


@pytest.mark.parametrize("M,K,d,pad", [(37, 53, 8, 0), (300, 200, 128, 0), (128, 400, 602, 6), (500, 500, 256, 0),
                                         (64, 64, 130, 2), (90, 70, 30, 2), (5000, 3000, 602, 6)])
def test_two_lane_group_column_sweep_vs_oracle(dev, M, K, d, pad, G=2):
    """G = 2 plan (two 16-row bins per wavefront, 128-column passes, half-wave execution masks): the product,
    its fusions and beta against the oracle; bit-identical reruns and paces."""
    from stochastic_gcn_amd import ops
    a = rand_csr(M, K, 0.08 if M < 1000 else 0.01, M + d, long_rows=[(0, min(K, 300)), (M // 2, min(K, 150))])
    rng = np.random.RandomState(d)
    B = rng.standard_normal((K, d + pad)).astype(np.float32)
    A = ops.ColumnSweepCSR(a, dev, T=32, G=G)
    assert A.nfix >= 1 and A.G == G
    Bd = T(B, dev)[:, :d]
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    out = ops.spmm_cs(A, Bd)
    assert onp.rel_err(out.cpu().numpy(), ref) <= TOL
    assert torch.equal(ops.spmm_cs(A, Bd), out)
    for p in (-1, 150, 400):                         # pacing is timing only
        A.pace[d] = p
        assert torch.equal(ops.spmm_cs(A, Bd), out)
    H = rng.standard_normal((K + 500, d + pad)).astype(np.float32)
    g = rng.choice(K + 500, K, replace=False).astype(np.int32)
    rs, cs = rng.rand(M).astype(np.float32), rng.rand(K).astype(np.float32)
    c0 = rng.standard_normal((M, d + pad)).astype(np.float32)
    o2 = T(c0, dev)
    ops.spmm_cs(A, T(H, dev)[:, :d], out=o2[:, :d], gidx=T(g, dev), rscale=T(rs, dev), cscale=T(cs, dev), beta=0.5)
    ref2 = onp.spmm(a.indptr, a.indices, a.data, H[:, :d], gidx=g, rscale=rs, cscale=cs, C_in=c0[:, :d], beta=0.5)
    assert onp.rel_err(o2[:, :d].cpu().numpy(), ref2) <= TOL
    if pad:
        np.testing.assert_array_equal(o2[:, d:].cpu().numpy(), c0[:, d:])
    assert ("g%dk" % G) in A.variant(d)
    # the kernel and its 64-bit-offset form apply every bin's entries in the same order: bit-identical products
    from stochastic_gcn_amd._ffi import lib
    try:
        for knob in (b"cs_g2_wide",):
            lib.sgcn_tune(knob, 1)
            assert torch.equal(ops.spmm_cs(A, Bd), out)
            o3 = T(c0, dev)
            ops.spmm_cs(A, T(H, dev)[:, :d], out=o3[:, :d], gidx=T(g, dev), rscale=T(rs, dev), cscale=T(cs, dev), beta=0.5)
            assert torch.equal(o3, o2)
            lib.sgcn_tune(knob, 0)
    finally:
        lib.sgcn_tune(b"cs_g2_wide", 0)


@pytest.mark.parametrize("M,K,d,pad", [(37, 53, 8, 0), (300, 200, 128, 0), (128, 400, 602, 6), (500, 500, 256, 0),
                                         (64, 64, 130, 2), (90, 70, 30, 2), (5000, 3000, 602, 6)])
def test_four_lane_group_column_sweep_vs_oracle(dev, M, K, d, pad):
    """G = 4 plan (four 16-row bins per wavefront, 64-column passes, quarter-wave execution masks)."""
    test_two_lane_group_column_sweep_vs_oracle(dev, M, K, d, pad, G=4)


@pytest.mark.parametrize("G", [1, 2, 4])
@pytest.mark.parametrize("d,pad", [(256, 0), (602, 6), (40, 0)])
def test_column_sweep_clock_in_work_coordinates(dev, G, d, pad):
    """A matrix whose nonzeros are NOT spread evenly over the column ids (R-MAT-like skew): the plan carries a warp table
    (sgcn_csplan_t.dev_warp), the bins of a wave are aligned in positions, the paced kernels look positions up -- and the
    product is the oracle's, bit-identical at every pace and with the table ignored (pacing is timing only)."""
    from stochastic_gcn_amd import ops
    from stochastic_gcn_amd._ffi import lib
    from test_csplan import _skewed_csr
    M, K = 6000, 50000
    a = _skewed_csr(M, K, 150000, 7 + G)
    a = sp.vstack([a[:M - 1], sp.csr_matrix(np.ones((1, K), np.float32))]).tocsr()       # + one row that must be split
    a.sort_indices()
    rng = np.random.RandomState(d)
    B = rng.standard_normal((K, d + pad)).astype(np.float32)
    A = ops.ColumnSweepCSR(a, dev, G=G, T=4096)
    assert A.warp is not None and A.warp.numel() == ((K - 1) >> A.warp_shift) + 1 and A.nfix >= 1
    Alin = ops.ColumnSweepCSR(a, dev, G=G, T=4096, warp=False, align=getattr(A, 'align', 2048))
    assert Alin.warp is None
    Bd = T(B, dev)[:, :d]
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    outs = []
    for p in (-1, 60, 150, 400, 2000):                        # unpaced, too fast, plausible, slow: the same bits
        A.pace[d] = p
        outs.append(ops.spmm_cs(A, Bd))
        assert onp.rel_err(outs[-1].cpu().numpy(), ref) <= TOL
        assert torch.equal(outs[-1], outs[0])
    try:                                                       # the table ignored: the linear clock on the same plan
        lib.sgcn_tune(b"cs_nowarp", 1)
        assert torch.equal(ops.spmm_cs(A, Bd), outs[0])
    finally:
        lib.sgcn_tune(b"cs_nowarp", 0)
    Alin.pace[d] = 150
    assert onp.rel_err(ops.spmm_cs(Alin, Bd).cpu().numpy(), ref) <= TOL
    # fusions on the warped plan
    rs, cs = rng.rand(M).astype(np.float32), rng.rand(K).astype(np.float32)
    c0 = rng.standard_normal((M, d + pad)).astype(np.float32)
    o2 = T(c0, dev)
    A.pace[d] = 150
    ops.spmm_cs(A, Bd, out=o2[:, :d], rscale=T(rs, dev), cscale=T(cs, dev), beta=0.5)
    ref2 = onp.spmm(a.indptr, a.indices, a.data, B[:, :d], rscale=rs, cscale=cs, C_in=c0[:, :d], beta=0.5)
    assert onp.rel_err(o2[:, :d].cpu().numpy(), ref2) <= TOL
    if pad:
        np.testing.assert_array_equal(o2[:, d:].cpu().numpy(), c0[:, d:])


def test_column_sweep_plan_cache_keeps_the_warp_table(dev, tmp_path):
    from stochastic_gcn_amd import ops
    from test_csplan import _skewed_csr
    a = _skewed_csr(3000, 40000, 90000, 3)
    path = str(tmp_path / "plan.npz")
    A, hit = ops.ColumnSweepCSR.cached(a, dev, path, G=2)
    assert not hit and A.warp is not None
    A.store_if_cached()
    A2, hit = ops.ColumnSweepCSR.cached(a, dev, path, G=2)
    assert hit and A2.warp is not None and A2.warp_shift == A.warp_shift and torch.equal(A2.warp, A.warp)
    B = torch.randn((40000, 64), device=dev)
    assert torch.equal(ops.spmm_cs(A, B), ops.spmm_cs(A2, B))


def test_two_lane_group_full_size_vs_oracle_rows(dev):
    from stochastic_gcn_amd import ops, synthetic
    n, _, full_adj, *_ = synthetic.reddit_like(with_features=False)
    A = ops.ColumnSweepCSR(full_adj, dev, G=2)
    assert A.ntiles % 4096 == 0
    g = torch.Generator(device=dev); g.manual_seed(3)
    Bfull = torch.zeros((n, 608), device=dev)
    Bfull[:, :602] = torch.randn((n, 602), device=dev, generator=g)
    B = Bfull[:, :602]
    best = A.autotune(B)
    c = ops.spmm_cs(A, B)
    deg = np.diff(full_adj.indptr)
    rows = np.unique(np.concatenate([np.argsort(deg)[-20:], np.argsort(deg)[:20],
                                     np.random.RandomState(1).choice(n, 1500, replace=False)]))
    sub = full_adj[rows].tocsr()
    ref = onp.spmm(sub.indptr, sub.indices, sub.data, B.cpu().numpy())
    assert onp.rel_err(c[torch.from_numpy(rows).to(dev)].cpu().numpy(), ref) <= TOL
    print("G=2 full size: %.3f ms at pace %s" % best)
    # ... and the BACKWARD product bench.py times beside it: dB = A^T . dC on its own, separately built and separately
    # autotuned G = 2 plan of the transpose, against the oracle on sampled rows of A^T (heaviest columns of A included)
    full_t = full_adj.T.tocsr()
    full_t.sort_indices()
    AT = ops.ColumnSweepCSR(full_t, dev, G=2)
    dCfull = torch.zeros((n, 608), device=dev)
    dCfull[:, :602] = torch.randn((n, 602), device=dev, generator=g)
    dC = dCfull[:, :602]
    best_t = AT.autotune(dC)
    assert "g2k" in AT.variant(602)
    db = ops.spmm_cs(AT, dC)
    degt = np.diff(full_t.indptr)
    rows = np.unique(np.concatenate([np.argsort(degt)[-20:], np.argsort(degt)[:20],
                                     np.random.RandomState(2).choice(n, 1500, replace=False)]))
    sub = full_t[rows].tocsr()
    ref = onp.spmm(sub.indptr, sub.indices, sub.data, dC.cpu().numpy())
    assert onp.rel_err(db[torch.from_numpy(rows).to(dev)].cpu().numpy(), ref) <= TOL
    print("G=2 full size, A^T: %.3f ms at pace %s" % best_t)


def test_column_sweep_paced_full_size_vs_oracle_rows(dev):
    """The configuration bench.py times -- full-size S-Reddit, d = 602 (pitch 608), the AUTOTUNED clock
    pace -- against the CPU oracle on sampled rows (incl. the longest), not against another HIP kernel."""
    from stochastic_gcn_amd import ops, synthetic
    n, _, full_adj, *_ = synthetic.reddit_like(with_features=False)
    Acs = ops.ColumnSweepCSR(full_adj, dev)
    g = torch.Generator(device=dev); g.manual_seed(2)
    Bfull = torch.zeros((n, 608), device=dev)
    Bfull[:, :602] = torch.randn((n, 602), device=dev, generator=g)
    B = Bfull[:, :602]
    best = Acs.autotune(B)
    assert Acs.pace[602] == best[1]
    c = ops.spmm_cs(Acs, B)
    deg = np.diff(full_adj.indptr)
    rows = np.unique(np.concatenate([np.argsort(deg)[-20:], np.argsort(deg)[:20],
                                     np.random.RandomState(0).choice(n, 1500, replace=False)]))
    sub = full_adj[rows].tocsr()
    ref = onp.spmm(sub.indptr, sub.indices, sub.data, B.cpu().numpy())
    assert onp.rel_err(c[torch.from_numpy(rows).to(dev)].cpu().numpy(), ref) <= TOL
    # a second pace (and the unpaced sweep) give bit-identical results: pacing is timing only
    for p in (-1, 380):
        Acs.pace[602] = p
        assert torch.equal(ops.spmm_cs(Acs, B), c)


@pytest.mark.parametrize("d,pad", [(64, 0), (602, 6), (256, 0)])
def test_grouped_column_sweep_equals_plain_plan_and_oracle(dev, d, pad):
    """Locality-preserving plan (community-ordered columns read through the position map, tiles inside
    row communities, XCD-aware placement, unpaced) == the plain plan == the oracle."""
    from stochastic_gcn_amd import ops, synthetic
    data = synthetic.reddit_sbm(n=9000, m=300000, classes=9, splits=(6000, 1000, 2000), p_in=0.8, seed=2)
    a = data[2]
    comm, nc = ops.reorder_labels(a)
    assert nc > 3
    rng = np.random.RandomState(d)
    B = rng.standard_normal((a.shape[1], d + pad)).astype(np.float32)
    Bd = T(B, dev)[:, :d]
    plain = ops.ColumnSweepCSR(a, dev, T=48)
    grouped = ops.ColumnSweepCSR(a, dev, T=48, col_labels=comm, row_labels=comm)
    assert grouped.grouped and grouped.ntiles >= plain.ntiles
    c0 = ops.spmm_cs(plain, Bd)
    c1 = ops.spmm_cs(grouped, Bd)
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    assert onp.rel_err(c1.cpu().numpy(), ref) <= TOL
    assert float((c0 - c1).abs().max() / c0.abs().max()) <= 1e-5          # same product, other summation order
    assert torch.equal(ops.spmm_cs(grouped, Bd), c1)                       # deterministic
    # fusions through the position map: history gather (gidx), row scale, beta
    H = rng.standard_normal((20000, d + pad)).astype(np.float32)
    gi = rng.choice(20000, a.shape[1], replace=False).astype(np.int32)
    rs = rng.rand(a.shape[0]).astype(np.float32)
    cin = rng.standard_normal((a.shape[0], d + pad)).astype(np.float32)
    o2 = T(cin, dev)
    ops.spmm_cs(grouped, T(H, dev)[:, :d], out=o2[:, :d], gidx=T(gi, dev), rscale=T(rs, dev), beta=0.25)
    ref2 = onp.spmm(a.indptr, a.indices, a.data, H[:, :d], gidx=gi, rscale=rs, C_in=cin[:, :d], beta=0.25)
    assert onp.rel_err(o2[:, :d].cpu().numpy(), ref2) <= TOL
    # labels for rows only (row block of a sharded matrix): still the same product
    rows_only = ops.ColumnSweepCSR(a[:4000], dev, T=48, col_labels=comm, row_labels=comm[:4000])
    c2 = ops.spmm_cs(rows_only, Bd)
    assert onp.rel_err(c2.cpu().numpy(), ref[:4000]) <= TOL


@pytest.mark.parametrize("M,K,d,pad,thr", [(700, 30000, 602, 6, 0), (3000, 20000, 70, 2, 40), (300, 8192, 128, 0, 0),
                                           (64, 5000, 1024, 0, 24)])
def test_column_range_plan_vs_oracle(dev, M, K, d, pad, thr):
    """ColumnSweepCSR(col_ranges=2) (round 6: a small row block's rows split by column range, a range per half of the XCDs,
    all ranges on one clock) against the CPU oracle: unpaced and at several clocks (pacing and placement only: the SAME bits),
    with the fusions, beta, padded pitch untouched, empty rows, rows that live in one range, bit-identical reruns."""
    from stochastic_gcn_amd import ops
    a = rand_csr(M, K, 0.004, M + d, long_rows=[(0, min(K, 3000)), (M // 2, min(K, 900))]).tolil()
    a[3, :] = 0                                                   # an empty row
    a[4, :] = 0
    a[4, K - 1] = 2.0                                             # a row in the upper range only
    a = a.tocsr().astype(np.float32)
    a.sort_indices()
    rng = np.random.RandomState(d)
    B = rng.standard_normal((K, d + pad)).astype(np.float32)
    A = ops.ColumnSweepCSR(a, dev, T=thr, col_ranges=2)
    assert A.ranged == 2 and A.nfix >= 1 and not A.grouped
    Bd = T(B, dev)[:, :d]
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    A.pace[d] = -1
    out = ops.spmm_cs(A, Bd)
    assert onp.rel_err(out.cpu().numpy(), ref) <= TOL
    assert float(out[3].abs().max()) == 0.0
    for pace in (60, 250, 2000):
        A.pace[d] = pace
        assert torch.equal(ops.spmm_cs(A, Bd), out)
    best = A.autotune(Bd)
    assert best[1] == A.pace[d] and torch.equal(ops.spmm_cs(A, Bd), out)
    assert "true>" in A.variant(d).split(" x ")[0] or A.pace[d] <= 0          # a paced launch looks positions up (WARP form)
    one = ops.ColumnSweepCSR(a, dev, T=thr or 64)
    c1 = ops.spmm_cs(one, Bd)
    assert float((c1 - out).abs().max() / c1.abs().max()) <= 1e-5             # the 1-D plan: same product, other summation order
    H = rng.standard_normal((K + 500, d + pad)).astype(np.float32)
    gi = rng.choice(K + 500, K, replace=False).astype(np.int32)
    rs, cs = rng.rand(M).astype(np.float32), rng.rand(K).astype(np.float32)
    c0 = rng.standard_normal((M, d + pad)).astype(np.float32)
    o2 = T(c0, dev)
    ops.spmm_cs(A, T(H, dev)[:, :d], out=o2[:, :d], gidx=T(gi, dev), rscale=T(rs, dev), cscale=T(cs, dev), beta=0.5)
    ref2 = onp.spmm(a.indptr, a.indices, a.data, H[:, :d], gidx=gi, rscale=rs, cscale=cs, C_in=c0[:, :d], beta=0.5)
    assert onp.rel_err(o2[:, :d].cpu().numpy(), ref2) <= TOL
    if pad:
        np.testing.assert_array_equal(o2[:, d:].cpu().numpy(), c0[:, d:])


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (33, 41, 70), (512, 128, 256), (1021, 128, 1204), (200, 300, 50),
                                   (64, 128, 128), (31, 7, 33),
                                   # weight-gradient shapes of the Reddit step: split-K (4..8 output tiles, long K)
                                   (128, 128, 1021), (256, 128, 512), (128, 41, 512), (1204, 128, 1021), (5, 3, 4000)])
def test_gemm_all_transposes_vs_numpy(dev, M, N, K):
    """fp32 MFMA GEMM vs float64 NumPy; asymmetric operands (a swapped row/col map must fail)."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    for ta in (False, True):
        for tb in (False, True):
            a = T(A.T.copy() if ta else A, dev)
            b = T(B.T.copy() if tb else B, dev)
            out = ops.gemm(a, b, trans_a=ta, trans_b=tb)
            assert onp.rel_err(out.cpu().numpy(), ref) <= 1e-5, (ta, tb)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    out = T(c0, dev)
    ops.gemm(T(A, dev), T(B, dev), out=out, accumulate=True)
    assert onp.rel_err(out.cpu().numpy(), ref + c0) <= 1e-5
    out2 = T(c0, dev)
    ops.gemm(T(A, dev), T(B, dev), out=out2, accumulate=True)
    assert torch.equal(out, out2)                       # split-K partials are summed in a fixed order
    # identity check with an asymmetric B
    eye = T(np.eye(K, dtype=np.float32), dev)
    np.testing.assert_array_equal(ops.gemm(eye, T(B, dev)).cpu().numpy(), B)


@pytest.mark.parametrize("M,N,K,norm,relu", [(257, 128, 96, True, True), (40, 41, 128, False, False),
                                             (100, 16, 20, True, True), (65, 128, 1204, False, True),
                                             (50, 32, 64, True, False)])
def test_dense_fwd_fused_vs_oracle(dev, M, N, K, norm, relu):
    from stochastic_gcn_amd import ops
    from oracle import model_np as mnp
    rng = np.random.RandomState(M + K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    off = rng.standard_normal((1, N)).astype(np.float32) * 0.1
    sc = (1 + 0.1 * rng.standard_normal((1, N))).astype(np.float32)
    y, ctx = ops.dense_fwd(T(X, dev), T(W, dev), T(off, dev) if norm else None, T(sc, dev) if norm else None, relu)
    ref = (X @ W).astype(np.float32)
    if norm:
        ref, (xhat, rstd) = mnp.layer_norm_fwd(ref, off, sc)
        assert onp.rel_err(ctx[0].cpu().numpy(), xhat) <= TOL
        assert onp.rel_err(ctx[1].cpu().numpy(), rstd.ravel()) <= TOL
    if relu:
        ref = np.maximum(ref, 0)
    assert onp.rel_err(y.cpu().numpy(), ref) <= TOL


@pytest.mark.parametrize("d,pad", [(300, 4), (320, 0), (330, 6), (602, 6), (700, 4), (260, 0)])
def test_column_sweep_extra_plane_widths(dev, d, pad):
    """Widths around the 256 / 320-column pass boundaries of the pinned sweep kernel (64 float4 +
    up to 64 extra fp32 columns per pass), incl. ragged last vectors and pitch padding."""
    from stochastic_gcn_amd import ops
    a = rand_csr(400, 350, 0.06, d, long_rows=[(5, 300)])
    rng = np.random.RandomState(d)
    B = rng.standard_normal((350, d + pad)).astype(np.float32)
    A = ops.ColumnSweepCSR(a, dev, T=40)
    out_full = torch.full((400, d + pad), 3.0, device=dev)
    ops.spmm_cs(A, T(B, dev)[:, :d], out=out_full[:, :d])
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    assert onp.rel_err(out_full[:, :d].cpu().numpy(), ref) <= TOL
    if pad:
        assert torch.all(out_full[:, d:] == 3.0)
    A.pace[d] = 300                                   # paced and unpaced sweeps agree bit for bit
    o2 = ops.spmm_cs(A, T(B, dev)[:, :d])
    assert torch.equal(o2, out_full[:, :d])


# ---- counter-based dropout (sgcn_dropout_t): standalone and fused into the GEMMs -------------------
@pytest.mark.parametrize("n,d,keep", [(1, 1, 0.5), (37, 19, 0.8), (1021, 1204, 0.8), (512, 256, 0.3), (64, 128, 1.0)])
def test_dropout_is_the_oracles_hash_mask(dev, n, d, keep):
    from stochastic_gcn_amd import ops
    from oracle import model_np as mnp
    rng = np.random.RandomState(n + d)
    x = rng.standard_normal((n, d + 3)).astype(np.float32)
    key = ops.dropout_key(1, 5, n)
    assert key == mnp.dropout_key(1, 5, n)
    m = mnp.hash_mask(key, (n, d), keep) if keep < 1 else np.ones((n, d), np.float32)
    want = x[:, :d] * (m * np.float32(1.0 / keep))
    got = ops.dropout(T(x, dev)[:, :d], ops.Drop(keep, key))
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    if keep < 1:
        assert abs(m.mean() - keep) < 4 * np.sqrt(keep * (1 - keep) / m.size) + 1e-3     # P(keep) = keep
    v = rng.standard_normal(777).astype(np.float32)                                       # 1-D (sparse values)
    got = ops.dropout(T(v, dev), ops.Drop(0.6, key))
    np.testing.assert_array_equal(got.cpu().numpy(), v * (mnp.hash_mask(key, (777,), 0.6) * np.float32(1 / 0.6)))


@pytest.mark.parametrize("n,K,N,norm", [(300, 96, 128, True), (1021, 1204, 128, True), (77, 40, 41, False)])
def test_fused_dropout_dense_layer_forward_and_backward(dev, n, K, N, norm):
    """[dropout(x) ; mu] @ W -> LN -> ReLU in one launch (no concatenation, no mask tensor), the
    weight gradient with the mask recomputed, the input gradient with the mask in the epilogue --
    against the oracle's explicit-mask arithmetic."""
    from stochastic_gcn_amd import ops
    from oracle import model_np as mnp
    rng = np.random.RandomState(K)
    x = rng.standard_normal((n, K)).astype(np.float32)
    mu = rng.standard_normal((n, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    off = (0.1 * rng.standard_normal((1, N))).astype(np.float32)
    sc = (1 + 0.1 * rng.standard_normal((1, N))).astype(np.float32)
    keep, key = 0.8, ops.dropout_key(3, 2, 11)
    drop = ops.Drop(keep, key)
    m = mnp.hash_mask(key, (n, K), keep) * np.float32(1.0 / keep)
    xd = (x * m).astype(np.float32)
    stacked = np.concatenate([xd, mu], 0)
    pre = (stacked.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)
    if norm:
        pre, _ = mnp.layer_norm_fwd(pre, off, sc)
    want = np.maximum(pre, 0)
    y, ctx = ops.dense_fwd(T(x, dev), T(W, dev), T(off, dev) if norm else None, T(sc, dev) if norm else None,
                           True, x2=T(mu, dev), drop=drop)
    assert onp.rel_err(y.cpu().numpy(), want) <= TOL
    # the same without the second stream, and through the plain GEMM
    y1 = ops.gemm(T(x, dev), T(W, dev), drop_a=drop)
    assert onp.rel_err(y1.cpu().numpy(), xd.astype(np.float64) @ W.astype(np.float64)) <= 1e-5
    # backward pieces
    g = rng.standard_normal((n, N)).astype(np.float32)
    dW = ops.gemm(T(x, dev), T(g, dev), trans_a=True, drop_a=drop)
    assert onp.rel_err(dW.cpu().numpy(), xd.T.astype(np.float64) @ g.astype(np.float64)) <= 1e-5
    dx = ops.gemm(T(g, dev), T(W, dev), trans_b=True, drop_c=drop)
    want_dx = (g.astype(np.float64) @ W.T.astype(np.float64)) * m
    assert onp.rel_err(dx.cpu().numpy(), want_dx) <= 1e-5
    np.testing.assert_array_equal(dx.cpu().numpy() == 0, m == 0)           # exactly the dropped elements


@pytest.mark.parametrize("n,K,N,norm,relu,with_drop", [(300, 96, 128, True, True, True), (512, 256, 128, True, True, False),
                                                       (77, 40, 41, False, False, True), (1021, 128, 128, True, True, True),
                                                       (50, 20, 160, True, True, True)])
def test_dense_bwd_composite_matches_the_three_steps(dev, n, K, N, norm, relu, with_drop):
    """sgcn_dense_bwd_f32 (one call) == ln_act_bwd -> gemm(dW) -> gemm(dx), bit for bit."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(n + K + N)
    x = T(rng.standard_normal((n, K)).astype(np.float32), dev)
    W = T((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32), dev)
    off = T((0.1 * rng.standard_normal((1, N))).astype(np.float32), dev) if norm else None
    sc = T((1 + 0.1 * rng.standard_normal((1, N))).astype(np.float32), dev) if norm else None
    drop = ops.Drop(0.7, 999) if with_drop else None
    if N <= 128 or not (norm or relu):
        y, ctx = ops.dense_fwd(x, W, off, sc, relu, drop=drop)
    else:
        y, ctx = ops.ln_act_fwd(ops.gemm(x, W, drop_a=drop), off, sc, relu)
    dy = T(rng.standard_normal((n, N)).astype(np.float32), dev)
    # three steps
    dW1 = torch.zeros((K, N), device=dev); do1 = torch.zeros((1, N), device=dev); ds1 = torch.zeros((1, N), device=dev)
    g = ops.ln_act_bwd(dy, y, ctx, sc, relu, do1 if norm else None, ds1 if norm else None) if (norm or relu) else dy
    ops.gemm(x, g, out=dW1, trans_a=True, accumulate=True, drop_a=drop)
    dx1 = ops.gemm(g, W, trans_b=True, drop_c=drop)
    # one call
    dW2 = torch.zeros((K, N), device=dev); do2 = torch.zeros((1, N), device=dev); ds2 = torch.zeros((1, N), device=dev)
    dx2 = ops.dense_bwd(dy, y, ctx, sc, relu, x, W, dW2, do2 if norm else None, ds2 if norm else None,
                        need_dx=True, drop=drop)
    assert torch.equal(dW1, dW2) and torch.equal(dx1, dx2) and torch.equal(do1, do2) and torch.equal(ds1, ds2)
    assert ops.dense_bwd(dy, y, ctx, sc, relu, x, W, dW2, do2 if norm else None, ds2 if norm else None,
                         need_dx=False, drop=drop) is None


@pytest.mark.parametrize("d", [16, 128, 37])
def test_spmm_addend_epilogue(dev, d):
    """C = A . B + [add ; 0] in one launch == the three-launch form (zeros, copy, beta = 1)."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(d)
    M, K, n_add = 700, 300, 300
    a = sp.random(M, K, density=0.05, format='lil', dtype=np.float32, random_state=rng)
    a[5] = 0                                           # an empty row inside and outside the addend range
    a[650] = 0
    a = a.tocsr()
    a.eliminate_zeros()
    B = rng.standard_normal((K, 2 * d)).astype(np.float32)
    g = rng.standard_normal((n_add, 2 * d)).astype(np.float32)
    A = ops.DeviceCSR.from_scipy(a, dev, plan_T=8)     # small T: split rows exercise the fix-up epilogue too
    Bd, gd = T(B, dev), T(g, dev)
    got = ops.spmm(A, Bd[:, d:], add=gd[:, :d], add_rows=n_add)
    want = torch.zeros((M, d), device=dev)
    want[:n_add] = gd[:, :d]
    ops.spmm(A, Bd[:, d:], out=want, beta=1.0)
    assert onp.rel_err(got.cpu().numpy(), want.cpu().numpy()) <= 1e-6
    ref = a.dot(B[:, d:].astype(np.float64)); ref[:n_add] += g[:, :d]
    assert onp.rel_err(got.cpu().numpy(), ref) <= TOL


def test_dense_layer_reads_its_rows_through_an_index(dev):
    """ops.GatheredRows: [dropout(F[idx]) ; F[idx]] @ W -> LN -> ReLU and its backward without ever
    materialising F[idx] == the same on the gathered copy, bit for bit."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(4)
    Nf, K, N, n = 5000, 96, 128, 777
    F = T(rng.standard_normal((Nf, K)).astype(np.float32), dev)
    idx = T(rng.choice(Nf, n, replace=False).astype(np.int32), dev)
    W = T((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32), dev)
    off = T((0.1 * rng.standard_normal((1, N))).astype(np.float32), dev)
    sc = T((1 + 0.1 * rng.standard_normal((1, N))).astype(np.float32), dev)
    drop = ops.Drop(0.8, 4242)
    lazy = ops.GatheredRows(F, idx)
    x = ops.gather_rows(F, idx)
    y1, c1 = ops.dense_fwd(x, W, off, sc, True, x2=x, drop=drop)
    y2, c2 = ops.dense_fwd(lazy, W, off, sc, True, x2=lazy, drop=drop)
    assert torch.equal(y1, y2) and torch.equal(c1[0], c2[0]) and torch.equal(c1[1], c2[1])
    dy = T(rng.standard_normal((n, N)).astype(np.float32), dev)
    out = []
    for xin in (x, lazy):
        dW = torch.zeros((K, N), device=dev); do = torch.zeros((1, N), device=dev); ds = torch.zeros((1, N), device=dev)
        dx = ops.dense_bwd(dy, y1[:n], (c1[0][:n], c1[1][:n]), sc, True, xin, W, dW, do, ds, need_dx=True, drop=drop)
        out.append((dW, do, ds, dx))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert torch.equal(lazy.materialize(), x)


@pytest.mark.parametrize("n,c", [(1, 1), (37, 121), (512, 41), (300, 3)])
def test_sigmoid_ce_vs_numpy(dev, n, c):
    """Multitask loss (gcn/models.py:77-79,86-90,198-200) in float64 NumPy."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(n + c)
    z = (rng.standard_normal((n, c)) * 3).astype(np.float32)
    y = (rng.rand(n, c) < 0.3).astype(np.float32)
    stats, dz, pred = ops.sigmoid_ce(T(z, dev), T(y, dev), want_grad=True, want_pred=True)
    z64 = z.astype(np.float64)
    ce = np.maximum(z64, 0) - z64 * y + np.log1p(np.exp(-np.abs(z64)))
    p = 1.0 / (1.0 + np.exp(-z64))
    st = stats[:4].cpu().numpy()
    assert abs(st[2] - ce.mean()) <= 1e-5 * max(1.0, ce.mean()) and abs(st[0] - ce.sum()) <= 1e-4 * ce.sum()
    assert st[1] == ((z > 0) == (y > 0.5)).sum() and abs(st[3] - ((z > 0) == (y > 0.5)).mean()) <= 1e-6
    assert onp.rel_err(pred.cpu().numpy(), p.astype(np.float32)) <= 1e-5
    assert onp.rel_err(dz.cpu().numpy(), ((p - y) / (n * c)).astype(np.float32)) <= 1e-5
    s2, dz2, pr2 = ops.sigmoid_ce(T(z, dev), T(y, dev), want_grad=False, want_pred=False)
    assert dz2 is None and pr2 is None and torch.equal(s2[:4], stats[:4])            # deterministic


def test_l2_penalty_range_only(dev):
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(0)
    th = rng.standard_normal(5000).astype(np.float32)
    g0 = rng.standard_normal(5000).astype(np.float32)
    theta, grad = T(th, dev), T(g0, dev)
    loss = torch.tensor([0.0, 0.0, 1.25, 0.0], device=dev)
    ops.l2_penalty(theta, 1000, 3333, 5e-4, grad=grad, loss=loss[2:3])
    want_g = g0.copy()
    want_g[1000:3333] += np.float32(5e-4) * th[1000:3333]
    np.testing.assert_allclose(grad.cpu().numpy(), want_g, rtol=1e-6, atol=1e-9)
    want_l = 1.25 + 0.5 * 5e-4 * (th[1000:3333].astype(np.float64) ** 2).sum()
    got = loss.cpu().numpy()
    assert abs(got[2] - want_l) <= 1e-5 * want_l and got[0] == 0 and got[1] == 0 and got[3] == 0
    ops.l2_penalty(theta, 10, 10, 5e-4, grad=grad, loss=loss[2:3])                # empty range: no-op
    assert abs(loss.cpu().numpy()[2] - got[2]) == 0


@pytest.mark.parametrize("n,f,per", [(1, 5, 2), (616, 1433, 18), (107, 500, 50), (4000, 300, 40), (50, 20000, 7)])
def test_csr_transpose_index_is_the_stable_transpose(dev, n, f, per):
    """Device counting sort == SciPy's CSR transpose (rows ascending inside every column), bit-exact; the
    > 16,384-column case takes the global-memory counter path."""
    import scipy.sparse as sp
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(n)
    rows = np.repeat(np.arange(n), per)
    cols = np.concatenate([rng.choice(f, per, replace=False) for _ in range(n)])
    a = sp.csr_matrix((rng.rand(n * per).astype(np.float32) + 0.1, (rows, cols)), shape=(n, f))
    a.sort_indices()
    A = ops.DeviceCSR.from_scipy(a, dev, with_plan=False)
    A.coo_rows = T(np.repeat(np.arange(n, dtype=np.int32), np.diff(a.indptr)), dev)
    t_rowptr, t_row, t_src = ops.csr_transpose_index(A)
    at = a.T.tocsr()            # SciPy's transpose conversion is stable: ascending rows per column
    at.sort_indices()
    np.testing.assert_array_equal(t_rowptr.cpu().numpy(), at.indptr)
    np.testing.assert_array_equal(t_row.cpu().numpy(), at.indices)
    val_t = ops.gather_f32(A.val, t_src)
    np.testing.assert_array_equal(val_t.cpu().numpy(), at.data)
    # twice the same (no atomics decide an order)
    r2 = ops.csr_transpose_index(A)
    assert all(torch.equal(x, y) for x, y in zip((t_rowptr, t_row, t_src), r2))
    # empty matrix
    e = sp.csr_matrix((3, 7), dtype=np.float32)
    E = ops.DeviceCSR.from_scipy(e, dev, with_plan=False)
    E.coo_rows = torch.zeros(0, dtype=torch.int32, device=dev)
    tp, tr_, ts = ops.csr_transpose_index(E)
    assert tp.cpu().numpy().tolist() == [0] * 8 and tr_.numel() == 0




def test_column_sweep_plan_cache_round_trip_both_group_counts(dev, tmp_path):
    """ColumnSweepCSR.cached: a plan written to disk (G = 1 and G = 2, with its autotuned pace) is re-loaded
    for the same matrix and the same group count only, and multiplies bit-identically."""
    from stochastic_gcn_amd import ops
    a = rand_csr(3000, 2500, 0.02, 11, long_rows=[(7, 900)])
    B = T(np.random.RandomState(2).standard_normal((2500, 160)).astype(np.float32), dev)
    for G in (1, 2):
        path = str(tmp_path / ("plan%d.npz" % G))
        A, hit = ops.ColumnSweepCSR.cached(a, dev, path, G=G)
        assert not hit and A.G == G
        A.pace[160], A.tuned_ms[160] = 300, 50.0       # (a pace travels with the product's time at it: the guard's yardstick)
        A.store_if_cached()
        ref = ops.spmm_cs(A, B)
        A2, hit2 = ops.ColumnSweepCSR.cached(a, dev, path, G=G)
        assert hit2 and A2.G == G and A2.pace == {160: 300} and A2.tuned_ms == {160: 50.0}
        assert torch.equal(ops.spmm_cs(A2, B), ref)
        assert A2.variant(160) == A.variant(160)
        other, hit3 = ops.ColumnSweepCSR.cached(a, dev, path, G=3 - G)          # same file, other group count: rebuilt
        assert not hit3 and other.G == 3 - G
        b = a.copy(); b.data = b.data * 2
        assert not ops.ColumnSweepCSR.cached(b, dev, path, G=G)[1]              # another matrix: rebuilt
        with open(path, "r+b") as f:                                            # a truncated / corrupt cache file is a
            f.truncate(1000)                                                    # cache miss, not a crash at startup
        assert not ops.ColumnSweepCSR.cached(a, dev, path, G=G)[1]
    assert [ops.ColumnSweepCSR.choose_g(d) for d in (32, 128, 256, 320, 602, 640)] == [2, 2, 2, 1, 2, 2]


# ---- LDS-staged column sweep (sgcn_spmm_lds.hip) ---------------------------------------------------------------------
@pytest.mark.parametrize("M,K,d,pad", [(37, 53, 8, 0), (300, 200, 128, 0), (128, 400, 602, 6), (500, 500, 256, 0),
                                         (64, 64, 130, 2), (90, 70, 30, 2), (5000, 3000, 602, 6), (2000, 1500, 129, 3)])
@pytest.mark.parametrize("min_reuse", [1, 2])
@pytest.mark.parametrize("unit", [False, True])
def test_lds_sweep_vs_oracle(dev, M, K, d, pad, min_reuse, unit):
    """The LDS-staged sweep (+ its residual through the ordinary sweep) against the oracle: plain product, split rows,
    labels, row scale / beta, bit-identical reruns, pitch padding untouched -- for general values (packed FMA per nonzero)
    and for a row-normalised matrix (unit plan: the value folded into the row scale, packed ADD per nonzero)."""
    from stochastic_gcn_amd import ops
    a = rand_csr(M, K, 0.08 if M < 1000 else 0.01, M + d, long_rows=[(0, min(K, 300)), (M // 2, min(K, 150))])
    if unit:                                               # D^-1 A: one value per row (gcn/utils.py:299-309)
        a.data[:] = 1.0
        a = sp.diags((1.0 / np.maximum(np.diff(a.indptr), 1)).astype(np.float32)).dot(a).tocsr().astype(np.float32)
        a.sort_indices()
    rng = np.random.RandomState(d)
    B = rng.standard_normal((K, d + pad)).astype(np.float32)
    Bd = T(B, dev)[:, :d]
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    lab = (rng.randint(0, 3, M).astype(np.int32), rng.randint(0, 3, K).astype(np.int32))
    for labels, T_, slots in ((None, 0, 128), (lab, 32, 80), (lab, 32, 128)):     # ring: 2 x 128 slots or 3 x 80
        A = ops.LdsSweepCSR(a, dev, labels=labels, T=T_, min_reuse=min_reuse, ring_slots=slots)
        assert bool(A.unit) == unit and (A.S, A.nparts) == ((128, 2) if slots == 128 else (80, 3))
        if T_:
            assert A.nfix >= 1
        if min_reuse == 1:
            assert A.residual is None
        out_full = torch.full((M, d + pad), 7.0, device=dev)
        out = ops.spmm_lds(A, Bd, out=out_full[:, :d])
        assert onp.rel_err(out.cpu().numpy(), ref) <= TOL
        if pad:
            assert torch.all(out_full[:, d:] == 7.0)
        assert torch.equal(ops.spmm_lds(A, Bd), out)                   # deterministic
    rs = rng.rand(M).astype(np.float32)
    c0 = rng.standard_normal((M, d + pad)).astype(np.float32)
    o2 = T(c0, dev)
    ops.spmm_lds(A, Bd, out=o2[:, :d], rscale=T(rs, dev), beta=0.5)
    ref2 = onp.spmm(a.indptr, a.indices, a.data, B[:, :d], rscale=rs, C_in=c0[:, :d], beta=0.5)
    assert onp.rel_err(o2[:, :d].cpu().numpy(), ref2) <= TOL
    if pad:
        np.testing.assert_array_equal(o2[:, d:].cpu().numpy(), c0[:, d:])


def test_lds_sweep_on_communities_vs_oracle_and_column_sweep(dev):
    """A graph WITH communities at a size where tiles, chunks and the XCD placement all matter (a few hundred chunks
    per tile): the LDS sweep == the oracle == the plain column sweep up to summation order."""
    from stochastic_gcn_amd import ops, synthetic
    data = synthetic.reddit_sbm(n=30000, m=1500000, classes=7, splits=(20000, 4000, 6000), p_in=0.8, seed=2)
    a = data[2]
    comm = data[6].argmax(1).astype(np.int32)
    d = 602
    rng = np.random.RandomState(0)
    B = np.zeros((a.shape[1], 608), np.float32)
    B[:, :d] = rng.standard_normal((a.shape[1], d))
    Bd = T(B, dev)[:, :d]
    ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d])
    A = ops.LdsSweepCSR(a, dev, labels=comm, min_reuse=2)
    assert A.host_stats["reuse"] > 3 and A.residual is not None and A.unit
    c1 = ops.spmm_lds(A, Bd)
    assert onp.rel_err(c1.cpu().numpy(), ref) <= TOL
    Ag = ops.LdsSweepCSR(a, dev, labels=comm, min_reuse=2, general=True)       # the same plan with the values kept per nonzero
    assert not Ag.unit
    cg = ops.spmm_lds(Ag, Bd)
    assert float((cg - c1).abs().max() / c1.abs().max()) <= 1e-5
    c0 = ops.spmm_cs(ops.ColumnSweepCSR(a, dev, G=2), Bd)
    assert float((c0 - c1).abs().max() / c0.abs().max()) <= 1e-5
    assert torch.equal(ops.spmm_lds(A, Bd), c1)
    # the planned part alone + the residual alone = the whole (linearity over the split)
    loc = ops.spmm_lds(A, Bd, local_only=True)
    res = ops.spmm_cs(A.residual, Bd)
    assert float((loc + res - c1).abs().max() / c1.abs().max()) <= 1e-5
    # the transpose (one value per COLUMN: the backward product): a unit plan on the pattern + a row scale of the operand
    at = a.T.tocsr().astype(np.float32)
    at.sort_indices()
    At = ops.LdsSweepCSR(at, dev, labels=comm, min_reuse=2)
    assert At.unit and At.col_fold is not None
    reft = onp.spmm(at.indptr, at.indices, at.data, B[:, :d])
    rs = rng.rand(at.shape[0]).astype(np.float32)
    c0t = T(rng.standard_normal((at.shape[0], 608)).astype(np.float32), dev)
    keep = c0t.clone()
    ct = ops.spmm_lds(At, Bd)
    assert onp.rel_err(ct.cpu().numpy(), reft) <= TOL and torch.equal(ops.spmm_lds(At, Bd), ct)
    ops.spmm_lds(At, Bd, out=c0t[:, :d], rscale=T(rs, dev), beta=0.5)
    ref2 = onp.spmm(at.indptr, at.indices, at.data, B[:, :d], rscale=rs, C_in=keep[:, :d].cpu().numpy(), beta=0.5)
    assert onp.rel_err(c0t[:, :d].cpu().numpy(), ref2) <= TOL and torch.equal(c0t[:, d:], keep[:, d:])
    assert torch.equal(Bd, T(B, dev)[:, :d])                                    # the caller's operand is not touched


def test_lds_sweep_edge_cases(dev):
    """No nonzeros at all, a single row / column, and a plan whose every nonzero is residual (no column reaches min_reuse):
    the tiles still write their rows (rscale (.) 0 + beta C), the pitch padding stays untouched."""
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(3)
    for a, mr in ((sp.csr_matrix((5, 7), dtype=np.float32), 1), (sp.csr_matrix((700, 1), dtype=np.float32), 3),
                  (rand_csr(1, 5, 0.9, 1), 1), (rand_csr(300, 4000, 0.001, 2), 3)):
        M, K = a.shape
        d, pitch = 30, 36
        B = rng.standard_normal((K, pitch)).astype(np.float32)
        c0 = rng.standard_normal((M, pitch)).astype(np.float32)
        rs = rng.rand(M).astype(np.float32)
        A = ops.LdsSweepCSR(a, dev, min_reuse=mr)
        out = T(c0, dev)
        ops.spmm_lds(A, T(B, dev)[:, :d], out=out[:, :d], rscale=T(rs, dev), beta=0.5)
        ref = onp.spmm(a.indptr, a.indices, a.data, B[:, :d], rscale=rs, C_in=c0[:, :d], beta=0.5)
        assert onp.rel_err(out[:, :d].cpu().numpy(), ref) <= TOL
        np.testing.assert_array_equal(out[:, d:].cpu().numpy(), c0[:, d:])


def test_lds_sweep_full_size_sbm_vs_oracle_rows(dev):
    """What `bench.py --workload reddit-sbm` times -- the full-size graph (N = 232,965, 22.7 M nonzeros, p_in 0.8), d = 602,
    communities from label propagation on the graph alone, the planned part through the LDS ring (pair words, requests from
    inside the chunk statement) and the residual on the four-group column sweep at its autotuned clock -- against the CPU
    oracle on sampled rows (the heaviest and the emptiest included); the same for the transpose (one value per column: a
    unit plan + a row scale of the operand); reruns bit-identical."""
    from stochastic_gcn_amd import ops, synthetic
    n, _, a, *_ = synthetic.reddit_sbm(p_in=0.8)
    d = 602
    g = torch.Generator(device=dev); g.manual_seed(7)
    Bfull = torch.zeros((n, 608), device=dev)
    Bfull[:, :d] = torch.randn((n, d), device=dev, generator=g)
    B = Bfull[:, :d]
    Bh = B.cpu().numpy()
    at = a.T.tocsr().astype(np.float32)
    at.sort_indices()
    for m, want_fold in ((a, False), (at, True)):
        A = ops.LdsSweepCSR.for_graph(m, dev)
        assert A is not None and A.unit and (A.col_fold is not None) == want_fold
        assert A.residual is not None and A.residual.G == 4 and "g4k" in A.variant(d)
        A.autotune(B)
        c = ops.spmm_lds(A, B)
        deg = np.diff(m.indptr)
        rows = np.unique(np.concatenate([np.argsort(deg)[-20:], np.argsort(deg)[:20],
                                         np.random.RandomState(4).choice(n, 1500, replace=False)]))
        sub = m[rows].tocsr()
        ref = onp.spmm(sub.indptr, sub.indices, sub.data, Bh)
        assert onp.rel_err(c[torch.from_numpy(rows).to(dev)].cpu().numpy(), ref) <= TOL
        assert torch.equal(ops.spmm_lds(A, B), c)
        assert torch.equal(B, Bfull[:, :d]) and float(Bfull[:, d:].abs().max()) == 0.0
        print("LDS sweep, full size%s: %.1f %% of the nonzeros planned, %.1f nonzeros per staged piece"
              % (" (transpose)" if want_fold else "", 100.0 * A.host_stats["local_nnz"] / m.nnz, A.host_stats["reuse"]))


def test_column_sweep_lost_lock_guard_retunes(dev):
    """The clock-paced sweep watches itself: with a pace that is deliberately too fast (the lock-step is lost, the
    product costs ~2x) two timed samples in a row exceed 1.3 x the tuned time and the plan re-tunes on the spot; a
    healthy plan is never re-tuned and the guard leaves its results alone."""
    from stochastic_gcn_amd import ops, synthetic
    n, _, full_adj, *_ = synthetic.reddit_like(with_features=False)
    d = 602
    g = torch.Generator(device=dev); g.manual_seed(1)
    B = torch.zeros((n, 608), device=dev)
    B[:, :d] = torch.randn((n, d), device=dev, generator=g)
    Bd = B[:, :d]
    A = ops.ColumnSweepCSR(full_adj, dev, G=2)
    t_tuned, pace = A.autotune(Bd)
    assert pace > 0 and A.tuned_ms[d] == t_tuned
    ref = ops.spmm_cs(A, Bd).clone()

    def run(k):
        for _ in range(k):
            out = ops.spmm_cs(A, Bd)
            torch.cuda.synchronize()            # (the guard never waits: let its samples complete between calls)
        return out
    run(4 * A.GUARD_EVERY)
    assert A._guard[d]["retunes"] == 0 and A._guard[d]["last_ms"] is not None       # healthy: sampled, left alone
    assert A._guard[d]["last_ms"] <= 1.25 * t_tuned
    A.pace[d] = max(60, pace // 2)                  # far too fast: every wave runs ahead of the window
    slow = run(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.spmm_cs(A, Bd); e1.record(); e1.synchronize()
    assert e0.elapsed_time(e1) > 1.3 * t_tuned, "the test needs a pace that actually loses the lock-step"
    assert torch.equal(slow, ref)                   # (the pace never changes the result)
    run(3 * A.GUARD_EVERY)
    assert A._guard[d]["retunes"] == 1              # two slow samples in a row -> one re-tune
    assert abs(A.pace[d] - pace) <= 0.25 * pace
    e0.record(); out = ops.spmm_cs(A, Bd); e1.record(); e1.synchronize()
    assert e0.elapsed_time(e1) <= 1.2 * t_tuned and torch.equal(out, ref)


def test_cached_column_sweep_pace_comes_back_with_its_time_and_arms_the_guard(dev, tmp_path):
    """a pace restored from the plan cache was tuned on whatever box wrote the file -- exactly the case the
    lost-lock guard exists for -- so the file carries the product's time at that pace and the guard samples the very
    first products of a loaded plan; a file that holds paces without times (an older build's) gives its paces up."""
    from stochastic_gcn_amd import ops, synthetic
    _, _, a, *_ = synthetic.reddit_like(n=40000, m=1500000, f=8, classes=5, splits=(30000, 4000, 6000), seed=3,
                                        with_features=False)
    d = 256
    g = torch.Generator(device=dev); g.manual_seed(2)
    B = torch.randn((a.shape[0], d), device=dev, generator=g)
    path = str(tmp_path / "g.csplan.npz")
    A, hit = ops.ColumnSweepCSR.cached(a, dev, path, G=2)
    assert not hit
    t_tuned, pace = A.autotune(B)
    if pace <= 0:                                   # a graph this small may run best unpaced: give it a (slow, safe) clock
        A.pace[d] = pace = 400
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.spmm_cs(A, B); e0.record(); ops.spmm_cs(A, B); e1.record(); e1.synchronize()
        A.tuned_ms[d] = t_tuned = e0.elapsed_time(e1)
    A.store_if_cached()
    ref = ops.spmm_cs(A, B).clone()
    A2, hit = ops.ColumnSweepCSR.cached(a, dev, path, G=2)
    assert hit and A2.pace == A.pace and A2.tuned_ms == {d: t_tuned}
    for _ in range(2 * A2.GUARD_EVERY):
        out = ops.spmm_cs(A2, B)
        torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert A2._guard[d]["last_ms"] is not None and A2._guard[d]["retunes"] == 0        # armed, sampled, healthy
    # an older file: paces, no times
    z = dict(np.load(path))
    z.pop("tuned_ms")
    np.savez(path, **z)
    A3, hit = ops.ColumnSweepCSR.cached(a, dev, path, G=2)
    assert hit and d not in A3.pace and not A3.tuned_ms
    # a launch that fails between the guard's two events leaves no half-recorded sample behind
    torch.cuda.synchronize()
    A2._guard_after(d, B)                           # (reads the last finished sample: nothing pending now)
    assert A2._guard[d]["pending"] is None
    A2._guard[d]["calls"] = A2.GUARD_EVERY - 1      # the next product is a sampled one
    bad = torch.empty((a.shape[0], d + 1), device=dev)[:, :d]        # row pitch 257 floats: refused by the foreign call
    with pytest.raises(Exception, match="aligned"):
        ops.spmm_cs(A2, B, out=bad)
    assert A2._guard[d]["pending"] is None
    ops.spmm_cs(A2, B)
