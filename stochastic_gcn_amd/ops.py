"""Device ops of the hot path: thin wrappers that hand raw HBM pointers of PyTorch-ROCm tensors
to the C-ABI of libsgcn.so (include/sgcn.h).  PyTorch is used for device memory and streams
only; every op here is a hand-written HIP kernel and raises if the library is unavailable or
a tensor is not on a GPU (no eager / CPU fallback).

Reference seams replaced:
  spmm            dot(x, y, sparse=True)                    gcn/layers.py:31-37
  vr_aggregate    VRAggregator._call                        gcn/layers.py:298-319,350-362
  gather_rows     history.dense_slice / tf.gather           gcn/_history.pyx:53-62, layers.py:304
  scatter_rows    tf.scatter_update                         gcn/models.py:165
  csr_slice       history.slice                             gcn/_history.pyx:25-51
  dense_fwd / dense_bwd / gemm / dropout   the dense layers around the aggregators (dot + MyLayerNorm
                  + relu + tf.nn.dropout and their autodiff) gcn/layers.py:87-138,365-433
  spmm_cs         the same product as spmm for a STATIC graph (column sweep, ColumnSweepCSR)
"""
import ctypes as C
import time

import numpy as np
import torch

from . import _ffi
from ._ffi import check, lib
from .scheduler import build_plan


_STREAM_CACHE = None


def _stream():
    """Raw hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~9 us of
    Python per call (15 calls per training step), so a step brackets its launches with
    pin_stream()/unpin_stream() and the handle is looked up once."""
    if _STREAM_CACHE is not None:
        return _STREAM_CACHE
    return torch.cuda.current_stream().cuda_stream


def pin_stream():
    global _STREAM_CACHE
    _STREAM_CACHE = torch.cuda.current_stream().cuda_stream


def unpin_stream():
    global _STREAM_CACHE
    _STREAM_CACHE = None


def _dev(t, dtype, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must live in HBM (got a %s tensor): the SpMM/history path has no "
                           "CPU fallback" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _rows2d(t, name):
    """(ptr, ld) of a 2-D fp32 tensor whose rows are contiguous (row pitch may exceed width)."""
    _dev(t, torch.float32, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("%s must be 2-D with unit column stride" % name)
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


class DevArray(object):
    """A typed window ``base[off : off + n]`` of a 1-D device buffer that is NOT a tensor: the ops
    only need its address and length, and a torch slice costs ~2.5 us of host time -- a packed
    minibatch has ~25 such windows, i.e. a tenth of the launch-bound training step.  ``tensor()``
    materialises the view for the rare consumer that wants one."""
    __slots__ = ("base", "off", "n", "dtype", "_ptr")
    is_cuda = True

    def __init__(self, base, base_ptr, off, n):
        self.base, self.off, self.n, self.dtype = base, off, n, base.dtype
        self._ptr = base_ptr + 4 * off          # int32 / fp32 elements

    def data_ptr(self):
        return self._ptr

    @property
    def shape(self):
        return (self.n,)

    @property
    def device(self):
        return self.base.device

    def numel(self):
        return self.n

    def tensor(self):
        return self.base[self.off:self.off + self.n]


def as_tensor(x):
    return x.tensor() if isinstance(x, DevArray) else x


class DevicePlan(object):
    """Device copy of a host work plan + its partial-sum workspace."""

    def __init__(self, seg, fix, nslots, device):
        self.nseg, self.nfix, self.nslots = int(seg.shape[0]), int(fix.shape[0]), int(nslots)
        self.seg = torch.from_numpy(seg).to(device, non_blocking=True)
        self.fix = torch.from_numpy(fix).to(device, non_blocking=True) if self.nfix else None
        self.ws = None
        self.device = device

    def struct(self, d):
        ldw = (d + 3) // 4 * 4
        need = self.nslots * ldw
        if need and (self.ws is None or self.ws.numel() < need):
            self.ws = torch.empty(need, dtype=torch.float32, device=self.device)
        return _ffi.Plan(self.seg.data_ptr(), self.nseg, _ptr(self.fix), self.nfix, self.nslots,
                         _ptr(self.ws), 0 if self.ws is None else self.ws.numel())


class DeviceCSR(object):
    """CSR matrix resident in HBM (int32 rowptr/col, fp32 val) + optional plan / transpose."""

    def __init__(self, shape, rowptr, col, val, plan=None, transpose=None, host_rowptr=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self.rowptr, self.col, self.val = rowptr, col, val
        self.plan, self.transpose, self.host_rowptr = plan, transpose, host_rowptr

    @property
    def nnz(self):
        return int(self.col.shape[0])

    @staticmethod
    def from_arrays(shape, rowptr, col, val, device, plan_T=0, with_plan=True):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        plan = None
        if with_plan:
            seg, fix, nslots = build_plan(rowptr, plan_T)
            plan = DevicePlan(seg, fix, nslots, device)
        return DeviceCSR(shape, torch.from_numpy(rowptr).to(device, non_blocking=True),
                         torch.from_numpy(col).to(device, non_blocking=True),
                         torch.from_numpy(val).to(device, non_blocking=True), plan,
                         host_rowptr=rowptr)

    @staticmethod
    def from_scipy(a, device, plan_T=0, with_plan=True, with_transpose=False):
        a = a.tocsr()
        m = DeviceCSR.from_arrays(a.shape, a.indptr, a.indices, a.data, device, plan_T, with_plan)
        if with_transpose:
            at = transpose_host(a)
            m.transpose = DeviceCSR.from_arrays(at.shape, at.indptr, at.indices, at.data, device,
                                                plan_T, with_plan)
        return m

    @staticmethod
    def from_host(h, device, plan_T=0, with_plan=True):
        """From a scheduler.HostCSR bundle (adds the transposed CSR when the sampler built it)."""
        m = DeviceCSR.from_arrays(h.shape, h.rowptr, h.col, h.val, device, plan_T, with_plan)
        if h.t_rowptr is not None:
            m.transpose = DeviceCSR.from_arrays((h.shape[1], h.shape[0]), h.t_rowptr, h.t_col,
                                                h.t_val, device, plan_T, with_plan)
        return m


def spmm(A, B, out=None, gidx=None, rscale=None, cscale=None, beta=0.0, d=None, add=None, add_rows=0):
    """out[M x d] = rscale (.) (A (cscale (.) B[gidx])) + beta * out   (sgcn_spmm_csr_f32);
    with ``add``: out[i] += add[i] for i < add_rows in the same launch (sgcn_spmm_csr_add_f32)."""
    M, K = A.shape
    bptr, ldb = _rows2d(B, "B")
    d = int(B.shape[1] if d is None else d)
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an existing `out`")
        out = torch.empty((M, d), dtype=torch.float32, device=B.device)
    cptr, ldc = _rows2d(out, "out")
    if out.shape[0] != M or out.shape[1] < d:
        raise ValueError("out has shape %s, need (%d, >=%d)" % (tuple(out.shape), M, d))
    rows_needed = K if gidx is None else None
    if rows_needed is not None and B.shape[0] < rows_needed:
        raise ValueError("B has %d rows, A has %d columns" % (B.shape[0], K))
    plan = A.plan.struct(d) if A.plan is not None else None
    if add is not None:
        aptr, ldadd = _rows2d(add, "add")
        check(lib.sgcn_spmm_csr_add_f32(
            A.rowptr.data_ptr(), _ptr(A.col), _ptr(A.val), M, K, d, bptr, ldb,
            _ptr(_dev(gidx, torch.int32, "gidx")), _ptr(_dev(rscale, torch.float32, "rscale")),
            _ptr(_dev(cscale, torch.float32, "cscale")), cptr, ldc, float(beta),
            C.byref(plan) if plan is not None else None, aptr, ldadd, int(add_rows), _stream()))
        return out
    check(lib.sgcn_spmm_csr_f32(
        A.rowptr.data_ptr(), _ptr(A.col), _ptr(A.val), M, K, d, bptr, ldb,
        _ptr(_dev(gidx, torch.int32, "gidx")), _ptr(_dev(rscale, torch.float32, "rscale")),
        _ptr(_dev(cscale, torch.float32, "cscale")), cptr, ldc, float(beta),
        C.byref(plan) if plan is not None else None, _stream()))
    return out


def vr_aggregate(A, P, h, mu, Hbar, ifield, ffield, s, cvd, concat_self, out_h=None, out_mu=None):
    """Fused control-variate aggregator forward (sgcn_vr_aggregate_f32)."""
    n1, n0 = A.shape
    nf = P.shape[1]
    hptr, ldx = _rows2d(h, "h")
    d = int(h.shape[1])
    width = 2 * d if concat_self else d
    if cvd:
        mptr, ldm = _rows2d(mu, "mu")
        if ldm != ldx or tuple(mu.shape) != tuple(h.shape):
            raise ValueError("h and mu must share shape and row pitch")
    else:
        mptr = None
    Hptr, ldh = _rows2d(Hbar, "Hbar")
    if out_h is None:
        out_h = torch.empty((n1, width), dtype=torch.float32, device=h.device)
    if cvd and out_mu is None:
        out_mu = torch.empty((n1, width), dtype=torch.float32, device=h.device)
    ohp, ldo = _rows2d(out_h, "out_h")
    omp = None
    if cvd:
        omp, ldo2 = _rows2d(out_mu, "out_mu")
        if ldo2 != ldo:
            raise ValueError("out_h and out_mu must share row pitch")
    plan = P.plan.struct(d) if P.plan is not None else None
    check(lib.sgcn_vr_aggregate_f32(
        A.rowptr.data_ptr(), _ptr(A.col), _ptr(A.val), P.rowptr.data_ptr(), _ptr(P.col),
        _ptr(P.val), n1, n0, nf, d, hptr, mptr, ldx, Hptr, ldh,
        _ptr(_dev(ifield, torch.int32, "ifield")), _ptr(_dev(ffield, torch.int32, "ffield")),
        _ptr(_dev(s, torch.float32, "s")), ohp, omp, ldo, int(bool(cvd)), int(bool(concat_self)),
        C.byref(plan) if plan is not None else None, _stream()))
    return out_h, out_mu


def vr_aggregate_two_phase(A, P, h, mu, Hbar, ifield, ffield, s, cvd, concat_self, stream_pre=None):
    """The same aggregate as ``vr_aggregate`` through sgcn_vr_aggregate_pre_f32 + _post_f32 (what the step
    program issues, the first phase beside the dense layers): bit-identical to the fused call."""
    n1, n0 = A.shape
    nf = P.shape[1]
    hptr, ldx = _rows2d(h, "h")
    d = int(h.shape[1])
    width = 2 * d if concat_self else d
    Hptr, ldh = _rows2d(Hbar, "Hbar")
    ldw = (d + 3) // 4 * 4
    accP = torch.empty((n1, ldw), dtype=torch.float32, device=h.device)
    plan = P.plan.struct(d) if P.plan is not None else None
    check(lib.sgcn_vr_aggregate_pre_f32(P.rowptr.data_ptr(), _ptr(P.col), _ptr(P.val), n1, nf, d, Hptr, ldh,
                                        _ptr(_dev(ffield, torch.int32, "ffield")), accP.data_ptr(),
                                        C.byref(plan) if plan is not None else None, _stream()))
    out_h = torch.empty((n1, width), dtype=torch.float32, device=h.device)
    out_mu = torch.empty((n1, width), dtype=torch.float32, device=h.device) if cvd else None
    check(lib.sgcn_vr_aggregate_post_f32(A.rowptr.data_ptr(), _ptr(A.col), _ptr(A.val), n1, n0, d, hptr,
                                         _rows2d(mu, "mu")[0] if cvd else None, ldx, Hptr, ldh,
                                         _ptr(_dev(ifield, torch.int32, "ifield")), _ptr(_dev(s, torch.float32, "s")),
                                         out_h.data_ptr(), _ptr(out_mu), width, int(bool(cvd)), int(bool(concat_self)),
                                         accP.data_ptr(), _stream()))
    return out_h, out_mu


def gather_rows(inp, idx, out=None, d=None):
    """out[i, :d] = inp[idx[i], :d]   (sgcn_gather_rows_f32)."""
    iptr, ldi = _rows2d(inp, "inp")
    _dev(idx, torch.int32, "idx")
    n = int(idx.shape[0])
    d = int(inp.shape[1] if d is None else d)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=inp.device)
    optr, ldo = _rows2d(out, "out")
    check(lib.sgcn_gather_rows_f32(iptr, ldi, idx.data_ptr(), n, d, optr, ldo, _stream()))
    return out


def scatter_rows(H, idx, src, d=None):
    """H[idx[i], :d] = src[i, :d]  (idx unique)   (sgcn_scatter_rows_f32)."""
    hptr, ldh = _rows2d(H, "H")
    sptr, lds = _rows2d(src, "src")
    _dev(idx, torch.int32, "idx")
    n = int(idx.shape[0])
    d = int(src.shape[1] if d is None else d)
    check(lib.sgcn_scatter_rows_f32(hptr, ldh, idx.data_ptr(), n, d, sptr, lds, _stream()))
    return H


def csr_slice(A, rows_host, rows_dev=None, with_coo_rows=False):
    """CSR row slice A[rows] -> DeviceCSR   (sgcn_csr_slice_indptr + sgcn_csr_slice_f32).

    ``A.host_rowptr`` (kept by DeviceCSR.from_*) feeds the host prefix pass, mirroring the
    reference's two-phase c_indptr / c_slice (gcn/history.cpp:50-72)."""
    if A.host_rowptr is None:
        raise ValueError("csr_slice needs A.host_rowptr")
    rows_host = np.ascontiguousarray(rows_host, dtype=np.int32)
    n = int(rows_host.shape[0])
    o_p = np.empty(n + 1, dtype=np.int32)
    check(lib.sgcn_csr_slice_indptr(n, rows_host.ctypes.data, A.host_rowptr.ctypes.data,
                                    o_p.ctypes.data))
    nnz = int(o_p[n])
    dev = A.val.device
    if rows_dev is None:
        rows_dev = torch.from_numpy(rows_host).to(dev, non_blocking=True)
    o_p_dev = torch.from_numpy(o_p).to(dev, non_blocking=True)
    o_d = torch.empty(nnz, dtype=torch.float32, device=dev)
    o_c = torch.empty(nnz, dtype=torch.int32, device=dev)
    o_r = torch.empty(nnz, dtype=torch.int32, device=dev) if with_coo_rows else None
    check(lib.sgcn_csr_slice_f32(n, rows_dev.data_ptr(), A.val.data_ptr(), A.col.data_ptr(),
                                 A.rowptr.data_ptr(), o_p_dev.data_ptr(), o_d.data_ptr(),
                                 o_c.data_ptr(), _ptr(o_r), _stream()))
    out = DeviceCSR((n, A.shape[1]), o_p_dev, o_c, o_d, host_rowptr=o_p)
    out.coo_rows = o_r
    return out


# ---- fused dense-side kernels (sgcn_dense.hip) ---------------------------------------------------
def ln_act_fwd(x, offset, scale, relu, eps=1e-9):
    """y = act(LN(x)*scale + offset) (norm skipped when offset/scale are None).
    Returns (y, ctx) with ctx = (xhat, rstd) or None."""
    xp, ldx = _rows2d(x, "x")
    n, d = int(x.shape[0]), int(x.shape[1])
    y = torch.empty((n, d), dtype=torch.float32, device=x.device)
    norm = offset is not None
    xhat = torch.empty((n, d), dtype=torch.float32, device=x.device) if norm else None
    rstd = torch.empty((n,), dtype=torch.float32, device=x.device) if norm else None
    check(lib.sgcn_ln_act_fwd_f32(xp, ldx, _ptr(offset), _ptr(scale), n, d, float(eps), int(bool(relu)),
                                  y.data_ptr(), d, _ptr(xhat), _ptr(rstd), _stream()))
    return y, ((xhat, rstd) if norm else None)


def ln_act_bwd(dy, y, ctx, scale, relu, doffset=None, dscale=None):
    """dx of ln_act_fwd; accumulates the LN parameter gradients into doffset/dscale."""
    gp, ldg = _rows2d(dy, "dy")
    yp, ldy = _rows2d(y, "y")
    n, d = int(dy.shape[0]), int(dy.shape[1])
    dx = torch.empty((n, d), dtype=torch.float32, device=dy.device)
    norm = ctx is not None
    ws = _gemm_ws(int(lib.sgcn_ln_act_bwd_ws_floats(n, d)), dy.device) if norm else None   # shared scratch
    check(lib.sgcn_ln_act_bwd_f32(gp, ldg, yp, ldy, _ptr(ctx[0]) if norm else None,
                                  _ptr(ctx[1]) if norm else None, _ptr(scale) if norm else None, n, d,
                                  int(bool(relu)), dx.data_ptr(), d, _ptr(doffset), _ptr(dscale),
                                  _ptr(ws), _stream()))
    return dx


def softmax_ce(logits, labels, want_grad=True, want_pred=False):
    """(stats[4] = {sum CE, #correct, mean CE, accuracy}, dlogits or None, pred or None)
    (sgcn_softmax_ce_f32)."""
    zp, ldz = _rows2d(logits, "logits")
    lp, ldl = _rows2d(labels, "labels")
    n, c = int(logits.shape[0]), int(logits.shape[1])
    # [stats | per-row scratch | with pred: the rows' classes, argmax(pred) + 4096 * argmax(labels)]
    stats = torch.empty(4 + (3 if want_pred else 2) * n, dtype=torch.float32, device=logits.device)
    dz = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_grad else None
    pred = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_pred else None
    check(lib.sgcn_softmax_ce_f32(zp, ldz, lp, ldl, n, c, _ptr(dz), c, _ptr(pred), c, stats.data_ptr(),
                                  stats.data_ptr() + 16, _stream()))
    return stats, dz, pred


def sigmoid_ce(logits, labels, want_grad=True, want_pred=False):
    """Multitask (ppi) loss: (stats[4] = {sum CE, #correct elements, mean CE, element accuracy}, dlogits
    or None, pred = sigmoid(logits) or None)   (sgcn_sigmoid_ce_f32)."""
    zp, ldz = _rows2d(logits, "logits")
    lp, ldl = _rows2d(labels, "labels")
    n, c = int(logits.shape[0]), int(logits.shape[1])
    stats = torch.empty(4 + 2 * n, dtype=torch.float32, device=logits.device)
    dz = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_grad else None
    pred = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_pred else None
    check(lib.sgcn_sigmoid_ce_f32(zp, ldz, lp, ldl, n, c, _ptr(dz), c, _ptr(pred), c, stats.data_ptr(),
                                  stats.data_ptr() + 16, _stream()))
    return stats, dz, pred


def l2_penalty(theta, lo, hi, wd, grad=None, loss=None):
    """Weight decay over the flat-buffer range [lo, hi): grad += wd * theta, loss[0] += 0.5 * wd * |theta|^2
    (either may be None)   (sgcn_l2_penalty_f32)."""
    _dev(theta, torch.float32, "theta")
    check(lib.sgcn_l2_penalty_f32(theta.data_ptr(), int(lo), int(hi), float(wd), _ptr(grad), _ptr(loss), _stream()))


def csr_transpose_index(A):
    """(t_rowptr, t_row, t_src) of a row-sliced DeviceCSR with ``coo_rows`` (sgcn_csr_transpose_index):
    the structure of A^T, rows ascending inside every column, + the source position of each entry."""
    if getattr(A, "coo_rows", None) is None:
        raise ValueError("csr_transpose_index needs the COO row ids (csr_slice(..., with_coo_rows=True))")
    dev = A.col.device
    ncols, nnz = int(A.shape[1]), int(A.col.shape[0])
    t_rowptr = torch.empty(ncols + 1, dtype=torch.int32, device=dev)
    t_row = torch.empty(nnz, dtype=torch.int32, device=dev)
    t_src = torch.empty(nnz, dtype=torch.int32, device=dev)
    need = int(lib.sgcn_csr_transpose_ws_ints(ncols, nnz))
    ws = torch.empty(max(need, 1), dtype=torch.int32, device=dev)
    check(lib.sgcn_csr_transpose_index(ncols, nnz, _ptr(A.col), _ptr(A.coo_rows), t_rowptr.data_ptr(),
                                       t_row.data_ptr(), t_src.data_ptr(), ws.data_ptr(), _stream()))
    return t_rowptr, t_row, t_src


def gather_f32(src, idx):
    """out[i] = src[idx[i]]   (sgcn_gather_f32)."""
    _dev(src, torch.float32, "src")
    _dev(idx, torch.int32, "idx")
    out = torch.empty(idx.shape[0], dtype=torch.float32, device=src.device)
    check(lib.sgcn_gather_f32(src.data_ptr(), idx.data_ptr(), int(idx.shape[0]), out.data_ptr(), _stream()))
    return out


def adam_step(theta, grad, m, v, lr_t, beta1, beta2, eps=1e-8):
    check(lib.sgcn_adam_f32(theta.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(),
                            int(theta.numel()), float(lr_t), float(beta1), float(beta2), float(eps),
                            _stream()))


# ---- column-sweep SpMM for static graphs (sgcn_spmm_cs.hip) --------------------------------------
def transpose_host(a, threads=0):
    """A^T of a SciPy CSR as a SciPy CSR with sorted rows, by the library's parallel counting sort
    (sgcn_csr_transpose_host): what the plan of a backward product (the autodiff of gcn/layers.py:31-37) is built from.
    Same arrays as ``a.T.tocsr()`` (rows ascend inside a column), in a fraction of its single-threaded time."""
    import scipy.sparse as sp
    a = a.tocsr()
    M, K = int(a.shape[0]), int(a.shape[1])
    rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
    col = np.ascontiguousarray(a.indices, dtype=np.int32)
    val = np.ascontiguousarray(a.data, dtype=np.float32)
    trp = np.empty(K + 1, dtype=np.int32)
    tc = np.empty(col.shape[0], dtype=np.int32)
    tv = np.empty(col.shape[0], dtype=np.float32)
    check(lib.sgcn_csr_transpose_host(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, K, int(threads),
                                      trp.ctypes.data, tc.ctypes.data, tv.ctypes.data))
    t = sp.csr_matrix((tv, tc, trp), shape=(K, M), copy=False)
    t.has_sorted_indices = True
    return t


def reorder_labels(a, max_iters=0, seed=1, min_size=64):
    """Community label per vertex of a square sparse matrix's pattern (sgcn_reorder_lp: seeded
    asynchronous label propagation on the host, graph only).  Returns (comm int32[n], ncomm)."""
    a = a.tocsr()
    if a.shape[0] != a.shape[1]:
        raise ValueError("reorder_labels needs a square (vertex x vertex) matrix")
    rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
    col = np.ascontiguousarray(a.indices, dtype=np.int32)
    comm = np.empty(a.shape[0], dtype=np.int32)
    nc = C.c_int32()
    check(lib.sgcn_reorder_lp(rowptr.ctypes.data, col.ctypes.data, int(a.shape[0]), int(max_iters), int(seed),
                              int(min_size), comm.ctypes.data, C.byref(nc)))
    return comm, int(nc.value)


class ColumnSweepCSR(object):
    """A static CSR re-laid for the column sweep (include/sgcn.h sgcn_csplan_t): built once on the
    host (full-graph / PP products), then multiplied many times.

    ``col_labels`` / ``row_labels`` (community label per column vertex / per row, e.g. from
    ``reorder_labels``): a locality-preserving plan for graphs that HAVE communities.  Columns are
    renumbered so that a community's vertices are contiguous (the sweep order; B itself is not
    moved -- the kernel reads B rows through the position -> vertex map), tiles are formed inside
    row communities in community order, a launch's consecutive tiles go to the same XCD, and the
    sweep runs unpaced (the waves of an XCD then work on the same few communities' B rows, which is
    what the clock pacing provides on a graph without structure).  Results equal the unlabelled
    plan's up to fp32 summation order."""

    WARP_BUCKETS = 16384      # the warp table's size bound (64 KiB: scalar cache / L2 resident)
    WARP_AUTO_DEV = 0.01      # 'auto': a table is kept when some column's share of the work in front of it is off its
                              # share of the ids by more than this (1 % of the sweep ~ a third of an L2 window)

    @classmethod
    def make_warp(cls, cols, K, mode='auto'):
        """The sweep clock in work coordinates (include/sgcn.h sgcn_csplan_t.dev_warp): ``(table, shift)`` with
        table[b] = share of the nonzeros in columns < (b << shift), scaled to [0, K) -- or ``(None, 0)`` when the nonzeros
        are spread evenly enough over the column ids for the linear clock (S-Reddit: hubs carry random ids), or
        ``mode`` is False.  R-MAT is the case it exists for: a third of the nonzeros in the first sixteenth of the ids."""
        K = int(K)
        if mode is False or K <= 0 or cols.shape[0] == 0:
            return None, 0
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        table = np.empty(cls.WARP_BUCKETS, dtype=np.uint32)
        nb, shift = C.c_int32(), C.c_int32()
        check(lib.sgcn_cs_warp_table(cols.ctypes.data, cols.shape[0], K, cls.WARP_BUCKETS, 0 if mode == 'auto' else 1,
                                     cls.WARP_AUTO_DEV, 0, table.ctypes.data, C.byref(nb), C.byref(shift)))
        if nb.value == 0:
            return None, 0
        return np.ascontiguousarray(table[:nb.value]), int(shift.value)

    def __init__(self, a, device, R=16, T=0, round_tiles=0, col_labels=None, row_labels=None, G=1, align='auto', warp='auto',
                 col_ranges=0):
        """G = 2: two 16-row lane groups per wavefront on 128-column passes (sgcn_csplang_*), half the passes of
        the dense operand through every XCD per register byte; ``align``: columns one bin of a wave may run ahead
        of the slowest (the plan pads the bins that are ahead; 'auto': a third of the L2 window, ``auto_align``; with a
        warp table -- ``warp``, ``make_warp`` -- it counts sweep positions).  What ``choose_g(d)`` picks for most widths -- the
        bench and the training path call it: S-Reddit d = 602 3.51 vs 3.67 ms sustained, S-RMAT d = 256 2.61 vs
        2.82 ms.  G = 4 (four groups, 64-column passes, every row of a 233 k-row graph resident in one round): instruction-
        bound on a full graph (3.75 ms on S-Reddit; its gathers alone 3.21: DESIGN.md 3.2), but the better sweep for a SPARSE
        matrix such as the LDS sweep's residual, which is what uses it."""
        a = a.tocsr()
        self.G = int(G)
        self.ranged = 0
        if R != 16:
            raise ValueError("the column-sweep kernels keep 16-row bins (R = 16)")
        if self.G not in (1, 2, 4):
            raise ValueError("G must be 1, 2 or 4 lane groups per wavefront")
        if col_ranges and int(col_ranges) > 1:
            if self.G != 1 or col_labels is not None or row_labels is not None:
                raise ValueError("col_ranges needs an unlabelled plan with one lane group per wavefront")
            self._init_ranged(a, device, int(col_ranges), T, round_tiles)
            return
        if self.G != 1:
            if col_labels is not None or row_labels is not None or R != 16:
                raise ValueError("G = 2 plans are ungrouped and use 16-row bins")
            self._init_g2(a, device, T, round_tiles, align, warp)
            return
        rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
        col = np.ascontiguousarray(a.indices, dtype=np.int32)
        val = np.ascontiguousarray(a.data, dtype=np.float32)
        M = rowptr.shape[0] - 1
        self.grouped = col_labels is not None or row_labels is not None
        self.pos2col = None
        if col_labels is not None:
            col_labels = np.ascontiguousarray(col_labels, dtype=np.int32)
            if col_labels.shape[0] != a.shape[1]:
                raise ValueError("col_labels must have one label per column")
            pos2col = np.argsort(col_labels, kind='stable').astype(np.int32)     # sweep position -> vertex
            col2pos = np.empty_like(pos2col)
            col2pos[pos2col] = np.arange(pos2col.shape[0], dtype=np.int32)
            col = np.ascontiguousarray(col2pos[col])                              # the plan lives in positions
            self.pos2col = torch.from_numpy(pos2col).to(device)
        rg = None
        if row_labels is not None:
            rg = np.ascontiguousarray(row_labels, dtype=np.int32)
            if rg.shape[0] != M:
                raise ValueError("row_labels must have one label per row")
        rgp = rg.ctypes.data if rg is not None else None
        if not T and not self.grouped:
            T = self.auto_t(rowptr, 1, int(round_tiles or (_ffi.lib.sgcn_tune_get(b"cs_round") or 4096)))
        self.T = int(T)
        t_build = time.perf_counter()
        tile_ptr, colrow, valout, tile_rows, tile_slots, fix, nt, nfix, nslots = self._build(
            rowptr, col, val, M, 1, R, T, 0, 0, rgp, None, 0)
        t_build = time.perf_counter() - t_build
        self.shape = (int(a.shape[0]), int(a.shape[1]))
        self.R, self.ntiles, self.nfix, self.nslots = R, nt, nfix, nslots
        self.round_tiles = round_tiles
        self._tile_nnz = np.diff(tile_ptr).astype(np.int64)
        self._hint, self._hint_round = None, None
        self.pace = {}          # d -> ns per nonzero of the heaviest tile (autotuned), -1 = unpaced
        self.tuned_ms, self._guard, self._tuning = {}, {}, False
        t_up = time.perf_counter()
        to = lambda x: torch.from_numpy(x).to(device)          # noqa: E731
        self.tile_ptr, self.colrow, self.val = to(tile_ptr), to(colrow), to(valout)
        self.tile_rows, self.tile_slots = to(tile_rows), to(tile_slots)
        self.fix = to(fix) if nfix else None
        self.ws, self.device = None, device
        self.nnz = int(col.shape[0])
        # the clock's coordinates (grouped plans run unpaced: no table)
        wtab, self.warp_shift = (None, 0) if self.grouped else self.make_warp(col, self.shape[1], warp)
        self.warp = None if wtab is None else torch.from_numpy(wtab.view(np.int32)).to(device)
        # what the plan cost to set up (the product it serves may run once: gcn/utils.py:321-322)
        self.setup_s = {"host_plan_s": t_build, "upload_s": time.perf_counter() - t_up,
                        "host_threads": int(lib.sgcn_host_threads())}

    @staticmethod
    def auto_align(K, nnz, M, G, rnd, d_piece_bytes=None):
        """How far (in sweep positions) one bin of a wave may run ahead of the slowest: a third of the L2 window.  An
        XCD's 4 MiB hold 4 MiB / piece bytes of B (8,192 pieces of a 128-column pass); the window is that share of the
        nonzeros an XCD gathers in one round of resident tiles, in positions.  S-Reddit: 1,300 positions -> the measured
        default of 2,048 stays; a sparse block of a large graph (S-RMAT 10 M: 24.6 M nonzeros over 10 M columns in ten
        rounds) gets ~90,000 -- aligned to 2,048 COLUMNS its bins were mostly pads."""
        piece = d_piece_bytes or (512 if G == 2 else 256)
        rows_round = max(rnd * 16 * G, 1)
        rounds = max(-(-int(M) // rows_round), 1)
        nnz_xcd_round = max(int(nnz) / rounds / 8.0, 1.0)
        window = float(K) * ((4 << 20) / piece) / nnz_xcd_round
        return int(max(2048, min(0.35 * window, K / 8.0)))

    @staticmethod
    def auto_t(rowptr, G, rnd):
        """The split threshold for a matrix that does not fill its rounds of resident tiles: rows longer than T become
        strided virtual rows, and a block with few rows (an eighth of S-Reddit: 29 k rows against the 65 k / 131 k a round
        holds) leaves most wavefront slots of its launches empty -- its sweep is latency-bound.  0 = the library's default
        (8 x the mean degree, sgcn_csplan.cpp default_t) when the virtual rows it gives fill >= 3/4 of the rounds they
        need; otherwise the smallest T >= 24 whose virtual rows fill 70 % of those rounds (an eighth of S-Reddit, one
        group: T = 400 -> 1.34 ms per fwd + bwd, 100 -> 1.21, 61 (a full round) -> 1.25)."""
        deg = np.diff(np.asarray(rowptr, dtype=np.int64))
        if deg.shape[0] == 0:
            return 0
        avg = int(deg.sum()) // deg.shape[0]
        t0 = int(min(1024, max(64, 8 * avg)))                   # (sgcn_csplan.cpp default_t)
        cap_round = int(rnd) * 16 * max(int(G), 1)

        def vrows(t):
            return int(np.maximum(1, -(-deg // t)).sum())
        v0 = vrows(t0)
        cap = -(-v0 // cap_round) * cap_round
        if v0 >= 0.75 * cap:
            return 0
        lo, hi = 24, t0
        while lo < hi:
            mid = (lo + hi) // 2
            if vrows(mid) <= 0.7 * cap:
                hi = mid
            else:
                lo = mid + 1
        return lo

    # ---- a small row block with its rows split BY COLUMN RANGE (round 6) --------------------------------------------------
    RANGE_FILL = 0.95     # virtual rows of a ranged plan: split until they fill this share of one round of resident tiles

    @staticmethod
    def choose_ranges(M, nnz, K, G, rnd=4096):
        """2 when a block is small enough that its rows are split anyway to fill ONE round of resident tiles (auto_t) and
        the 1-D sweep is bound by B crossing the fabric: an XCD holds M / 8 random rows whose nnz / 8 nonzeros touch
        1 - exp(-nnz / 8K) of B's rows (an eighth of S-Reddit: 79 %, 8 x 0.79 x 561 MB = 3.5 GB for 70 MB of output);
        with the rows cut in two column ranges, range j on XCDs 4j .. 4j + 3, an XCD holds M / 4 half-rows on K / 2
        columns: 2.15 GB.  0 otherwise (a quarter of S-Reddit: 2 x 58 k virtual rows do not fit a round, and two rounds
        sweep every range twice)."""
        if G != 1 or M <= 0 or nnz <= 0 or K < 4096:
            return 0
        if 2 * M > ColumnSweepCSR.RANGE_FILL * int(rnd) * 16:
            return 0
        f1 = 1.0 - np.exp(-nnz / 8.0 / K)
        f2 = 0.5 * (1.0 - np.exp(-nnz / 8.0 / (K / 2.0)))
        return 2 if f2 <= 0.75 * f1 else 0

    def _init_ranged(self, a, device, NR, T, round_tiles):
        """Rows split by column range instead of by stride: piece j of a row = its nonzeros in columns [cut_j, cut_j+1)
        (ranges of equal nonzeros, cut on warp-bucket boundaries), the pieces of range j form the tiles of XCDs
        [8 j / NR, 8 (j + 1) / NR) (tiles in range order + the launch's tile range cut into eight contiguous pieces:
        xcd_map), and ALL ranges are swept at once on one clock: the warp table maps a column to its position inside its
        range, scaled to [0, K).  A row's pieces meet in the ordered fix-up like strided pieces do (range 0 first):
        deterministic, and nothing in the kernels changes.  Built from the shipped plan builder: the NR column-restricted
        copies of the block stacked as NR x M rows with the range as the row label, then rows and workspace slots renamed."""
        import scipy.sparse as sp
        a = a.tocsr()
        if not a.has_sorted_indices:         # (a private copy: the caller's matrix is not reordered behind its back)
            a = a.copy()
            a.sort_indices()
        M, K = int(a.shape[0]), int(a.shape[1])
        rowptr = np.ascontiguousarray(a.indptr, dtype=np.int64)
        col = np.ascontiguousarray(a.indices, dtype=np.int32)
        val = np.ascontiguousarray(a.data, dtype=np.float32)
        rnd = int(round_tiles or (_ffi.lib.sgcn_tune_get(b"cs_round") or 4096))
        shift = 0
        while (K >> shift) > self.WARP_BUCKETS:
            shift += 1
        bucket = 1 << shift
        cum = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=K).astype(np.int64))])
        cuts = [0]
        for j in range(1, NR):
            c = int(np.searchsorted(cum, cum[-1] * j // NR))
            cuts.append(max(cuts[-1], min(K, (c + bucket // 2) // bucket * bucket)))
        cuts = np.asarray(cuts + [K], dtype=np.int64)
        rows_of = np.repeat(np.arange(M, dtype=np.int64), np.diff(rowptr))
        rng_id = np.searchsorted(cuts, col, side="right") - 1
        stacked = sp.csr_matrix((val, (rows_of + rng_id * M, col)), shape=(NR * M, K))      # (columns stay sorted inside a row)
        stacked.sort_indices()
        srp = np.ascontiguousarray(stacked.indptr, dtype=np.int32)
        scol = np.ascontiguousarray(stacked.indices, dtype=np.int32)
        sval = np.ascontiguousarray(stacked.data, dtype=np.float32)
        labels = np.ascontiguousarray(np.repeat(np.arange(NR, dtype=np.int32), M))
        sdeg = np.diff(srp.astype(np.int64))
        if not T:      # split further (by stride, inside a range) until the pieces fill RANGE_FILL of one round
            cap = int(self.RANGE_FILL * rnd * 16)
            pieces = lambda t: int(np.maximum(sdeg > 0, -(-sdeg // t)).sum())      # noqa: E731
            lo, hi = 24, int(max(64, sdeg.max() if sdeg.size else 64))
            if pieces(hi) > cap:
                T = hi
            else:
                while lo < hi:
                    mid = (lo + hi) // 2
                    if pieces(mid) <= cap:
                        hi = mid
                    else:
                        lo = mid + 1
                T = lo
        self.T, self.ranged, self.range_cuts = int(T), NR, [int(x) for x in cuts]
        t_build = time.perf_counter()
        tile_ptr, colrow, valout, tile_rows, tile_slots, fix, nt, nfix, _ = self._build(
            srp, scol, sval, NR * M, 1, 16, self.T, 0, 0, labels.ctypes.data, None, 0)
        # ---- rename: stacked row j * M + r -> row r; its workspace slots -> consecutive slots of row r, range by range
        npieces = (sdeg > 0).astype(np.int64)                      # an empty piece takes no slot (and writes nothing)
        old_first = np.full(NR * M, -1, dtype=np.int64)
        if nfix:
            npieces[fix[:, 0]] = fix[:, 2]
            old_first[fix[:, 0]] = fix[:, 1]
        per_row = npieces.reshape(NR, M)
        total = per_row.sum(axis=0)                                # pieces of original row r
        split = total >= 2
        base = np.zeros(M, dtype=np.int64)
        base[split] = np.cumsum(total[split]) - total[split]
        before = np.cumsum(per_row, axis=0) - per_row              # pieces of row r in the ranges in front of j
        new_first = (base[None, :] + before).reshape(-1)           # first new slot of stacked row j * M + r
        tr, ts = tile_rows.astype(np.int64), tile_slots.astype(np.int64)
        valid = tr >= 0
        srow = np.where(valid, tr, 0)
        q = np.where(ts >= 0, ts - old_first[srow], 0)             # piece index inside the stacked row
        orig = srow % M
        # (an empty piece writes nothing -- except range 0's piece of a row that is empty altogether: it is the row's one
        # writer, C[r] = beta * C[r])
        live = valid & ((sdeg[srow] > 0) | ((srow < M) & (total[orig] == 0)))
        new_rows = np.where(live, orig, -1).astype(np.int32)
        new_slots = np.where(live & split[orig], new_first[srow] + q, -1).astype(np.int32)
        rs = np.nonzero(split)[0]
        new_fix = np.stack([rs, base[rs], total[rs]], axis=1).astype(np.int32) if rs.size else np.zeros((0, 3), np.int32)
        t_build = time.perf_counter() - t_build
        self.grouped, self.pos2col = False, None
        self.shape = (M, K)
        self.R, self.ntiles, self.nfix, self.nslots = 16, nt, int(new_fix.shape[0]), int(total[split].sum())
        self.round_tiles = round_tiles
        self._tile_nnz = np.diff(tile_ptr).astype(np.int64)
        self._hint, self._hint_round = None, None
        self.pace, self.tuned_ms, self._guard, self._tuning = {}, {}, {}, False
        t_up = time.perf_counter()
        to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)          # noqa: E731
        self.tile_ptr, self.colrow, self.val = to(tile_ptr), to(colrow), to(valout)
        self.tile_rows, self.tile_slots = to(new_rows), to(new_slots)
        self.fix = to(new_fix) if self.nfix else None
        self.ws, self.device = None, device
        self.nnz = int(col.shape[0])
        # the clock's coordinates: a column's position INSIDE its range, scaled to [0, K) -- every range starts at 0
        nb = -(-K // bucket)
        first = np.arange(nb, dtype=np.int64) * bucket
        rj = np.searchsorted(cuts, first, side="right") - 1
        lo_, hi_ = cuts[rj], cuts[rj + 1]
        table = ((first - lo_) * K // np.maximum(hi_ - lo_, 1)).astype(np.uint32)
        self.warp, self.warp_shift = to(table.view(np.int32)), shift
        self.setup_s = {"host_plan_s": t_build, "upload_s": time.perf_counter() - t_up,
                        "host_threads": int(lib.sgcn_host_threads())}

    def _init_g2(self, a, device, T, round_tiles, align, warp='auto'):
        """G = 2 / 4 lane groups per wavefront (sgcn_csplang_*)"""
        G = self.G
        rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
        col = np.ascontiguousarray(a.indices, dtype=np.int32)
        val = np.ascontiguousarray(a.data, dtype=np.float32)
        M = rowptr.shape[0] - 1
        rnd = int(round_tiles or (_ffi.lib.sgcn_tune_get(b"cs_round") or 4096))
        if not T:
            T = self.auto_t(rowptr, G, rnd)
        self.T = int(T)
        self.shape = (int(a.shape[0]), int(a.shape[1]))
        # the clock's coordinates first: the bins of a wave are aligned in them
        wtab, self.warp_shift = self.make_warp(col, self.shape[1], warp)
        self.warp = None if wtab is None else torch.from_numpy(wtab.view(np.int32)).to(device)
        wp = wtab.ctypes.data if wtab is not None else None
        if align is None or align == 'auto':
            align = self.auto_align(self.shape[1], col.shape[0], M, G, rnd)
        self.align = align = int(align)
        t_build = time.perf_counter()
        tile_ptr, colrow, valout, tile_rows, tile_slots, fix, nt, nfix, nslots = self._build(
            rowptr, col, val, M, G, 16, T, rnd, align, None, wp, self.warp_shift)
        t_build = time.perf_counter() - t_build
        ne = int(colrow.shape[0])
        self.pad_fraction = 1.0 - col.shape[0] / max(ne, 1)
        self.grouped, self.pos2col = False, None
        self.shape = (int(a.shape[0]), int(a.shape[1]))
        self.R, self.ntiles, self.nfix, self.nslots = 16, nt, nfix, nslots
        self.round_tiles = round_tiles
        self._tile_nnz = (np.diff(tile_ptr) // G).astype(np.int64)       # steps per tile (what the pace counts)
        self._hint, self._hint_round = None, None
        self.pace = {}
        self.tuned_ms, self._guard, self._tuning = {}, {}, False
        t_up = time.perf_counter()
        to = lambda x: torch.from_numpy(x).to(device)          # noqa: E731
        self.tile_ptr, self.colrow, self.val = to(tile_ptr), to(colrow), to(valout)
        self.tile_rows, self.tile_slots = to(tile_rows), to(tile_slots)
        self.fix = to(fix) if nfix else None
        self.ws, self.device = None, device
        self.nnz = int(col.shape[0])
        self.setup_s = {"host_plan_s": t_build, "upload_s": time.perf_counter() - t_up,
                        "host_threads": int(lib.sgcn_host_threads())}

    @staticmethod
    def _build(rowptr, col, val, M, G, R, T, rnd, align, rgp, wp, warp_shift, threads=0):
        """sgcn_csplan_build + export: the plan's host arrays, built in one pass on every core the process may use."""
        h = C.c_void_p()
        check(lib.sgcn_csplan_build(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, G, R, int(T), int(rnd), int(align),
                                    rgp, wp, int(warp_shift), int(threads), C.byref(h)))
        try:
            nt, ne, nfix, nslots = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
            check(lib.sgcn_csbuild_sizes(h, C.byref(nt), C.byref(ne), C.byref(nfix), C.byref(nslots), None, None))
            tile_ptr = np.empty(nt.value + 1, dtype=np.int64)
            colrow = np.empty(ne.value, dtype=np.int32)
            valout = np.empty(ne.value, dtype=np.float32)
            tile_rows = np.empty(nt.value * R * G, dtype=np.int32)
            tile_slots = np.empty(nt.value * R * G, dtype=np.int32)
            fix = np.empty((nfix.value, 3), dtype=np.int32)
            check(lib.sgcn_csbuild_export(h, tile_ptr.ctypes.data, colrow.ctypes.data, valout.ctypes.data,
                                          tile_rows.ctypes.data, tile_slots.ctypes.data,
                                          fix.ctypes.data if nfix.value else None))
        finally:
            lib.sgcn_csbuild_free(h)
        return tile_ptr, colrow, valout, tile_rows, tile_slots, fix, nt.value, nfix.value, nslots.value

    # ---- on-disk plan cache (beside the dataset's .npz) -------------------------
    @staticmethod
    def matrix_key(a):
        """Cheap identity of a CSR (shape, nnz, CRC of its three arrays): a cached plan is reused only
        for the matrix it was built from."""
        import zlib
        a = a.tocsr()
        crc = 0
        for x in (a.indptr, a.indices, a.data):
            crc = zlib.crc32(np.ascontiguousarray(x).view(np.uint8), crc)
        return "%dx%d:%d:%08x" % (a.shape[0], a.shape[1], a.nnz, crc)

    @staticmethod
    def choose_g(d, avg_degree=None, rows=None):
        """1, 2 or 4 lane groups per wavefront for operands of width d: a G = 2 launch is ~0.73 of a G = 1 launch
        (measured on S-Reddit: 0.323 ms with the packed-FMA kernel vs 0.445 ms) and a plan needs half the rounds of resident tiles, but
        ceil(d / 128) passes over the feature dimension instead of ceil(d / 320).  What two groups buy is fewer
        fabric misses.  A SPARSE matrix with more rows than one round of two-group tiles holds (a GPU's block of
        S-RMAT 10 M / 200 M: 1.05 M rows of 23 nonzeros) takes four: twice the rows -- and nonzeros -- per sweep of B, which
        is what its L2 hits come from (3.16 against 3.49 ms, profiles/r43_warp_probe.jsonl; the same rule as the LDS
        plan's residual)."""
        if avg_degree is not None and avg_degree <= 40 and rows is not None and rows > 4096 * 32:
            return 4
        dp = (int(d) + 3) // 4 * 4
        # (dense graphs follow the same rule since the split threshold doubled: S-Reddit-114M, degree 490, d = 602: two groups
        # 13.0 ms, one 14.0 -- with the old threshold 15.1 vs 14.7; the hub block of S-RMAT 10 M, 334 nonzeros per row,
        # d = 256: 2.04 vs 2.59 ms)
        # rounds of resident tiles: one group holds 65,536 rows per round, two hold 131,072 -- half the rounds for a full
        # graph, but no fewer for a block that fits one round either way (an eighth of S-Reddit, 29 k rows, d = 602: two
        # passes of one group 1.34 ms per fwd + bwd, five passes of two groups 1.72)
        r1 = r2 = None
        if rows is not None and rows > 0:
            r1, r2 = -(-int(rows) // (4096 * 16)), -(-int(rows) // (4096 * 32))
        if r1 is None:
            r1, r2 = 2, 1
        return 2 if -(-dp // 128) * 0.73 * r2 <= -(-dp // 320) * r1 else 1

    def save(self, path, key):
        if self.grouped or getattr(self, 'ranged', 0):
            raise ValueError("grouped / column-range plans are not cached (they are cheap to rebuild)")
        t = lambda x: x.cpu().numpy()          # noqa: E731
        blob = dict(key=np.array(key), G=int(getattr(self, 'G', 1)), pad_fraction=float(getattr(self, 'pad_fraction', 0.0)),
                    R=self.R, shape=np.array(self.shape, np.int64), nslots=self.nslots,
                    round_tiles=self.round_tiles, tile_ptr=t(self.tile_ptr), colrow=t(self.colrow), val=t(self.val),
                    tile_rows=t(self.tile_rows), tile_slots=t(self.tile_slots),
                    fix=t(self.fix) if self.fix is not None else np.zeros((0, 3), np.int32),
                    pace=np.array([[d, p] for d, p in sorted(self.pace.items())], np.int64).reshape(-1, 2),
                    # the product's time at that pace: what the lost-lock guard compares against (a pace without it
                    # -- a file of an older build -- is not restored: the guard could not watch it)
                    tuned_ms=np.array([[d, self.tuned_ms[d]] for d in sorted(self.pace) if d in self.tuned_ms],
                                      np.float64).reshape(-1, 2),
                    warp=t(self.warp) if getattr(self, 'warp', None) is not None else np.zeros(0, np.int32),
                    warp_shift=int(getattr(self, 'warp_shift', 0)))
        import os
        import tempfile
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        # every rank of a data-parallel job may get here at the same time: each writes its OWN temporary
        # file (same directory, so the rename stays atomic) and the last complete one wins
        fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".tmp", dir=os.path.dirname(path) or ".")
        try:
            with os.fdopen(fd, "wb") as f:
                np.savez(f, **blob)
            os.replace(tmp, path)
        except BaseException:
            try:
                os.unlink(tmp)
            except OSError:
                pass
            raise

    @classmethod
    def load(cls, path, device, key, g=None):
        """The plan cached at ``path`` if it was built from the matrix ``key`` identifies (and with ``g`` lane
        groups, when given), else None."""
        import os
        if not os.path.exists(path):
            return None
        try:
            return cls._load(path, device, key, g)
        except Exception:            # truncated / corrupt / foreign file: a cache miss, never a crash at startup
            return None

    @classmethod
    def _load(cls, path, device, key, g):
        z = np.load(path)
        if str(z["key"]) != key:
            return None
        G = int(z["G"]) if "G" in z.files else 1
        if g is not None and G != g:
            return None
        self = cls.__new__(cls)
        self.grouped, self.pos2col, self.G, self.ranged = False, None, G, 0
        self.pad_fraction = float(z["pad_fraction"]) if "pad_fraction" in z.files else 0.0
        self.shape = tuple(int(x) for x in z["shape"])
        self.R, self.nslots, self.round_tiles = int(z["R"]), int(z["nslots"]), int(z["round_tiles"])
        tile_ptr = z["tile_ptr"]
        self.ntiles, self.nfix = int(tile_ptr.shape[0] - 1), int(z["fix"].shape[0])
        self._tile_nnz = (np.diff(tile_ptr) // G).astype(np.int64)         # steps per tile (what the pace counts)
        self._hint, self._hint_round = None, None
        # a cached pace is only as good as the box and clock it was tuned on: it comes back WITH the time it gave there,
        # so the run-time guard (_guard_before / _guard_after) is armed from the first product on; paces stored without
        # that time (files of older builds) are dropped and tuned again
        tuned = {int(d): float(ms) for d, ms in z["tuned_ms"]} if "tuned_ms" in z.files else {}
        self.pace = {int(d): int(p) for d, p in z["pace"] if int(d) in tuned or int(p) <= 0}
        self.tuned_ms, self._guard, self._tuning = {d: ms for d, ms in tuned.items() if d in self.pace}, {}, False
        to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)          # noqa: E731
        self.tile_ptr, self.colrow, self.val = to(tile_ptr), to(z["colrow"]), to(z["val"])
        self.tile_rows, self.tile_slots = to(z["tile_rows"]), to(z["tile_slots"])
        self.fix = to(z["fix"]) if self.nfix else None
        self.ws, self.device = None, device
        self.nnz = int(z["colrow"].shape[0])
        self.warp, self.warp_shift = None, 0
        if "warp" in z.files and z["warp"].shape[0]:
            self.warp, self.warp_shift = to(z["warp"]), int(z["warp_shift"])
        return self

    @classmethod
    def cached(cls, a, device, path=None, G=1):
        """Plan of ``a`` (with G lane groups per wavefront): loaded from ``path`` when that file holds such a
        plan of this very matrix, otherwise built (and written to ``path``).  Returns (plan, came_from_cache)."""
        if path is None:
            return cls(a, device, G=G), False
        # the identity of the matrix AND of the build parameters the plan depends on (tiles per launch round)
        key = "%s:r%d:Tauto:aauto:w1" % (cls.matrix_key(a), int(_ffi.lib.sgcn_tune_get(b"cs_round") or 4096))   # cached() builds with the default T / align
        hit = cls.load(path, device, key, g=G)
        if hit is not None:
            return hit, True
        plan = cls(a, device, G=G)
        plan._cache = (path, key)
        return plan, False

    def store_if_cached(self):
        """Write the plan (with the paces autotuned so far) to the path ``cached`` was given."""
        c = getattr(self, "_cache", None)
        if c is not None:
            try:
                self.save(*c)
            except OSError:           # read-only dataset directory: the cache is an optimisation only
                pass

    def struct(self, d):
        ldw = (d + 3) // 4 * 4
        need = self.nslots * ldw
        if need and (self.ws is None or self.ws.numel() < need):
            self.ws = torch.empty(need, dtype=torch.float32, device=self.device)
        rnd = self.round_tiles or (_ffi.lib.sgcn_tune_get(b"cs_round") or 4096)
        if self._hint_round != rnd:         # heaviest tile of every launch, for pacing
            nl = -(-self.ntiles // rnd)
            pad = np.zeros(nl * rnd, dtype=np.int64)
            pad[:self.ntiles] = self._tile_nnz
            self._hint = np.ascontiguousarray(pad.reshape(nl, rnd).max(axis=1))
            self._hint_round = rnd
        return _ffi.CsPlan(self.R, self.ntiles, self.tile_ptr.data_ptr(), self.colrow.data_ptr(),
                           self.val.data_ptr(), self.tile_rows.data_ptr(), self.tile_slots.data_ptr(),
                           _ptr(self.fix), self.nfix, self.nslots, _ptr(self.ws),
                           0 if self.ws is None else self.ws.numel(), rnd, self._hint.ctypes.data,
                           -1 if self.grouped else int(self.pace.get(d, 0)), int(getattr(self, 'G', 1)),
                           1 if (self.grouped or getattr(self, 'ranged', 0)) else 0, _ptr(getattr(self, 'warp', None)),
                           int(getattr(self, 'warp_shift', 0)))

    def variant(self, d):
        """The kernel variant / launch geometry sgcn_spmm_cs_f32 uses for this plan and width."""
        buf = C.create_string_buffer(256)
        plan = self.struct(d)
        check(lib.sgcn_spmm_cs_variant(C.byref(plan), int(d), buf, 256))
        return buf.value.decode()

    def autotune(self, B, d=None, candidates=None, reps=2, refine=True):
        """Pick the sweep clock for this plan and row width by timing a few candidates (the
        sustainable pace depends on the graph, d and the chip's clocks; too fast loses the
        lock-step and with it the L2 hits, too slow leaves the memory system idle)."""
        d = int(B.shape[1] if d is None else d)
        self._tuning = True
        try:
            return self._autotune(B, d, candidates, reps, refine)
        finally:
            self._tuning = False

    def _autotune(self, B, d, candidates, reps, refine):
        if candidates is None:           # ns per step of the heaviest tile (a G = 2 step is one load for two nonzeros)
            candidates = (-1, 230, 250, 270, 290, 310, 340, 380) if getattr(self, 'ranged', 0) else \
                (-1, 200, 220, 240, 260, 280, 320, 380) if getattr(self, 'G', 1) == 1 else \
                (-1, 130, 160, 190, 210, 230, 250, 280, 320)
        if self.grouped:                 # grouped plans run unpaced (see the class docstring)
            self.pace[d] = -1
            return (None, -1)
        out = torch.empty((self.shape[0], (d + 3) // 4 * 4), dtype=torch.float32, device=B.device)[:, :d]
        def timed(p, reps=reps):
            self.pace[d] = p
            spmm_cs(self, B, out=out, d=d)                       # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                spmm_cs(self, B, out=out, d=d)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        best = None
        for p in candidates:
            t = timed(p)
            if best is None or t < best[0]:
                best = (t, p)
        # a winner at the slow end of the list: the optimum may lie beyond it (sparse blocks of large graphs, whose steps
        # mostly miss the L2: S-RMAT 10 M, 360 - 400 ns per step) -- keep slowing the clock while that pays
        ext = 0
        while refine and best[1] > 0 and best[1] == max(candidates) and best[1] < 2000 and ext < 8:
            ext += 1                                  # (bounded: timing noise must not decide when this ends)
            p = max(best[1] + 1, int(best[1] * 1.15))
            candidates = tuple(candidates) + (p,)
            t = timed(p)
            if t < best[0]:
                best = (t, p)
        if refine and best[1] > 0:
            # The optimum sits right above a cliff (too fast a clock and the waves lose the lock-step for good),
            # and a two-launch burst tolerates a faster clock than a sustained run does (measured: 268 ns wins the
            # burst, 7.5 ms per fwd+bwd sustained; 274 ns: 7.2; 280: 7.4).  So: a finer look around the coarse
            # winner, then a 2 % guard band above the fastest clock that held.
            for p in sorted({int(best[1] * f) for f in (0.94, 0.96, 0.98, 1.02)} - set(candidates)):
                t = timed(p)
                if t < best[0]:
                    best = (t, p)
            # (round 4: 1 % instead of 2 -- the product costs 3.17 ms at 193 ns, 3.22 at 197, 3.39 at 189,
            # profiles/r27_headline_knobs.jsonl -- now that a lost lock-step is detected and re-tuned at run time: _guard_*)
            best = (best[0], int(best[1] * 1.01 + 0.5))
            # ... and a sustained check of the choice: a run of back-to-back products must not cost more than the
            # burst did (a lost lock-step costs 60 %, not a few); if it does, back the clock off in 4 % steps
            burst = best[0]
            for _ in range(4):
                t = timed(best[1], reps=6)
                if t <= 1.06 * burst:
                    best = (t, best[1])
                    break
                best = (t, int(best[1] * 1.04 + 0.5))
        self.pace[d] = best[1]
        self.tuned_ms[d] = float(best[0])
        self._guard.pop(d, None)
        return best

    # ---- lost-lock guard ----------------------------------------------------------------------------
    # The clock-paced sweep is tuned once per plan and width; a pace that has become too fast (another box, a clock or
    # thermal change) loses the lock-step for good and the product costs ~2x, silently.  Every GUARD_EVERY-th product
    # of a tuned plan is timed with a pair of events that are read back -- without waiting -- by a later call; two
    # samples in a row above GUARD_FACTOR x the tuned time re-run the autotune on that call's operand.
    GUARD_EVERY = 8
    GUARD_FACTOR = 1.3

    def _guard_before(self, d):
        """Called by spmm_cs ahead of the launches: returns the event pair to record around them, or None."""
        if self.grouped or self.pace.get(d, 0) <= 0 or d not in self.tuned_ms:
            return None
        g = self._guard.setdefault(d, {"calls": 0, "pending": None, "strikes": 0, "retunes": 0, "last_ms": None})
        g["calls"] += 1
        if g["pending"] is not None or g["calls"] % self.GUARD_EVERY:
            return None
        g["pending"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        return g["pending"]

    def _guard_after(self, d, B):
        """Called by spmm_cs behind the launches: reads a finished sample, re-tunes after two slow ones in a row."""
        g = self._guard.get(d)
        if g is None or g["pending"] is None or not g["pending"][1].query():
            return
        e0, e1 = g["pending"]
        g["pending"] = None
        g["last_ms"] = e0.elapsed_time(e1)
        if g["last_ms"] > self.GUARD_FACTOR * self.tuned_ms[d]:
            g["strikes"] += 1
        else:
            g["strikes"] = 0
        if g["strikes"] >= 2:
            n = g["retunes"] + 1
            self.autotune(B, d=d)                      # (resets the guard's state for d)
            self._guard[d] = {"calls": 0, "pending": None, "strikes": 0, "retunes": n, "last_ms": None}


def spmm_cs(A, B, out=None, gidx=None, rscale=None, cscale=None, beta=0.0, d=None):
    """Column-sweep variant of ``spmm`` for a ColumnSweepCSR (sgcn_spmm_cs_f32)."""
    M, K = A.shape
    bptr, ldb = _rows2d(B, "B")
    d = int(B.shape[1] if d is None else d)
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an existing `out`")
        pitch = (d + 3) // 4 * 4
        out = torch.empty((M, pitch), dtype=torch.float32, device=B.device)[:, :d]
    cptr, ldc = _rows2d(out, "out")
    plan = A.struct(d)
    if A.pos2col is not None:        # the plan's columns are sweep positions: B row = pos2col[position]
        if cscale is not None:
            raise ValueError("a grouped column-sweep plan does not take cscale (scale B instead)")
        gidx = A.pos2col if gidx is None else _dev(gidx, torch.int32, "gidx")[A.pos2col.long()]
    gp, rp, cp = _ptr(_dev(gidx, torch.int32, "gidx")), _ptr(_dev(rscale, torch.float32, "rscale")), \
        _ptr(_dev(cscale, torch.float32, "cscale"))
    ev = A._guard_before(d) if not A._tuning else None
    if ev is not None:
        ev[0].record()
    try:
        check(lib.sgcn_spmm_cs_f32(C.byref(plan), M, K, d, bptr, ldb, gp, rp, cp, cptr, ldc, float(beta), _stream()))
        if ev is not None:
            ev[1].record()
    except BaseException:
        if ev is not None:          # a sample whose second event was never recorded must not stay pending
            A._guard[d]["pending"] = None
        raise
    if not A._tuning and A._guard:
        A._guard_after(d, B)
    return out


# ---- LDS-staged column sweep for graphs with locality (sgcn_spmm_lds.hip) --------------------------
class LdsPlanHost(object):
    """The host arrays of an LDS-sweep plan (include/sgcn.h sgcn_ldsplan_*): what ``LdsSweepCSR`` uploads, and what
    the CPU tests decode (``decode`` rebuilds the planned matrix from the kernel's own operands)."""

    def __init__(self, a, labels=None, VW=2, T=0, min_reuse=2, general=False, ring_slots=0):
        a = a.tocsr()
        rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
        col = np.ascontiguousarray(a.indices, dtype=np.int32)
        val = np.ascontiguousarray(a.data, dtype=np.float32)
        M, K = int(a.shape[0]), int(a.shape[1])
        # One value per COLUMN (the transpose of a row-normalised adjacency: the backward of a mean aggregation) and not one
        # per row: the plan is built on the pattern -- a unit plan -- and the values become a scale of the operand's rows
        # (``col_fold``; ops.spmm_lds multiplies B by it first: M . B = pattern(M) . (col_fold (.) B)).
        self.col_fold = None
        if not general and val.shape[0]:
            vb = val.view(np.int32)
            first = np.zeros(M, dtype=np.int32)
            nz = np.diff(rowptr) > 0
            first[nz] = vb[rowptr[:-1][nz]]
            row_unit = bool(np.all(vb == np.repeat(first, np.diff(rowptr))))
            if not row_unit:
                cf = np.zeros(K, dtype=np.int32)
                cf[col] = vb                                   # (any nonzero of the column: all equal, or the check fails)
                if bool(np.all(cf[col] == vb)):
                    self.col_fold = np.where(np.bincount(col, minlength=K) > 0, cf, np.float32(1.0).view(np.int32)).view(np.float32)
                    val = np.ones_like(val)
        col_pos = row_group = None
        if labels is not None:
            row_labels, col_labels = labels if isinstance(labels, tuple) else (labels, labels)
            if col_labels is not None:
                col_labels = np.ascontiguousarray(col_labels, dtype=np.int32)
                if col_labels.shape[0] != K:
                    raise ValueError("one label per column")
                pos2col = np.argsort(col_labels, kind='stable').astype(np.int32)        # sweep position -> column
                col_pos = np.empty_like(pos2col)
                col_pos[pos2col] = np.arange(K, dtype=np.int32)
            if row_labels is not None:
                row_group = np.ascontiguousarray(row_labels, dtype=np.int32)
                if row_group.shape[0] != M:
                    raise ValueError("one label per row")
        h = C.c_void_p()
        check(lib.sgcn_ldsplan_create(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, K,
                                      col_pos.ctypes.data if col_pos is not None else None,
                                      row_group.ctypes.data if row_group is not None else None,
                                      int(VW), int(T), int(min_reuse), 1 if general else 0, int(ring_slots), C.byref(h)))
        try:
            sizes = np.zeros(19, dtype=np.int64)
            check(lib.sgcn_ldsplan_sizes(h, sizes.ctypes.data))
            nt, nch, nent, nfix, nslots, rnnz, staged, unit = (int(x) for x in sizes[:8])
            self.xcd_tile_ptr = [int(x) for x in sizes[8:17]]
            self.VW, self.NW, self.RW, self.S, self.U, self.unit = int(VW), 8, 192 // int(VW), int(sizes[17]), 8, unit
            self.nparts = int(sizes[18])
            R = self.NW * self.RW
            self.tile_chunk_ptr = np.empty(nt + 1, dtype=np.int32)
            self.chunk_cols = np.empty(nch * self.S, dtype=np.int32)
            self.chunk_hdr = np.empty(nch * self.NW * 32, dtype=np.int32)
            self.ent_ptr = np.empty(nch * self.NW + 1, dtype=np.int64)
            self.words = np.empty(nent + 256, dtype=np.uint32)
            self.vals = np.empty(nent + 256, dtype=np.float32)
            self.row_fold = np.empty(M, dtype=np.float32)
            self.tile_rows = np.empty(nt * R, dtype=np.int32)
            self.tile_slots = np.empty(nt * R, dtype=np.int32)
            self.fix = np.empty((nfix, 3), dtype=np.int32)
            res_rowptr = np.empty(M + 1, dtype=np.int32)
            res_col = np.empty(rnnz, dtype=np.int32)
            res_val = np.empty(rnnz, dtype=np.float32)
            check(lib.sgcn_ldsplan_export(h, self.tile_chunk_ptr.ctypes.data, self.chunk_cols.ctypes.data,
                                          self.chunk_hdr.ctypes.data, self.ent_ptr.ctypes.data, self.words.ctypes.data, self.vals.ctypes.data,
                                          self.row_fold.ctypes.data if M else None, self.tile_rows.ctypes.data,
                                          self.tile_slots.ctypes.data, self.fix.ctypes.data if nfix else None,
                                          res_rowptr.ctypes.data, res_col.ctypes.data if rnnz else None,
                                          res_val.ctypes.data if rnnz else None))
        finally:
            lib.sgcn_ldsplan_destroy(h)
        import scipy.sparse as sp
        self.shape = (M, K)
        self.ntiles, self.nchunks, self.nent, self.nfix, self.nslots, self.staged = nt, nch, nent, nfix, nslots, staged
        self.nnz = int(a.nnz)
        self.residual = sp.csr_matrix((res_val, res_col, res_rowptr), shape=(M, K))
        self.local_nnz = self.nnz - rnnz

    def planned(self):
        """(row, column, value, workspace slot) as ``decode``, with the column values folded back in."""
        r, c, v, sl = self.decode()
        return (r, c, v * self.col_fold[c], sl) if self.col_fold is not None else (r, c, v, sl)

    def decode(self):
        """(row, column, value, workspace slot) of every non-pad entry, read back from the kernel's operands: the ring
        slot of an entry's LDS address -> ``chunk_cols``, its register offset -> ``tile_rows``."""
        piece = 256 * self.VW
        part = self.S * piece
        rows, cols, vals, slots = [], [], [], []
        for t in range(self.ntiles):
            cb, ce = int(self.tile_chunk_ptr[t]), int(self.tile_chunk_ptr[t + 1])
            nc = ce - cb
            for w in range(self.NW):
                base = cb * self.NW + w * nc
                for k in range(nc):
                    e0, e1 = int(self.ent_ptr[base + k]), int(self.ent_ptr[base + k + 1])
                    hd = self.chunk_hdr[((cb + k) * self.NW + w) * 32:((cb + k) * self.NW + w + 1) * 32]
                    per = self.S // self.NW
                    assert np.array_equal(hd[:per], self.chunk_cols[(cb + k) * self.S + w * per:(cb + k) * self.S + (w + 1) * per])
                    assert hd[16] * self.U == e1 - e0 and (int(hd[17]) & 0xffffffff) | (int(hd[18]) << 32) == e0
                    if e1 == e0:
                        continue
                    assert (e1 - e0) % self.U == 0 and e1 - e0 <= (256 if self.unit else 128)
                    word = self.words[e0:e1]
                    addr = word & np.uint32(0x0003fe00)           # (the kernel's mask: LDS address bits of a 512-byte piece)
                    real = addr != self.nparts * part             # pads sit on the zero piece
                    assert self.unit or np.all(self.vals[e0:e1][~real] == 0)
                    assert np.all(addr[real] // part == k % self.nparts), "entry in the wrong part of the ring"
                    npair = int(hd[19]) * self.U                  # the first words of a unit plan's chunk carry TWO rows
                    assert 0 <= npair <= min(e1 - e0, 128) and (self.unit or npair == 0)
                    assert np.all(word[npair:] >> np.uint32(24) == 0)
                    for second in (False, True):
                        sel = real.copy()
                        if second:
                            sel[npair:] = False
                        if not sel.any():
                            continue
                        slot = (addr[sel] % part) // piece
                        lr = ((word[sel] >> np.uint32(24)) if second else (word[sel] & np.uint32(0xff))) // self.VW
                        place = (t * self.NW + w) * self.RW + lr.astype(np.int64)
                        rows.append(self.tile_rows[place])
                        slots.append(self.tile_slots[place])
                        cols.append(self.chunk_cols[(cb + k) * self.S + slot.astype(np.int64)])
                        # (unit plans: the value lives once per row, row_fold -- the words' value slots are not used)
                        vals.append(self.row_fold[self.tile_rows[place]] if self.unit else self.vals[e0:e1][sel])
        cat = lambda x, dt: np.concatenate(x) if x else np.zeros(0, dtype=dt)          # noqa: E731
        return cat(rows, np.int32), cat(cols, np.int32), cat(vals, np.float32), cat(slots, np.int32)


class LdsSweepCSR(object):
    """A static CSR re-laid for the LDS-staged column sweep (sgcn_spmm_lds.hip): for graphs whose rows share columns
    inside a tile of 768 rows (communities; ``labels`` = community per vertex, e.g. from ``reorder_labels``).  The
    nonzeros whose column a tile references fewer than ``min_reuse`` times are multiplied by the ordinary column sweep
    (``self.residual``: a ColumnSweepCSR) into the same output."""

    def __init__(self, a, device, labels=None, VW=2, T=0, min_reuse=2, residual_G=4, host=None, general=False, ring_slots=0,
                 residual_align=8192):
        h = host if host is not None else LdsPlanHost(a, labels=labels, VW=VW, T=T, min_reuse=min_reuse, general=general,
                                                      ring_slots=ring_slots)
        self.host_stats = dict(ntiles=h.ntiles, nchunks=h.nchunks, nent=h.nent, staged=h.staged, nnz=h.nnz,
                               local_nnz=h.local_nnz, pad_fraction=1.0 - h.local_nnz / max(h.nent, 1), unit=h.unit,
                               reuse=h.local_nnz / max(h.staged, 1))
        self.shape, self.nnz, self.device = h.shape, h.nnz, device
        self.VW, self.NW, self.RW, self.S, self.U, self.unit, self.nparts = h.VW, h.NW, h.RW, h.S, h.U, h.unit, h.nparts
        self.xcd_tile_ptr = list(h.xcd_tile_ptr)
        self.ntiles, self.nchunks, self.nent, self.nfix, self.nslots = h.ntiles, h.nchunks, h.nent, h.nfix, h.nslots
        to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)          # noqa: E731
        self.tile_chunk_ptr, self.chunk_cols, self.chunk_hdr = to(h.tile_chunk_ptr), to(h.chunk_cols), to(h.chunk_hdr)
        self.ent_ptr, self.words = to(h.ent_ptr), to(h.words.view(np.int32))
        self.vals = None if h.unit else to(h.vals)
        self.row_fold = to(h.row_fold) if h.unit else None
        self.tile_rows, self.tile_slots = to(h.tile_rows), to(h.tile_slots)
        self.fix = to(h.fix) if h.nfix else None
        self.ws = None
        self.col_fold = to(h.col_fold) if h.col_fold is not None else None       # scale of the operand's rows (see LdsPlanHost)
        self._scaled = None
        self.residual = None
        if h.residual.nnz:
            # four lane groups per wave: the residual's rows are all resident in ONE round of tiles, and a sparse residual is
            # bound by the bytes each XCD pulls over the fabric, not by the step's instructions (S-Reddit-SBM p_in 0.8, 5.0 M
            # nonzeros, sustained: 1.26 ms against 1.50 with two groups; bins aligned to 8,192 columns: profiles/r31_*)
            if residual_G == 4 and h.residual.nnz > 40 * max(h.shape[0], 1):
                residual_G, residual_align = 2, 2048       # a DENSE residual (most of the graph) is the full-graph case: two groups
            # (long residual rows are split at 256 nonzeros, not at the plan's default 8 x the mean row: fewer fix-up slots,
            # the same heaviest bin -- 1.275 -> 1.242 ms, profiles/lds_residual_probe.py 0.8 T)
            self.residual = ColumnSweepCSR(h.residual, device, G=residual_G, align=residual_align,
                                           T=256 if residual_G == 4 else 0) if residual_G else \
                DeviceCSR.from_scipy(h.residual, device)

    def struct(self, d):
        ldw = (d + 3) // 4 * 4
        need = self.nslots * ldw
        if need and (self.ws is None or self.ws.numel() < need):
            self.ws = torch.empty(need, dtype=torch.float32, device=self.device)
        return _ffi.LdsPlan(self.VW, self.NW, self.RW, self.S, self.U, self.nparts, self.unit, (C.c_int32 * 9)(*self.xcd_tile_ptr),
                            self.ntiles, self.nchunks, self.nent,
                            self.tile_chunk_ptr.data_ptr(), self.chunk_cols.data_ptr(), self.chunk_hdr.data_ptr(),
                            self.ent_ptr.data_ptr(), self.words.data_ptr(), _ptr(self.vals), _ptr(self.row_fold),
                            self.tile_rows.data_ptr(), self.tile_slots.data_ptr(),
                            _ptr(self.fix), self.nfix, self.nslots, _ptr(self.ws),
                            0 if self.ws is None else self.ws.numel())

    @classmethod
    def for_graph(cls, a, device, min_local=0.6, min_gain=3.0, min_reuse=3):
        """The LDS-sweep plan of a square adjacency IF the graph has the locality that pays for it, else None: communities
        from label propagation on the graph itself (``reorder_labels``), and a plan whose staged pieces serve at least
        ``min_gain`` nonzeros each while leaving at most 1 - ``min_local`` of the nonzeros to the residual sweep (measured
        on S-Reddit-SBM, profiles/r31_*: with 74 % of the nonzeros planned the two-kernel product takes 2.60 ms against 3.20
        for the plain column sweep, with 91 % 2.09 against 3.15; with 40 % the two draw level, 3.21; with 11-14 % it loses,
        3.6-3.7 against 3.2).  A graph without structure (S-Reddit: one label) costs the 0.3 s of the propagation."""
        a = a.tocsr()
        if a.shape[0] != a.shape[1] or a.nnz == 0:
            return None
        labels, ncomm = reorder_labels(a)
        if ncomm < 4:
            return None
        host = LdsPlanHost(a, labels=labels, min_reuse=min_reuse)
        if host.local_nnz < min_local * a.nnz or host.local_nnz < min_gain * host.staged:
            return None
        return cls(a, device, host=cls.auto_host(a, labels, host))

    # A residual costs ten launches and a sweep of B through every XCD however few nonzeros it has; below this share of the
    # nonzeros it is cheaper to stage EVERY column through the ring, single-use pieces included (no residual, no second
    # kernel).  Measured on S-Reddit-SBM, min_reuse 1 against 3 (profiles/r36_*): p_in 0.95 (9 % residual) 1.74 against
    # 2.09 ms, 0.9 (13 %) 2.09 / 2.17, 0.85 (19 %) 2.40 / 2.31, 0.8 (26 %) 2.69 / 2.44.
    ALL_STAGED_BELOW = 0.16

    @classmethod
    def auto_host(cls, a, labels, host=None, min_reuse=3):
        """The host plan for ``a``: ``min_reuse`` 3, or 1 (everything through the ring) when that leaves a small residual."""
        if host is None:
            host = LdsPlanHost(a, labels=labels, min_reuse=min_reuse)
        if 0 < host.residual.nnz < cls.ALL_STAGED_BELOW * max(a.nnz, 1):
            host = LdsPlanHost(a, labels=labels, min_reuse=1)
        return host

    def variant(self, d):
        """What spmm_lds dispatches for this plan and width, as text (bench.py: roofline.kernel)."""
        nslab = (int(d) + 127) // 128
        txt = ("sgcn::lds_spmm_kernel<%d, %d, %s> x 1 launch (%d tiles of 768 rows x %d passes of 128 columns = %d workgroups; "
               "%d chunks, %.1f nonzeros per staged piece, %.1f %% of the nonzeros)"
               % (self.S, self.nparts, "true" if self.unit else "false", self.ntiles, nslab, self.ntiles * nslab, self.nchunks,
                  self.host_stats["reuse"], 100.0 * self.host_stats["local_nnz"] / max(self.nnz, 1)))
        if isinstance(self.residual, ColumnSweepCSR):
            txt += " + residual: " + self.residual.variant(d)
        return txt

    def autotune(self, B, d=None):
        """The residual's sweep clock (the LDS kernel itself is not clock-paced).  (Running the two kernels side by side
        on split compute units -- CU-masked streams -- was measured and lost: profiles/r24_lds_side_by_side_probe.jsonl.)"""
        if isinstance(self.residual, ColumnSweepCSR):
            return self.residual.autotune(B, d=d)
        return None


def spmm_lds(A, B, out=None, rscale=None, beta=0.0, d=None, local_only=False):
    """C = rscale (.) (A B) + beta C for an LdsSweepCSR: the planned nonzeros through the LDS ring
    (sgcn_spmm_lds_f32), then the residual on top (column sweep with beta = 1)."""
    M, K = A.shape
    bptr, ldb = _rows2d(B, "B")
    d = int(B.shape[1] if d is None else d)
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an existing `out`")
        pitch = (d + 3) // 4 * 4
        out = torch.empty((M, pitch), dtype=torch.float32, device=B.device)[:, :d]
    cptr, ldc = _rows2d(out, "out")
    if A.col_fold is not None:               # one value per column: B's rows take it (one pass over B), the plan is a unit plan
        pitch = (d + 31) // 32 * 32               # rows on 128-byte lines: a piece of a row never straddles one more line than it must
        if A._scaled is None or A._scaled.shape[0] < K or A._scaled.shape[1] != pitch:
            A._scaled = torch.empty((K, pitch), dtype=torch.float32, device=B.device)
        check(lib.sgcn_scale_rows_f32(bptr, ldb, A.col_fold.data_ptr(), K, d, A._scaled.data_ptr(), pitch, _stream()))
        B = A._scaled[:K, :d]
        bptr, ldb = _rows2d(B, "B")
    plan = A.struct(d)
    check(lib.sgcn_spmm_lds_f32(C.byref(plan), M, K, d, bptr, ldb, _ptr(_dev(rscale, torch.float32, "rscale")),
                                cptr, ldc, float(beta), _stream()))
    if A.residual is not None and not local_only:
        if isinstance(A.residual, ColumnSweepCSR):
            spmm_cs(A.residual, B, out=out, rscale=rscale, beta=1.0, d=d)
        else:
            spmm(A.residual, B, out=out, rscale=rscale, beta=1.0, d=d)
    return out


# ---- fp32 MFMA GEMM / fused dense layer (sgcn_gemm.hip) ------------------------------------------
# Every dense product of the model runs on this library's own kernel, whatever its size (round 3: the size-keyed
# rocBLAS path above 512 M multiply-adds -- Exact-mode and large evaluation batches -- is gone: one code path; the
# kernel is a plain tiled GEMM with split-K, so a 233 k-row operand is just more workgroups).


_GEMM_WS = {}


def _gemm_ws(floats, device):
    """Split-K scratch, one growing buffer per device (all launches share one stream, so the
    next GEMM cannot overwrite partials the previous reduce has not consumed)."""
    w = _GEMM_WS.get(device)
    if w is None or w.numel() < floats:
        w = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=device)
        _GEMM_WS[device] = w
    return w


def dense_bwd(dy, y, ctx, scale, relu, x, W, dW, doffset=None, dscale=None, need_dx=True, drop=None):
    """Backward of ``dense_fwd`` in one call (sgcn_dense_bwd_f32): accumulates dW (and the LayerNorm
    parameter gradients), returns dx or None.  ``x`` is the UNdropped layer input; ``drop`` the
    dropout site that was applied to it in the forward."""
    gidx = None
    if isinstance(x, GatheredRows):
        x, gidx = x.src, x.idx
    n, N, K = int(dy.shape[0]), int(dy.shape[1]), int(x.shape[1])
    norm = ctx is not None
    gp, ldg = _rows2d(dy, "dy")
    xp, ldx = _rows2d(x, "x")
    wp, ldw = _rows2d(W, "W")
    dwp, lddw = _rows2d(dW, "dW")
    pre = norm or bool(relu)
    yp, ldy = _rows2d(y, "y") if pre else (None, 0)
    g_tmp = torch.empty((n, N), dtype=torch.float32, device=dy.device) if pre else None
    dx = torch.empty((n, K), dtype=torch.float32, device=dy.device) if need_dx else None
    key = (n, N, K, norm)
    need = _DENSE_BWD_WS.get(key)
    if need is None:
        need = ((int(lib.sgcn_ln_act_bwd_ws_floats(n, N)) + 3) // 4 * 4 if norm else 0) + \
            max(int(lib.sgcn_gemm_ws_floats(K, N, n)), int(lib.sgcn_gemm_ws_floats(n, K, N)))
        if len(_DENSE_BWD_WS) < 4096:
            _DENSE_BWD_WS[key] = need
    ws = _gemm_ws(need, dy.device) if need else None
    dr = C.byref(drop.struct(K)) if drop is not None else None
    check(lib.sgcn_dense_bwd_f32(n, N, K, gp, ldg, yp, ldy, _ptr(ctx[0]) if norm else None,
                                 _ptr(ctx[1]) if norm else None, _ptr(scale) if norm else None,
                                 int(bool(relu)), xp, ldx, wp, ldw, dwp, lddw, _ptr(doffset), _ptr(dscale),
                                 _ptr(dx), K, dr, _ptr(g_tmp), _ptr(ws), _ptr(gidx), _stream()))
    return dx


_DENSE_BWD_WS = {}


# ---- counter-based dropout (include/sgcn.h sgcn_dropout_t) ---------------------------------------
def _fmix32(h):
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def dropout_key(seed, layer_index, step):
    """The 32-bit key of one dropout site at one step: a hash of (seed, layer index, step)."""
    return _fmix32(_fmix32(int(seed) * 0x9E3779B1 + int(layer_index) * 0x85EBCA77 + 0x27D4EB2F)
                   + int(step) * 0xC2B2AE3D)


class Drop(object):
    """One dropout site: (keep, key) + the activation geometry the mask is defined on."""
    __slots__ = ("keep", "key")

    def __init__(self, keep, key):
        self.keep, self.key = float(keep), int(key)

    def struct(self, width, rows=-1):
        return _ffi.Dropout(self.key, self.keep, int(rows), int(width))


def dropout(x, drop, out=None):
    """x * mask / keep with the hash mask (sgcn_dropout_f32): the unfused form, and -- the mask being
    a function of the element index only -- also its own backward."""
    if x.dim() == 1:
        x2 = x.view(1, -1)
        return dropout(x2, drop, None if out is None else out.view(1, -1)).view(-1)
    xp, ldx = _rows2d(x, "x")
    n, d = int(x.shape[0]), int(x.shape[1])
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    op, ldo = _rows2d(out, "out")
    st = drop.struct(d)
    check(lib.sgcn_dropout_f32(xp, ldx, n, d, C.byref(st), op, ldo, _stream()))
    return out


def gemm(A, B, out=None, trans_a=False, trans_b=False, accumulate=False, drop_a=None, drop_c=None):
    """out = op(A) @ op(B) (+ out)   (sgcn_gemm_f32; exact fp32 on the matrix cores).
    drop_a: the stored A is a dropout input (masked while loaded); drop_c: mask the output."""
    ap, lda = _rows2d(A, "A")
    bp, ldb = _rows2d(B, "B")
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if K != K2:
        raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, K2))
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an existing `out`")
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    cp, ldc = _rows2d(out, "out")
    need = int(lib.sgcn_gemm_ws_floats(int(M), int(N), int(K)))     # > 0: split-K pays (small M x N, long K)
    ws = _gemm_ws(need, A.device) if need else None
    da = C.byref(drop_a.struct(A.shape[1])) if drop_a is not None else None
    dc = C.byref(drop_c.struct(N)) if drop_c is not None else None
    check(lib.sgcn_gemm_f32(int(trans_a), int(trans_b), int(M), int(N), int(K), ap, lda, bp, ldb, cp, ldc,
                            int(accumulate), _ptr(ws), da, dc, _stream()))
    return out


class GatheredRows(object):
    """``src[idx]`` that has not been gathered (tf.gather / history.dense_slice of the minibatch's
    input rows, gcn/vrgcn.py:43-45): the first dense layer's GEMMs read the rows through the index
    (dense_fwd / dense_bwd ``gidx``), so the n x F copy is never made.  Any other consumer calls
    ``materialize()``."""

    def __init__(self, src, idx):
        self.src, self.idx, self._m = src, idx, None

    @property
    def shape(self):
        return (int(self.idx.shape[0]), int(self.src.shape[1]))

    def materialize(self):
        if self._m is None:
            self._m = gather_rows(self.src, self.idx)
        return self._m


def dense_fwd(x, W, offset, scale, relu, eps=1e-9, x2=None, drop=None):
    """y = act(LN(x @ W) * scale + offset) in ONE launch (sgcn_dense_fwd_f32).
    Returns (y, ctx) like ln_act_fwd.  x2: a second operand stacked below x ([x ; x2] @ W without
    the concatenation); drop: dropout on the rows of x (never on x2), applied while the operand is
    loaded.  Needs N <= 128 when LayerNorm / ReLU is requested."""
    gidx = gidx2 = None
    if isinstance(x2, GatheredRows):
        x2, gidx2 = x2.src, x2.idx
    if isinstance(x, GatheredRows):
        x, gidx = x.src, x.idx
    n1 = int(x.shape[0] if gidx is None else gidx.shape[0])
    K, N = int(x.shape[1]), int(W.shape[1])
    M = n1 + (0 if x2 is None else int(x2.shape[0] if gidx2 is None else gidx2.shape[0]))
    xp, ldx = _rows2d(x, "x")
    x2p, ldx2 = _rows2d(x2, "x2") if x2 is not None else (None, 0)
    wp, ldw = _rows2d(W, "W")
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    norm = offset is not None
    xhat = torch.empty((M, N), dtype=torch.float32, device=x.device) if norm else None
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device) if norm else None
    dr = C.byref(drop.struct(K, rows=n1)) if drop is not None else None
    need = int(lib.sgcn_gemm_ws_floats(M, N, K)) if N <= 128 else 0        # few tiles, long K: split-K
    ws = _gemm_ws(need, x.device) if need else None
    check(lib.sgcn_dense_fwd_f32(M, N, K, xp, ldx, x2p, ldx2, n1, wp, ldw, _ptr(offset), _ptr(scale),
                                 float(eps), int(bool(relu)), y.data_ptr(), N, _ptr(xhat), _ptr(rstd), dr,
                                 _ptr(ws), _ptr(gidx), _ptr(gidx2), _stream()))
    return y, ((xhat, rstd) if norm else None)


# ---- deterministic dropout: element-/row-wise pieces (sgcn_det.hip; gcn/layers.py:141-202, 236-248, 320-349, 425-428) ----
def _flat(x, name):
    """(pointer, element count) of a contiguous fp32 device array (tensor or DevArray)."""
    if isinstance(x, torch.Tensor):
        _dev(x, torch.float32, name)
        if not x.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
    return x.data_ptr(), int(x.numel())


def _like(x):
    return torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)


def det_pre(mu, var, keep):
    """Variance of dropout(x) / keep for x ~ (mu, var): var / keep + (1 / keep - 1) mu^2 (var None: plain input)."""
    mp, n = _flat(mu, "mu")
    out = _like(mu)
    check(lib.sgcn_det_pre_f32(mp, _flat(var, "var")[0] if var is not None else None, n, float(keep), out.data_ptr(), _stream()))
    return out


def det_pre_bwd(mu, g, keep, d_mu, want_var):
    """d_mu += g 2 (1 / keep - 1) mu; returns d_var = g / keep (or None)."""
    mp, n = _flat(mu, "mu")
    d_var = _like(mu) if want_var else None
    check(lib.sgcn_det_pre_bwd_f32(mp, _flat(g, "g")[0], n, float(keep), _flat(d_mu, "d_mu")[0], _ptr(d_var), _stream()))
    return d_var


def square(x, c=1.0):
    """c * x^2 element-wise (tf.square of a weight matrix / of an adjacency's values)."""
    xp, n = _flat(x, "x")
    out = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
    check(lib.sgcn_square_f32(xp, n, float(c), out.data_ptr(), _stream()))
    return out


def addmul(acc, a, b, c=1.0):
    """acc += c * a * b element-wise."""
    ap, n = _flat(a, "a")
    check(lib.sgcn_addmul_f32(_flat(acc, "acc")[0], ap, _flat(b, "b")[0], n, float(c), _stream()))
    return acc


def det_lnvar_fwd(var1, rstd, scale, eps):
    n, d = int(var1.shape[0]), int(var1.shape[1])
    out = _like(var1)
    check(lib.sgcn_det_lnvar_fwd_f32(_flat(var1, "var1")[0], _ptr(rstd), _ptr(scale), n, d, float(eps), out.data_ptr(), _stream()))
    return out


def det_lnvar_bwd(g, var1, xhat, rstd, scale, eps, d_mu1, dscale):
    """Returns d_var1; adds the variance stream's share to d_mu1 and dscale."""
    n, d = int(var1.shape[0]), int(var1.shape[1])
    d_var1, tmp = _like(var1), _like(var1)
    check(lib.sgcn_det_lnvar_bwd_f32(_flat(g, "g")[0], _flat(var1, "var1")[0], _ptr(xhat), _ptr(rstd), _ptr(scale), n, d, float(eps),
                                     d_var1.data_ptr(), _flat(d_mu1, "d_mu1")[0], _ptr(dscale), tmp.data_ptr(), _stream()))
    return d_var1


def det_relu_fwd(mu, var):
    mp, n = _flat(mu, "mu")
    mo, vo = _like(mu), _like(mu)
    check(lib.sgcn_det_relu_fwd_f32(mp, _flat(var, "var")[0], n, mo.data_ptr(), vo.data_ptr(), _stream()))
    return mo, vo


def det_relu_bwd(mu, var, g_mu, g_var):
    mp, n = _flat(mu, "mu")
    d_mu, d_var = _like(mu), _like(mu)
    check(lib.sgcn_det_relu_bwd_f32(mp, _flat(var, "var")[0], _flat(g_mu, "g_mu")[0], _flat(g_var, "g_var")[0], n,
                                    d_mu.data_ptr(), d_var.data_ptr(), _stream()))
    return d_mu, d_var


def gauss_sample(mu, var, key):
    mp, n = _flat(mu, "mu")
    x = _like(mu)
    check(lib.sgcn_gauss_sample_f32(mp, _flat(var, "var")[0], n, int(key) & 0xFFFFFFFF, x.data_ptr(), _stream()))
    return x


def gauss_sample_bwd(var, g, key):
    vp, n = _flat(var, "var")
    d_var = _like(var)
    check(lib.sgcn_gauss_sample_bwd_f32(vp, _flat(g, "g")[0], n, int(key) & 0xFFFFFFFF, d_var.data_ptr(), _stream()))
    return d_var


def det_agg_prep(mu, var, Hm, Hv, ifield):
    """(delta_mu, ds2, msig2, ds, sbar) of the control-variate aggregator on (mu, var)."""
    n0, d = int(mu.shape[0]), int(mu.shape[1])
    hp, ldh = _rows2d(Hm, "Hm")
    vp, ldv = _rows2d(Hv, "Hv")
    if ldh != ldv:
        raise ValueError("the two histories must share a pitch")
    outs = [_like(mu) for _ in range(5)]
    check(lib.sgcn_det_agg_prep_f32(_flat(mu, "mu")[0], _flat(var, "var")[0], hp, vp, ldh, _ptr(_dev(ifield, torch.int32, "ifield")),
                                    n0, d, *[o.data_ptr() for o in outs], _stream()))
    return outs


def det_agg_prep_bwd(var, ds, sbar, g_ds2, g_msig2, add=None, add_rows=0):
    n0, d = int(var.shape[0]), int(var.shape[1])
    d_var = _like(var)
    ap, ldadd = _rows2d(add, "add") if add is not None else (None, 0)
    check(lib.sgcn_det_agg_prep_bwd_f32(_flat(var, "var")[0], _ptr(ds), _ptr(sbar), _flat(g_ds2, "g_ds2")[0], _flat(g_msig2, "g_msig2")[0],
                                        n0, d, ap, ldadd, int(add_rows), d_var.data_ptr(), _stream()))
    return d_var


def relu_eps(raw, eps, out=None):
    """relu(raw) + eps (out may be a column block of a wider array)."""
    rp, ldr = _rows2d(raw, "raw")
    n, d = int(raw.shape[0]), int(raw.shape[1])
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=raw.device)
    op, ldo = _rows2d(out, "out")
    check(lib.sgcn_relu_eps_f32(rp, ldr, n, d, float(eps), op, ldo, _stream()))
    return out


def gate(raw, g):
    """g where raw > 0, else 0 (dense result; g may be a column block)."""
    rp, ldr = _rows2d(raw, "raw")
    gp, ldg = _rows2d(g, "g")
    n, d = int(raw.shape[0]), int(raw.shape[1])
    out = torch.empty((n, d), dtype=torch.float32, device=raw.device)
    check(lib.sgcn_gate_f32(rp, ldr, gp, ldg, n, d, out.data_ptr(), _stream()))
    return out
