"""oracle/oracle_np.py -- TEST INFRASTRUCTURE ONLY (the parity checker; never the product).

NumPy / plain-C CPU restatement of the sparse ops on the hot path of thu-ml/stochastic_gcn.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED at the TensorFlow boundary: the reference executes these ops as TF-1 kernels
(`tensorflow-gpu`, un-pinned, /root/reference/setup.py:12-16) and holds no test or golden
vector for them.  Each function restates the op definition at the cited
reference call site; tests/test_oracle.py cross-checks the C restatement against
scipy.sparse (what the reference itself uses for the PP product, gcn/utils.py:321-322) and
against float64 NumPy.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_CPATH = os.path.join(_HERE, "liboracle_c.so")
_clib = None


def clib():
    """The plain-C restatement (oracle/oracle_c.c), built by oracle/Makefile."""
    global _clib
    if _clib is None:
        l = C.CDLL(_CPATH)
        P = C.c_void_p
        l.oracle_spmm_csr_f32.argtypes = [P, P, P, C.c_int32, C.c_int32, P, C.c_int64, P,
                                          C.c_int64, C.c_float]
        l.oracle_spmm_csr_gather_f32.argtypes = [P, P, P, C.c_int32, C.c_int32, P, C.c_int64, P,
                                                 P, C.c_int64]
        l.oracle_gather_rows_f32.argtypes = [P, C.c_int64, P, C.c_int32, C.c_int32, P, C.c_int64]
        l.oracle_scatter_rows_f32.argtypes = [P, C.c_int64, P, C.c_int32, C.c_int32, P, C.c_int64]
        _clib = l
    return _clib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def coo_to_csr(coo_tuple):
    """(idx[ne,2], w[ne], shape) feed-dict triple (gcn/_scheduler.pyx:84-91) -> scipy CSR that
    keeps the stored order of the nonzeros inside each row (no sorting, no duplicate merging),
    which is the accumulation order of tf.sparse_tensor_dense_matmul on row-grouped input."""
    idx, w, shape = coo_tuple
    idx = np.asarray(idx).reshape(-1, 2)
    rows = idx[:, 0]
    assert np.all(np.diff(rows) >= 0), "sampler emits rows in non-decreasing order"
    rowptr = np.zeros(int(shape[0]) + 1, dtype=np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    m = sp.csr_matrix((int(shape[0]), int(shape[1])), dtype=np.float32)
    m.indptr, m.indices, m.data = rowptr, _i32(idx[:, 1]), _f32(w)
    return m


def spmm(rowptr, col, val, B, gidx=None, rscale=None, cscale=None, C_in=None, beta=0.0):
    """dot(x, y, sparse=True) = tf.sparse_tensor_dense_matmul  (gcn/layers.py:31-37), with the
    optional fusions of the device kernel restated separately: B rows via gidx (tf.gather,
    gcn/layers.py:304-305), per-column / per-row scales, beta * C."""
    rowptr, col, val = _i32(rowptr), _i32(col), _f32(val)
    B = _f32(B)
    M, d = rowptr.shape[0] - 1, B.shape[1]
    if cscale is not None:
        val = (val * _f32(cscale)[col]).astype(np.float32)
    if gidx is not None:
        col = _i32(_i32(gidx)[col])
    out = np.zeros((M, d), dtype=np.float32)
    clib().oracle_spmm_csr_f32(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, d,
                               B.ctypes.data, B.shape[1], out.ctypes.data, d, 0.0)
    if rscale is not None:
        out = (out * _f32(rscale)[:, None]).astype(np.float32)
    if beta != 0.0:
        out = (out + np.float32(beta) * _f32(C_in)).astype(np.float32)
    return out


def spmm_f64(rowptr, col, val, B):
    """float64 scipy product: the 'true' value both fp32 implementations approximate."""
    M = len(rowptr) - 1
    a = sp.csr_matrix((np.asarray(val, np.float64), np.asarray(col), np.asarray(rowptr)),
                      shape=(M, B.shape[0]))
    return a.dot(np.asarray(B, np.float64))


def gather_rows(a, r):
    """history.dense_slice (gcn/_history.pyx:53-62) / tf.gather (gcn/layers.py:304-305)."""
    return _f32(a)[_i32(r)]


def scatter_rows(H, r, src):
    """tf.scatter_update(H, ifield, new) (gcn/models.py:160-166); r unique. In place.
    Negative ids are padding and skipped (include/sgcn.h sgcn_scatter_rows_f32)."""
    r = _i32(r)
    keep = r >= 0
    H[r[keep]] = _f32(src)[keep]
    return H


def csr_slice(a, r):
    """history.slice (gcn/_history.pyx:25-51): CSR row slice -> (indices[nnz,2], data, shape)
    or an empty csr_matrix when nnz == 0 (the reference's type inconsistency, :34-35)."""
    r = _i32(r)
    sub = a[r]
    if sub.nnz == 0:
        return sp.csr_matrix((len(r), a.shape[1]), dtype=a.dtype)
    coo_rows = np.repeat(np.arange(len(r), dtype=np.int32), np.diff(sub.indptr))
    idx = np.stack([coo_rows, sub.indices.astype(np.int32)], axis=1)
    return idx, sub.data.astype(np.float32), np.array([len(r), a.shape[1]], dtype=np.int32)


def vr_aggregate(adj, fadj, h, mu, Hbar, ifield, ffield, scale, cvd, concat_self):
    """VRAggregator._call (gcn/layers.py:298-319 cvd; :350-362 plain CV).

    adj / fadj are scipy CSR in stored order (coo_to_csr).  Returns (out_h, out_mu)
    (out_mu None for plain CV) and new_history."""
    n1 = adj.shape[0]
    Hbar = _f32(Hbar)
    if cvd:
        h, mu = _f32(h), _f32(mu)
        mu_small = Hbar[_i32(ifield)]                      # tf.gather(history, ifield)  :304
        mu_large = Hbar[_i32(ffield)]                      # tf.gather(history, ffield)  :305
        z = (h - mu).astype(np.float32)                    # :306
        delta_mu = (mu - mu_small).astype(np.float32)      # :307
        mu_mean = spmm(fadj.indptr, fadj.indices, fadj.data, mu_large)            # :308
        mu_nbr = (spmm(adj.indptr, adj.indices, adj.data, delta_mu) + mu_mean).astype(np.float32)
        h_nbr = (spmm(adj.indptr, adj.indices, adj.data, z) *
                 _f32(scale)[:, None] + mu_nbr).astype(np.float32)                # :311
        new_history = [mu]
        if concat_self:
            return (np.concatenate([h[:n1], h_nbr], axis=1),
                    np.concatenate([mu[:n1], mu_nbr], axis=1), new_history)
        return h_nbr, mu_nbr, new_history
    x = _f32(h)
    cur = spmm(adj.indptr, adj.indices, adj.data, x)                               # :353
    his = spmm(adj.indptr, adj.indices, adj.data, Hbar[_i32(ifield)])              # :354
    mean = spmm(fadj.indptr, fadj.indices, fadj.data, Hbar[_i32(ffield)])          # :355
    a_nbr = ((cur - his).astype(np.float32) + mean).astype(np.float32)             # :356
    new_history = [x]
    if concat_self:
        return np.concatenate([x[:n1], a_nbr], axis=1), None, new_history
    return a_nbr, None, new_history


def plain_aggregate(adj, x, concat_self):
    """PlainAggregator._call (gcn/layers.py:249-257)."""
    x = _f32(x)
    a_nbr = spmm(adj.indptr, adj.indices, adj.data, x)
    if concat_self:
        return np.concatenate([x[:adj.shape[0]], a_nbr], axis=1)
    return a_nbr


def rel_err(x, ref):
    """max |x - ref| / max(|ref|, tiny): the parity metric of SURVEY.md §8d."""
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(x - ref)) / max(np.max(np.abs(ref)), 1e-30))
