"""oracle/ref_binding.py -- TEST INFRASTRUCTURE ONLY.

ctypes access to ``oracle/_ref/libsgcn_ref.so`` = the REAL reference C++
(gcn/scheduler.cpp, gcn/mult.cpp, gcn/history.cpp) compiled from /root/reference by
``oracle/Makefile`` behind our forwarding shim ``oracle/ref_shim.cpp``.  Used to
  * generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py),
  * pin the Python restatement in oracle/sampler.py and the product sampler,
  * time the reference sampler as ``cpu_baseline`` kind "reference".

``RefPyScheduler.batch`` restates the array packing of the Cython wrapper
gcn/_scheduler.pyx:55-127 (which is not compiled here) on top of the real C++ class.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libsgcn_ref.so")

_I = C.POINTER(C.c_int)
_F = C.POINTER(C.c_float)


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(_PATH)
        l.ref_sched_create.restype = C.c_void_p
        l.ref_sched_create.argtypes = [_F, _I, _I, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        l.ref_sched_destroy.argtypes = [C.c_void_p]
        l.ref_sched_seed.argtypes = [C.c_void_p, C.c_int]
        l.ref_sched_start_batch.argtypes = [C.c_void_p, C.c_int, _I]
        l.ref_sched_expand.argtypes = [C.c_void_p, C.c_int]
        for n in ("ref_sched_isize", "ref_sched_fsize"):
            getattr(l, n).restype = C.c_int
            getattr(l, n).argtypes = [C.c_void_p, C.c_int]
        l.ref_sched_icopy.argtypes = [C.c_void_p, C.c_int, _I]
        l.ref_sched_fcopy.argtypes = [C.c_void_p, C.c_int, _F]
        l.ref_mult_create.restype = C.c_void_p
        l.ref_mult_create.argtypes = [_F, C.c_int]
        l.ref_mult_destroy.argtypes = [C.c_void_p]
        l.ref_mult_bit_size.restype = C.c_int
        l.ref_mult_bit_size.argtypes = [C.c_void_p]
        l.ref_mult_bit_copy.argtypes = [C.c_void_p, _F]
        l.ref_mult_query_u.restype = C.c_int
        l.ref_mult_query_u.argtypes = [C.c_void_p, C.c_float]
        l.ref_mult_query.restype = C.c_int
        l.ref_mult_query.argtypes = [C.c_void_p]
        l.ref_c_indptr.argtypes = [C.c_int, _I, _I, _I]
        l.ref_c_slice.argtypes = [C.c_int, _I, _F, _I, _I, _F, _I, _I]
        l.ref_c_dense_slice.argtypes = [C.c_int, C.c_int, _I, _F, _F]
        _lib = l
    return _lib


def _ip(a):
    return a.ctypes.data_as(_I)


def _fp(a):
    return a.ctypes.data_as(_F)


I_FIELD, I_FFIELD, I_EDG_S, I_EDG_T, I_FEDG_S, I_FEDG_T, I_ADJ_I = range(7)
F_SCALES, F_EDG_W, F_MEDG_W, F_FEDG_W, F_ADJ_W = range(5)


class RefScheduler:
    """The reference ``Scheduler`` (gcn/scheduler.h:6-28)."""

    def __init__(self, adj, num_data, L, cv, importance):
        self._w = np.ascontiguousarray(adj.data, dtype=np.float32)
        self._i = np.ascontiguousarray(adj.indices, dtype=np.int32)
        self._p = np.ascontiguousarray(adj.indptr, dtype=np.int32)
        self.cv = bool(cv)
        self._h = lib().ref_sched_create(_fp(self._w), _ip(self._i), _ip(self._p), int(num_data),
                                         int(self._w.shape[0]), int(L), int(cv), int(importance))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_sched_destroy(self._h)
            self._h = None

    def seed(self, s):
        lib().ref_sched_seed(self._h, int(s))

    def start_batch(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lib().ref_sched_start_batch(self._h, int(ids.shape[0]), _ip(ids))

    def expand(self, degree):
        lib().ref_sched_expand(self._h, int(degree))

    def ivec(self, which):
        n = lib().ref_sched_isize(self._h, which)
        out = np.zeros(n, dtype=np.int32)
        lib().ref_sched_icopy(self._h, which, _ip(out))
        return out

    def fvec(self, which):
        n = lib().ref_sched_fsize(self._h, which)
        out = np.zeros(n, dtype=np.float32)
        lib().ref_sched_fcopy(self._h, which, _fp(out))
        return out


class RefPyScheduler:
    """gcn/_scheduler.pyx:28-148 on top of the real C++ class (Cython not needed)."""

    def __init__(self, adj, labels, L, degrees, placeholders, seed, data=None, cv=False,
                 importance=False):
        self.c_sch = RefScheduler(adj, labels.shape[0], L, cv, importance)
        self.c_sch.seed(seed)
        self.labels, self.data, self.degrees, self.L = labels, data, degrees, L
        self.placeholders = placeholders
        self.start = 0

    def shuffle(self):
        np.random.shuffle(self.data)
        self.start = 0

    def batch(self, data):
        s = self.c_sch
        fields, ffields, adjs, madjs, fadjs, scales = [np.asarray(data)], [], [], [], [], []
        s.start_batch(data)
        for l in range(self.L):
            s.expand(self.degrees[self.L - l - 1])
            fields.append(s.ivec(I_FIELD))
            scales.append(s.fvec(F_SCALES))
            edg_i = np.stack([s.ivec(I_EDG_S), s.ivec(I_EDG_T)], axis=1).astype(np.int32)
            shape = (fields[-2].shape[0], fields[-1].shape[0])
            adjs.append((edg_i.reshape(-1, 2), s.fvec(F_EDG_W), shape))
            if s.cv:
                ffields.append(s.ivec(I_FFIELD))
                fedg_i = np.stack([s.ivec(I_FEDG_S), s.ivec(I_FEDG_T)], axis=1).astype(np.int32)
                fshape = (fields[-2].shape[0], ffields[-1].shape[0])
                madjs.append((edg_i.reshape(-1, 2).copy(), s.fvec(F_MEDG_W), np.copy(shape)))
                fadjs.append((fedg_i.reshape(-1, 2), s.fvec(F_FEDG_W), fshape))
        for lst in (fields, ffields, adjs, madjs, fadjs, scales):
            lst.reverse()
        return self.get_feed_dict(fields, ffields, adjs, madjs, fadjs, scales)

    def minibatch(self, batch_size):
        if self.start == self.data.shape[0]:
            return None
        end = min(self.data.shape[0], self.start + batch_size)
        batch = self.data[self.start:end]
        self.start = end
        return self.batch(batch)

    def get_feed_dict(self, fields, ffields, adjs, madjs, fadjs, scales):
        ph = self.placeholders
        fd = {ph['adj'][i]: adjs[i] for i in range(self.L)}
        fd.update({ph['scales'][i]: scales[i] for i in range(len(scales))})
        if self.c_sch.cv:
            fd.update({ph['madj'][i]: madjs[i] for i in range(len(madjs))})
            fd.update({ph['fadj'][i]: fadjs[i] for i in range(len(fadjs))})
            fd.update({ph['ffields'][i]: ffields[i] for i in range(len(ffields))})
        fd[ph['labels']] = self.labels[fields[-1]]
        for i in range(self.L + 1):
            fd[ph['fields'][i]] = fields[i]
        return fd


class RefMult:
    """The reference ``Mult`` (gcn/mult.h:8-27)."""

    def __init__(self, prob):
        p = np.ascontiguousarray(prob, dtype=np.float32)
        self._h = lib().ref_mult_create(_fp(p), int(p.shape[0]))
        if not self._h:
            raise RuntimeError("Prob is empty")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_mult_destroy(self._h)
            self._h = None

    @property
    def bit(self):
        out = np.zeros(lib().ref_mult_bit_size(self._h), dtype=np.float32)
        lib().ref_mult_bit_copy(self._h, _fp(out))
        return out

    def query_u(self, u):
        return lib().ref_mult_query_u(self._h, float(u))

    def query(self):
        return lib().ref_mult_query(self._h)


def ref_slice(a, r):
    """gcn/_history.pyx:25-51 on the real c_indptr / c_slice."""
    import scipy.sparse as sp
    r = np.ascontiguousarray(r, dtype=np.int32)
    N = len(r)
    indptr = np.zeros(N + 1, dtype=np.int32)
    a_p = np.ascontiguousarray(a.indptr, dtype=np.int32)
    lib().ref_c_indptr(N, _ip(r), _ip(a_p), _ip(indptr))
    nnz = int(indptr[N])
    if nnz == 0:
        return sp.csr_matrix((N, a.shape[1]), dtype=a.dtype)
    data = np.zeros(nnz, dtype=np.float32)
    indices = np.zeros((nnz, 2), dtype=np.int32)
    a_d = np.ascontiguousarray(a.data, dtype=np.float32)
    a_i = np.ascontiguousarray(a.indices, dtype=np.int32)
    lib().ref_c_slice(N, _ip(r), _fp(a_d), _ip(a_i), _ip(a_p), _fp(data), _ip(indices), _ip(indptr))
    return indices, data, np.array([N, a.shape[1]], dtype=np.int32)


def ref_dense_slice(a, r):
    """gcn/_history.pyx:53-62 on the real c_dense_slice."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    r = np.ascontiguousarray(r, dtype=np.int32)
    out = np.zeros((len(r), a.shape[1]), dtype=np.float32)
    lib().ref_c_dense_slice(len(r), a.shape[1], _ip(r), _fp(a), _fp(out))
    return out
