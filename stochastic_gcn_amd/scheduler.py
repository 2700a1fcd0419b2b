"""``scheduler.PyScheduler`` -- drop-in for the reference's Cython class
(gcn/_scheduler.pyx:28-148) on top of the host sampler of libsgcn.so.

Same constructor, ``shuffle / minibatch / batch / get_feed_dict / get_t`` and the same
feed-dict: per layer ``adj=(idx[ne,2] int32, w f32, (n1,n0))``, ``madj``, ``fadj``, ``fields``
(L+1), ``ffields`` (L), ``scales`` (L), index 0 = input-most layer (gcn/_scheduler.pyx:121-126),
``labels[fields[-1]]`` (gcn/_scheduler.pyx:138).  Placeholders are opaque dict keys, exactly
as the reference treats them (gcn/test_scheduler.py:25-33 uses plain strings).

MI355X-first addition: next to every COO triple the feed-dict carries the same matrix as a
device-ready CSR bundle under ``('csr', placeholder)`` -- row pointer, column ids, values,
the transposed CSR for the backward SpMM and the host work plan for power-law rows -- which
is what the HIP kernels consume (``stochastic_gcn_amd.ops``).  Index arrays are bit-exact
with the reference for the same seed and call sequence (tests/test_sampler.py).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check, lib

# selectors of sgcn_sched_view_* (include/sgcn.h)
(I_FIELD, I_FFIELD, I_EDG_S, I_EDG_T, I_FEDG_S, I_FEDG_T, I_EDG_P, I_FEDG_P, I_ADJ_I,
 I_TEDG_P, I_TEDG_T) = range(11)
F_SCALES, F_EDG_W, F_MEDG_W, F_FEDG_W, F_ADJ_W, F_TEDG_W = range(6)


class HostCSR(object):
    """A host CSR bundle emitted by the sampler for one layer (all numpy, owned copies)."""
    __slots__ = ("shape", "rowptr", "col", "val", "t_rowptr", "t_col", "t_val")

    def __init__(self, shape, rowptr, col, val, t_rowptr=None, t_col=None, t_val=None):
        self.shape = shape
        self.rowptr, self.col, self.val = rowptr, col, val
        self.t_rowptr, self.t_col, self.t_val = t_rowptr, t_col, t_val

    @property
    def nnz(self):
        return int(self.col.shape[0])


def build_plan(rowptr, T=0):
    """Host work plan (include/sgcn.h sgcn_plan_*): returns (seg[nseg,4] int32,
    fix[nfix,3] int32, nslots)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    M = rowptr.shape[0] - 1
    nseg, nfix, nslots = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.sgcn_plan_count(rowptr.ctypes.data, M, T, C.byref(nseg), C.byref(nfix), C.byref(nslots)))
    seg = np.empty((nseg.value, 4), dtype=np.int32)
    fix = np.empty((nfix.value, 3), dtype=np.int32)
    check(lib.sgcn_plan_fill(rowptr.ctypes.data, M, T, seg.ctypes.data,
                             fix.ctypes.data if nfix.value else None))
    return seg, fix, nslots.value


class _Sampler(object):
    """Thin owner of an ``sgcn_sched_t`` handle."""

    def __init__(self, adj, num_data, L, cv, importance):
        w = np.ascontiguousarray(adj.data, dtype=np.float32)
        i = np.ascontiguousarray(adj.indices, dtype=np.int32)
        p = np.ascontiguousarray(adj.indptr, dtype=np.int32)
        if p.shape[0] != num_data + 1:
            raise ValueError("adjacency has %d rows, labels have %d" % (p.shape[0] - 1, num_data))
        self.cv = bool(cv)
        self._h = C.c_void_p()
        check(lib.sgcn_sched_create(w.ctypes.data, i.ctypes.data, p.ctypes.data, int(num_data),
                                    int(w.shape[0]), int(L), int(cv), int(importance),
                                    C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:          # (interpreter shutdown clears module globals before the last objects die)
            lib.sgcn_sched_destroy(h)
            self._h = None

    def seed(self, s):
        check(lib.sgcn_sched_seed(self._h, int(s)))

    def start_batch(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        check(lib.sgcn_sched_start_batch(self._h, int(ids.shape[0]), ids.ctypes.data))

    def expand(self, degree):
        check(lib.sgcn_sched_expand(self._h, int(degree)))

    def ivec(self, which):
        ptr, n = _ffi.c_i32p(), C.c_int64()
        check(lib.sgcn_sched_view_i32(self._h, which, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    def fvec(self, which):
        ptr, n = _ffi.c_f32p(), C.c_int64()
        check(lib.sgcn_sched_view_f32(self._h, which, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.float32)
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()


class PyScheduler(object):
    def __init__(self, adj, labels, L, degrees, placeholders, seed, data=None, cv=False,
                 importance=False):
        self.c_sch = _Sampler(adj, labels.shape[0], L, cv, importance)
        self.c_sch.seed(seed)
        self.labels = labels
        self.data = data
        self.degrees = degrees
        self.L = L
        self.start = 0
        self.placeholders = placeholders
        self.t = 0

    def shuffle(self):
        # NumPy's global RNG, as the reference (gcn/_scheduler.pyx:50-53)
        np.random.shuffle(self.data)
        self.start = 0
        self.t = 0

    def batch(self, data):
        s = self.c_sch
        data = np.ascontiguousarray(data, dtype=np.int32)
        fields, ffields, adjs, madjs, fadjs, scales = [data], [], [], [], [], []
        csr_adj, csr_fadj = [], []
        s.start_batch(data)
        for l in range(self.L):
            s.expand(self.degrees[self.L - l - 1])
            fields.append(s.ivec(I_FIELD))
            scales.append(s.fvec(F_SCALES))
            edg_s, edg_t, edg_w = s.ivec(I_EDG_S), s.ivec(I_EDG_T), s.fvec(F_EDG_W)
            edg_i = np.empty((edg_s.shape[0], 2), dtype=np.int32)
            edg_i[:, 0] = edg_s
            edg_i[:, 1] = edg_t
            shape = (fields[-2].shape[0], fields[-1].shape[0])
            adjs.append((edg_i, edg_w, shape))
            csr_adj.append(HostCSR(shape, s.ivec(I_EDG_P), edg_t, edg_w,
                                   s.ivec(I_TEDG_P), s.ivec(I_TEDG_T), s.fvec(F_TEDG_W)))
            if s.cv:
                ffields.append(s.ivec(I_FFIELD))
                fedg_s, fedg_t, fedg_w = s.ivec(I_FEDG_S), s.ivec(I_FEDG_T), s.fvec(F_FEDG_W)
                fedg_i = np.empty((fedg_s.shape[0], 2), dtype=np.int32)
                fedg_i[:, 0] = fedg_s
                fedg_i[:, 1] = fedg_t
                fshape = (fields[-2].shape[0], ffields[-1].shape[0])
                madjs.append((np.copy(edg_i), s.fvec(F_MEDG_W), np.copy(shape)))
                fadjs.append((fedg_i, fedg_w, fshape))
                csr_fadj.append(HostCSR(fshape, s.ivec(I_FEDG_P), fedg_t, fedg_w))
        for lst in (fields, ffields, adjs, madjs, fadjs, scales, csr_adj, csr_fadj):
            lst.reverse()
        fd = self.get_feed_dict(fields, ffields, adjs, madjs, fadjs, scales)
        ph = self.placeholders
        for i in range(self.L):
            fd[('csr', ph['adj'][i])] = csr_adj[i]
        if s.cv:
            for i in range(len(csr_fadj)):
                fd[('csr', ph['fadj'][i])] = csr_fadj[i]
        return fd

    def minibatch(self, batch_size):
        if self.start == self.data.shape[0]:
            return None
        end = min(self.data.shape[0], self.start + batch_size)
        batch = self.data[self.start:end]
        self.start = end
        return self.batch(batch)

    def get_feed_dict(self, fields, ffields, adjs, madjs, fadjs, scales):
        ph = self.placeholders
        labels = self.labels[fields[-1]]
        feed_dict = {ph['adj'][i]: adjs[i] for i in range(self.L)}
        feed_dict.update({ph['scales'][i]: scales[i] for i in range(len(scales))})
        if self.c_sch.cv:
            feed_dict.update({ph['madj'][i]: madjs[i] for i in range(len(madjs))})
            feed_dict.update({ph['fadj'][i]: fadjs[i] for i in range(len(fadjs))})
            feed_dict.update({ph['ffields'][i]: ffields[i] for i in range(len(ffields))})
        feed_dict[ph['labels']] = labels
        for i in range(self.L + 1):
            feed_dict[ph['fields'][i]] = fields[i]
        return feed_dict

    def get_t(self):
        return self.t


class Mult(object):
    """Fenwick multinomial sampler (gcn/mult.h:8-27) -- exposed for parity tests."""

    def __init__(self, prob):
        p = np.ascontiguousarray(prob, dtype=np.float32)
        self._h = C.c_void_p()
        check(lib.sgcn_mult_create(p.ctypes.data, int(p.shape[0]), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.sgcn_mult_destroy(h)
            self._h = None

    @property
    def bit(self):
        ptr, n = _ffi.c_f32p(), C.c_int64()
        check(lib.sgcn_mult_tree(self._h, C.byref(ptr), C.byref(n)))
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    def query_u(self, u):
        r = C.c_int32()
        check(lib.sgcn_mult_query_u(self._h, float(u), C.byref(r)))
        return r.value

    def query(self):
        r = C.c_int32()
        check(lib.sgcn_mult_query(self._h, C.byref(r)))
        return r.value


# ---- packed minibatch: the fast path of the training loop ----------------------------------------
CSR_DESC = 11   # nrows ncols nnz rowptr col val seg nseg fix nfix nslots  (include/sgcn.h)


class StagingSlot(object):
    """Reusable (pinned when CUDA is present) host staging buffer for one in-flight batch: the
    int32 section and the fp32 section share ONE allocation ``[n_i ints | n_f floats]`` so the
    batch crosses PCIe in a single copy.  Producers touch it through raw addresses only."""

    def __init__(self, pin, words=0):
        self.pin = pin
        self.buf = None                    # torch int32 tensor (raw 4-byte words)
        self.np = None                     # the same memory as a NumPy int32 array
        self.ptr, self.cap = 0, 0
        self.event = None                  # recorded by the consumer after its H2D copy
        if words:
            self._alloc(words)

    def _alloc(self, words):
        import torch
        self.buf = torch.empty(int(words), dtype=torch.int32, pin_memory=self.pin)
        self.np = self.buf.numpy()
        self.ptr = self.buf.data_ptr()
        self.cap = int(self.buf.numel())

    def wait(self):
        """The previous H2D copy out of this slot must be done before it is overwritten."""
        if self.event is not None:
            self.event.synchronize()
            self.event = None

    def ensure(self, n_i, n_f):
        """Make room for the batch; returns (address of the int32 section, of the fp32 section)."""
        self.wait()
        need = n_i + n_f
        if self.buf is None or self.cap < need:
            self._alloc(max(int(need * 1.5), 1 << 18))
        return self.ptr, self.ptr + 4 * n_i


class PackedBatch(object):
    """One minibatch as two flat host arrays + a descriptor table (sgcn_sched_batch_packed).
    ``m`` is the descriptor table as a plain list of Python ints (what the launching thread
    reads); the NumPy views of the sections are made on demand (tests, feed_dict())."""

    def __init__(self, L, cv, meta, ibuf, fbuf, n_i, n_f, slot=None):
        self.L, self.cv, self.meta = L, cv, meta
        self._ibuf, self._fbuf, self.n_i, self.n_f = ibuf, fbuf, n_i, n_f
        self.slot = slot
        self._m = self._meta_ptr = None
        o = 4 + 2 * (L + 1)
        self.o_scales = o; o += 2 * L
        self.o_ffields = o; o += 2 * L
        self.o_labels = o; o += 3
        self.o_medg = o; o += 2 * L
        self.o_csr = o; o += 3 * L * CSR_DESC
        self.o_tmedg = o                  # (ABI v16) medg in the order of adj^T's nonzeros, L x (off, len)

    # lazily built host views --------------------------------------------------------------------
    @property
    def m(self):
        if self._m is None:
            self._m = self.meta.tolist()
        return self._m

    @property
    def meta_ptr(self):
        if self._meta_ptr is None:
            self._meta_ptr = self.meta.ctypes.data
        return self._meta_ptr

    @property
    def ibuf(self):
        if self._ibuf is None:
            self._ibuf = self.slot.np[:max(self.n_i, 1)]
        return self._ibuf

    @property
    def fbuf(self):
        if self._fbuf is None:
            n_i = max(self.n_i, 1)
            self._fbuf = self.slot.np[n_i:n_i + max(self.n_f, 1)].view(np.float32)
        return self._fbuf

    @property
    def _fields(self):
        return self.meta[4:4 + 2 * (self.L + 1)].reshape(self.L + 1, 2)

    @property
    def _scales(self):
        return self.meta[self.o_scales:self.o_scales + 2 * self.L].reshape(self.L, 2)

    @property
    def _ffields(self):
        return self.meta[self.o_ffields:self.o_ffields + 2 * self.L].reshape(self.L, 2)

    @property
    def _labels(self):
        return self.meta[self.o_labels:self.o_labels + 3]

    @property
    def _medg(self):
        return self.meta[self.o_medg:self.o_medg + 2 * self.L].reshape(self.L, 2)

    @property
    def _tmedg(self):
        return self.meta[self.o_tmedg:self.o_tmedg + 2 * self.L].reshape(self.L, 2)

    @property
    def _csr(self):
        return self.meta[self.o_csr:self.o_csr + 3 * self.L * CSR_DESC].reshape(self.L, 3, CSR_DESC)

    # numpy views (host) -------------------------------------------------------------------------
    def _i(self, off, n):
        return self.ibuf[off:off + n]

    def _f(self, off, n):
        return self.fbuf[off:off + n]

    def field(self, l):
        return self._i(*self._fields[l])

    def ffield(self, l):
        return self._i(*self._ffields[l])

    def scale(self, l):
        return self._f(*self._scales[l])

    def labels(self):
        off, r, c = self._labels
        return self._f(off, r * c).reshape(r, c)

    def csr(self, l, which):
        """HostCSR view of layer l: which = 0 adj, 1 adj^T, 2 fadj."""
        d = self._csr[l, which]
        return HostCSR((int(d[0]), int(d[1])), self._i(d[3], d[0] + 1), self._i(d[4], d[2]),
                       self._f(d[5], d[2]))

    def feed_dict(self, placeholders, labels=None):
        """The reference-format feed-dict (gcn/_scheduler.pyx:137-148) rebuilt from the packed
        arrays (tests compare it with PyScheduler.batch)."""
        ph, L = placeholders, self.L
        fd = {}
        for l in range(L):
            a = self.csr(l, 0)
            rows = np.repeat(np.arange(a.shape[0], dtype=np.int32), np.diff(a.rowptr))
            idx = np.stack([rows, a.col], axis=1).astype(np.int32).reshape(-1, 2)
            fd[ph['adj'][l]] = (idx, a.val.copy(), a.shape)
            fd[ph['scales'][l]] = self.scale(l).copy()
            if self.cv:
                p = self.csr(l, 2)
                prow = np.repeat(np.arange(p.shape[0], dtype=np.int32), np.diff(p.rowptr))
                fd[ph['fadj'][l]] = (np.stack([prow, p.col], axis=1).astype(np.int32).reshape(-1, 2),
                                     p.val.copy(), p.shape)
                fd[ph['madj'][l]] = (idx.copy(), self._f(*self._medg[l]).copy(), np.array(a.shape))
                fd[ph['ffields'][l]] = self.ffield(l).copy()
        for l in range(L + 1):
            fd[ph['fields'][l]] = self.field(l).copy()
        fd[ph['labels']] = self.labels().copy() if self._labels[1] else (
            labels[self.field(L)] if labels is not None else None)
        return fd


def _batch_packed(self, data, plan_T=0, slot=None):
    """One C call for the whole minibatch (no per-array Python work, runs without the GIL)."""
    data = np.ascontiguousarray(data, dtype=np.int32)
    L = self.L
    self._packed_setup()
    meta = np.zeros(self._meta_len, dtype=np.int64)
    n_i, n_f = C.c_int64(), C.c_int64()
    if slot is not None:
        # ONE foreign call per minibatch: sampling + CSR/plan packing + copy into the pinned slot
        slot.wait()
        rc = lib.sgcn_sched_batch_packed_into(self.c_sch._h, int(data.shape[0]), data.ctypes.data, L,
                                              self._deg32_ptr, self._lab32_ptr, self._lab32_cols, int(plan_T),
                                              meta.ctypes.data, self._meta_len, slot.ptr, slot.cap,
                                              C.byref(n_i), C.byref(n_f))
        if rc < 0:
            check(rc)
        if rc == 1:         # first use / outgrown: grow the slot, then copy
            pi, pf = slot.ensure(max(n_i.value, 1), max(n_f.value, 1))
            check(lib.sgcn_sched_packed_copy(self.c_sch._h, pi, pf))
        return PackedBatch(L, self.c_sch.cv, meta, None, None, n_i.value, n_f.value, slot)
    check(lib.sgcn_sched_batch_packed(self.c_sch._h, int(data.shape[0]), data.ctypes.data, L,
                                      self._deg32_ptr, self._lab32_ptr, self._lab32_cols, int(plan_T),
                                      meta.ctypes.data, self._meta_len, C.byref(n_i), C.byref(n_f)))
    n_i, n_f = n_i.value, n_f.value
    ib = np.empty(max(n_i, 1), dtype=np.int32)
    fb = np.empty(max(n_f, 1), dtype=np.float32)
    check(lib.sgcn_sched_packed_copy(self.c_sch._h, ib.ctypes.data, fb.ctypes.data))
    return PackedBatch(L, self.c_sch.cv, meta, ib, fb, n_i, n_f, None)


def _minibatch_packed(self, batch_size, plan_T=0, slot=None):
    if self.start == self.data.shape[0]:
        return None
    end = min(self.data.shape[0], self.start + batch_size)
    batch = self.data[self.start:end]
    self.start = end
    return self.batch_packed(batch, plan_T, slot)


def default_packers():
    """Packer threads beside the sampler's core thread: three when the process has the cores for them.  A data-parallel job
    runs one process per GPU on ONE host: with 8 ranks on a box whose container grants 16 cores, 8 x (launching thread +
    core + 3 packers) = 40 runnable threads would time-slice -- each rank then gets what is left of its share after the
    launching thread and the core (the sample sequence does not depend on the count: sgcn_prefetch_start)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))
    return int(min(3, max(0, n // ranks - 2)))


class NativePrefetcher(object):
    """An epoch's minibatches from the C++ sampler thread(s) of libsgcn.so (sgcn_prefetch_*).

    One scheduler: the same sample sequence as calling ``batch_packed`` in a loop, but no Python
    runs on the producer, so it never competes with the launching thread for the interpreter lock.
    A list of N schedulers (independent private CSR copies and RNG streams): thread k builds batches
    k, k + N, ... concurrently and the consumer still receives them in order -- the NON-PARITY fast
    mode (``--sampler_threads N``) for sampler-bound runs (non-PP / NS / Exact).

    ``next()`` blocks in C (lock released) and returns a PackedBatch living in one of the pinned
    staging slots; a slot goes back to the producers ``lag`` batches after it was handed out, once
    the H2D copy the consumer recorded on it has completed."""

    def __init__(self, sch, batches, plan_T=0, depth=2, pin=True, lag=2, packers=None):
        schs = list(sch) if isinstance(sch, (list, tuple)) else [sch]
        if packers is None:       # one sampler: its core on one thread, the packing on three others (same bits)
            packers = default_packers() if len(schs) == 1 else 0
        if len(schs) > 1:
            packers = 0
        self.packers = int(packers)
        for s_ in schs:
            s_._packed_setup()
        sch = schs[0]
        self.sch, self.L, self.lag = sch, sch.L, lag
        self.n_batches = len(batches)
        ids = np.ascontiguousarray(np.concatenate(batches) if batches else np.zeros(0), dtype=np.int32)
        off = np.zeros(len(batches) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in batches], out=off[1:])
        n_slots = max(1, depth) * max(len(schs), self.packers) + 1 + lag
        words = int(getattr(sch, "_slot_words", 0) or (1 << 20))
        pool = getattr(sch, "_slot_pool", None)
        if pool is None or len(pool) != n_slots or pool[0].cap < words or pool[0].pin != pin:
            pool = [StagingSlot(pin, words) for _ in range(n_slots)]
            sch._slot_pool = pool
        self.slots = pool
        for sl in pool:
            sl.wait()
        ptrs = (C.c_void_p * n_slots)(*[sl.ptr for sl in pool])
        caps = (C.c_int64 * n_slots)(*[sl.cap for sl in pool])
        handles = (C.c_void_p * len(schs))(*[s_.c_sch._h for s_ in schs])
        self._h = C.c_void_p()
        check(lib.sgcn_prefetch_start(handles, len(schs), len(batches), ids.ctypes.data, off.ctypes.data, self.L,
                                      sch._deg32_ptr, sch._lab32_ptr, sch._lab32_cols, int(plan_T), n_slots,
                                      ptrs, caps, int(lag), self.packers, C.byref(self._h)))
        self._keep = schs
        self.pending = []
        self.max_words = 0
        # next() runs on the launching thread, once per step: its out-parameters are made once
        self._out = (C.c_int32(), C.c_int64(), C.c_int64(), C.c_void_p())
        self._out_ref = tuple(C.byref(o) for o in self._out)
        self._meta_len, self._cv = int(sch._meta_len), sch.c_sch.cv

    def next(self):
        if self._h is None:
            return None
        while len(self.pending) > self.lag:            # hand old slots back to the producer
            idx = self.pending.pop(0)
            self.slots[idx].wait()
            check(lib.sgcn_prefetch_release(self._h, idx))
        meta = np.empty(self._meta_len, dtype=np.int64)          # (sgcn_prefetch_next writes all of it)
        meta_ptr = meta.ctypes.data
        slot, n_i, n_f, spill = self._out
        rc = lib.sgcn_prefetch_next(self._h, self._out_ref[0], meta_ptr, self._out_ref[1], self._out_ref[2], self._out_ref[3])
        if rc == 1:
            self.close()
            return None
        if rc:
            check(rc)
        n_i, n_f, idx = n_i.value, n_f.value, slot.value
        self.max_words = max(self.max_words, max(n_i, 1) + max(n_f, 1))
        if spill.value:      # outgrew the slot: copy out of the producer's heap buffer, free the slot at once
            ni, nf = max(n_i, 1), max(n_f, 1)
            raw = np.ctypeslib.as_array((C.c_int32 * (ni + nf)).from_address(spill.value)).copy()
            check(lib.sgcn_prefetch_release(self._h, idx))
            return PackedBatch(self.L, self.sch.c_sch.cv, meta, raw[:ni], raw[ni:].view(np.float32), n_i, n_f, None)
        self.pending.append(idx)
        pb = PackedBatch(self.L, self._cv, meta, None, None, n_i, n_f, self.slots[idx])
        pb._meta_ptr = meta_ptr
        return pb

    def close(self):
        if self._h is not None:
            st = (C.c_double * 4)()
            lib.sgcn_prefetch_stats(self._h, st)
            self.stats = dict(wait_slot_s=st[0], pack_s=st[1], copy_s=st[2], sample_s=st[3], packers=self.packers)
            lib.sgcn_prefetch_stop(self._h)
            self._h = None
            if self.max_words > self.slots[0].cap:      # size the next epoch's slots for what we saw
                self.sch._slot_words = int(self.max_words * 1.5)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _packed_setup(self):
    if getattr(self, "_deg32", None) is None:
        self._deg32 = np.ascontiguousarray(self.degrees, dtype=np.int32)
        lab = self.labels
        self._lab32 = lab if (isinstance(lab, np.ndarray) and lab.dtype == np.float32
                              and lab.flags['C_CONTIGUOUS']) else np.ascontiguousarray(lab, dtype=np.float32)
        self._meta_len = int(lib.sgcn_sched_packed_meta_len(self.L))
        self._deg32_ptr, self._lab32_ptr = self._deg32.ctypes.data, self._lab32.ctypes.data
        self._lab32_cols = int(self._lab32.shape[1])


PyScheduler._packed_setup = _packed_setup
PyScheduler.batch_packed = _batch_packed
PyScheduler.minibatch_packed = _minibatch_packed
