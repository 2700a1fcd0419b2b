"""The training step as a PROGRAM for the library's native launch loop (include/sgcn.h sgcn_step_run,
csrc/sgcn_step.cpp).

The reference executes a step as one ``sess.run`` of a static graph (gcn/vrgcn.py:72-82).  The eager
path of this package walks ``model.layers`` in Python every step and is host-bound (0.30 ms of
interpreter time around 0.26 ms of GPU work per Reddit CVD+PP step).  ``StepProgram`` walks the layers
ONCE, symbolically -- the same decisions as ``layers.py`` (pending dropout rides into the consuming
GEMM, the first dense layer reads its rows through the field index, the CVD streams run stacked ...) --
and emits the flat list of C-ABI calls a step consists of, with every argument affine in a small slot
table.  Per step the host then (1) copies the minibatch's staging buffer to the device, (2) fills the
slot table from the sampler's descriptor (three NumPy operations) and (3) makes ONE foreign call.

Intermediates live in a preallocated arena sized from the flags' row bounds (|field l| <= batch *
prod(1 + degree)); a minibatch that does not fit, a feed-dict that is not a PackedBatch, or a layer
stack outside the supported set (sparse input features, hidden width > 128 with LayerNorm) falls back to the eager path, which stays the reference implementation: both
run the same kernels with the same arguments and are bit-identical (tests/test_step_program_gpu.py).
"""
import ctypes as C

import numpy as np
import torch

from . import ops
from ._ffi import StepFill, StepOp, check, lib
from .flags import FLAGS
from .layers import AugmentedDropoutDense, Dense, DetDropoutFC, Dropout, PlainAggregator, VRAggregator
from .scheduler import CSR_DESC, PackedBatch

OP = dict(DENSE_FWD=1, DENSE_BWD=2, VR_AGG=3, SPMM=4, SOFTMAX_CE=5, ADAM=6, SCATTER_ROWS=7, MEMSET0=8,
          DROPOUT=9, L2_PENALTY=10, GATHER_ROWS=11, COPY2D=12, SIGMOID_CE=13, VR_AGG_PRE=14, VR_AGG_POST=15,
          AUX_SCATTER_ROWS=16, AUX_MEMSET0=17, DW_FLUSH=21, GRAD_STORE=22, MODE=23,
          CSR_SLICE=24, LN_ACT_FWD=25, LN_ACT_BWD=26, CSR_TRANSPOSE=27, GATHER_F32=28,       # 18-20: retired (include/sgcn.h)
          ALLREDUCE_AVG=29, HIST_PACK=30, ALLGATHER_I32=31, HIST_APPLY=32,
          # the --det_dropout stacks (ABI v16)
          GEMM=33, DET_PRE=34, DET_PRE_BWD=35, SQUARE=36, ADDMUL=37, DET_LNVAR_FWD=38, DET_LNVAR_BWD=39, DET_RELU_FWD=40,
          DET_RELU_BWD=41, GAUSS=42, GAUSS_BWD=43, DET_AGG_PREP=44, DET_AGG_PREP_BWD=45, RELU_EPS=46, GATE=47)
MAX_ARGS = 48
GEMM_WS_BOUND = 256 * 32 * 128 + 64    # sgcn_gemm_ws_floats(M, N, K) = S * M * N with S <= 256 / (tiles of 32 x 128): never above this
ARENA_LIMIT_BYTES = 2 << 30


class Unsupported(Exception):
    pass


def _fbits(x):
    return int(np.float32(x).view(np.uint32))


def K(v):
    """constant operand"""
    return (0, -1, int(v))


NULL = K(0)


class Rows(object):
    """row count  mul * slots[slot] + add  with its capacity (upper bound used to size buffers)"""
    __slots__ = ("mul", "slot", "add", "cap")

    def __init__(self, mul, slot, add, cap):
        self.mul, self.slot, self.add, self.cap = mul, slot, add, cap

    def op(self):
        return (self.mul, self.slot, self.add)

    def __add__(self, o):
        if self.slot != o.slot and self.slot >= 0 and o.slot >= 0:
            raise Unsupported("row count over two slots")
        return Rows(self.mul + o.mul, self.slot if self.slot >= 0 else o.slot, self.add + o.add, self.cap + o.cap)


class ST(object):
    """symbolic 2-D fp32 activation: address (affine), rows, cols, ld (floats)"""
    __slots__ = ("ptr", "rows", "cols", "ld")

    def __init__(self, ptr, rows, cols, ld):
        self.ptr, self.rows, self.cols, self.ld = ptr, rows, cols, ld

    def head(self, rows):                       # t[:rows]
        return ST(self.ptr, rows, self.cols, self.ld)

    def tail(self, start):                      # t[start:], start a Rows over one slot; self.ptr constant
        if self.ptr[1] >= 0:
            raise Unsupported("row offset of a slot-addressed tensor")
        rest = Rows(self.rows.mul - start.mul, self.rows.slot, self.rows.add - start.add, self.rows.cap - start.cap)
        return ST((start.mul * self.ld * 4, start.slot, self.ptr[2] + start.add * self.ld * 4), rest, self.cols, self.ld)

    def cols_from(self, c0, ncols):             # t[:, c0:c0+ncols]
        return ST((self.ptr[0], self.ptr[1], self.ptr[2] + 4 * c0), self.rows, ncols, self.ld)


class SGather(object):
    """src[idx] that has not been gathered (ops.GatheredRows)"""
    __slots__ = ("src", "idx", "rows", "_m")

    def __init__(self, src, idx, rows):
        self.src, self.idx, self.rows, self._m = src, idx, rows, None

    @property
    def cols(self):
        return self.src.cols


class SSparse(object):
    """the minibatch's row slice of the sparse feature matrix (layers.SparseInput): CSR arrays in the arena, sized for the
    worst minibatch; ``nnz`` is a slot the host fills per step; the transpose index is emitted once, by the first consumer"""
    __slots__ = ("o_p", "o_c", "o_d", "o_r", "rows", "nnz", "ncols", "cap", "vals", "tbox")

    def __init__(self, o_p, o_c, o_d, o_r, rows, nnz, ncols, cap, vals=None, tbox=None):
        self.o_p, self.o_c, self.o_d, self.o_r, self.rows, self.nnz, self.ncols, self.cap = o_p, o_c, o_d, o_r, rows, nnz, ncols, cap
        self.vals = o_d if vals is None else vals          # the values in use (after a sparse dropout: a second vector)
        self.tbox = [None] if tbox is None else tbox       # the transpose index, shared by the views of one slice

    def with_vals(self, vals):
        return SSparse(self.o_p, self.o_c, self.o_d, self.o_r, self.rows, self.nnz, self.ncols, self.cap, vals, self.tbox)


class SDet(object):
    """(mean, variance) of a --det_dropout activation: two dense STs of one shape"""
    __slots__ = ("mu", "var")

    def __init__(self, mu, var):
        self.mu, self.var = mu, var


class SDropped(object):
    """activation with a pending dropout (layers.Dropped)"""
    __slots__ = ("x", "site", "_m")

    def __init__(self, x, site):
        self.x, self.site, self._m = x, site, None


class StepProgram(object):
    def __init__(self, model, dropout):
        self.model = m = model
        self.dropout = float(dropout)
        from . import ops as _ops
        self.sparse = bool(m.sparse_input and m.sparse_mm)
        if self.sparse and not (isinstance(m.features_dev, _ops.DeviceCSR) and m.features_dev.host_rowptr is not None):
            raise Unsupported("sparse features are not a device CSR with its host row pointer")
        if not self.sparse and not isinstance(m.features_dev, torch.Tensor):
            raise Unsupported("features are not a dense device tensor")
        self.dev = m.device
        self.L = m.L
        self.cv = bool(m._history)
        # row bounds from the flags (same bound parallel.py uses for the history exchange)
        deg = FLAGS.degree if m.is_training else FLAGS.test_degree
        bs = FLAGS.batch_size if m.is_training else FLAGS.test_batch_size
        n = int(m.num_data)
        caps = [0] * (self.L + 1)
        caps[self.L] = min(n, int(bs))
        for l in range(self.L - 1, -1, -1):
            caps[l] = min(n, caps[l + 1] * (1 + int(deg)))
        self.caps = caps
        if self.sparse:
            # the slice's nonzeros: counted on the host per step (fill), bounded here by the caps[0] longest feature rows
            self._rowlen = np.diff(np.asarray(m.features_dev.host_rowptr, dtype=np.int64))
            k = min(caps[0], int(self._rowlen.shape[0]))
            self.nnz_cap = int(np.sort(self._rowlen)[::-1][:k].sum()) if k else 0
            if self.nnz_cap >= (1 << 31):
                raise Unsupported("feature slice beyond 2^31 nonzeros")
        self._pb = PackedBatch(self.L, self.cv, np.zeros(int(lib.sgcn_sched_packed_meta_len(self.L)), np.int64),
                               None, None, 0, 0, None)          # only for its offsets
        self.plan_ws_floats = 8 << 20          # partial sums of split rows (checked per step)
        self.ws_plan = torch.empty(self.plan_ws_floats, dtype=torch.float32, device=self.dev)
        # pass 1 (addresses relative to 0) sizes the arena AND the dense ops' scratch (split-K partial tiles,
        # LayerNorm-backward partial sums: evaluated at every op's row CAPACITY, so a batch that fits() can never
        # fail inside sgcn_step_run for want of scratch) and numbers the slots; pass 2 emits the program
        self.gemm_ws_floats, self._ws_gemm_ptr, self._ws_need = 0, 0, 0
        self._arena_base, self._n_meta, self._key_layers = 0, 0, []
        self._reset()
        self._build()
        if self._arena_off * 4 > ARENA_LIMIT_BYTES:
            raise Unsupported("activation arena of %.1f GB" % (self._arena_off * 4 / 2 ** 30))
        if self._ws_need * 4 > ARENA_LIMIT_BYTES:
            raise Unsupported("dense-layer scratch of %.1f GB" % (self._ws_need * 4 / 2 ** 30))
        self.gemm_ws_floats = max(self._ws_need, 4 << 20)
        self.ws_gemm = torch.empty(self.gemm_ws_floats, dtype=torch.float32, device=self.dev)
        self._ws_gemm_ptr = self.ws_gemm.data_ptr()
        self.arena = torch.empty(max(self._arena_off, 64), dtype=torch.float32, device=self.dev)
        self._arena_base, self._n_meta, self._key_layers = self.arena.data_ptr(), len(self._fill), sorted(self._keys_seen)
        self._reset()
        self._build()
        assert len(self._fill) == self._n_meta and sorted(self._keys_seen) == self._key_layers
        self._finalize()

    def _reset(self):
        self.ops_fb, self.ops_opt, self.ops_hist = [], [], []
        self._cur = self.ops_fb
        self._slot_of, self._fill = {}, []   # slot table: [meta-derived ...][dropout keys ...][lr_t]; fill = (meta index, mul, 0 | 1 ip | 2 fp)
        self._keys_seen = set()
        self._arena_off = 0
        self._nslot_checks = []              # (meta index of nslots, ldw)
        self._cap_checks = []                # (meta index of a row count, capacity)
        self.rows = []
        for l in range(self.L + 1):
            sl = self._slot('n', 5 + 2 * l)
            self._cap_checks.append((5 + 2 * l, self.caps[l]))
            self.rows.append(Rows(1, sl, 0, self.caps[l]))

    # ---- slot bookkeeping ----------------------------------------------------------------------
    def _slot(self, kind, mi):
        key = (kind, mi)
        if key not in self._slot_of:
            self._slot_of[key] = len(self._fill)
            self._fill.append((mi, 4 if kind in ('ip', 'fp') else 1, {'n': 0, 'ip': 1, 'fp': 2}[kind]))
        return self._slot_of[key]

    def _n(self, mi):
        return (1, self._slot('n', mi), 0)

    def _ip(self, mi):
        return (1, self._slot('ip', mi), 0)

    def _fp(self, mi):
        return (1, self._slot('fp', mi), 0)

    def _field_ptr(self, l):
        return self._ip(4 + 2 * l)

    def _csr(self, l, which):
        """descriptor base of layer l's adj (0) / adj^T (1) / fadj (2)"""
        return self._pb.o_csr + (3 * l + which) * CSR_DESC

    def _plan(self, b, d):
        self._nslot_checks.append((b + 10, (d + 3) // 4 * 4))
        return [K(1), self._ip(b + 6), self._n(b + 7), self._ip(b + 8), self._n(b + 9), self._n(b + 10),
                K(self.ws_plan.data_ptr()), K(self.plan_ws_floats)]

    def _key(self, layer_index):
        self._keys_seen.add(layer_index)
        pos = self._key_layers.index(layer_index) if layer_index in self._key_layers else 0
        return (1, self._n_meta + pos, 0)

    def _lr(self):
        return (1, self._n_meta + len(self._key_layers), 0)

    def _nnz(self):
        """the feature slice's nonzero count (sparse programs: the slot behind the step size)"""
        return (1, self._n_meta + len(self._key_layers) + 1, 0)

    # ---- arena ------------------------------------------------------------------------------------
    def _alloc(self, rows, cols, ld=None):
        ld = cols if ld is None else ld
        off = self._arena_off
        self._arena_off += (rows.cap * ld + 63) // 64 * 64
        return ST(K(self._arena_base + 4 * off), rows, cols, ld)

    def _alloc_vec(self, n_cap):
        """(operand, offset in floats) of a 1-D scratch vector"""
        off = self._arena_off
        self._arena_off += (n_cap + 63) // 64 * 64
        return K(self._arena_base + 4 * off), off

    # ---- op emission ------------------------------------------------------------------------------
    def _emit(self, op, args):
        if len(args) > MAX_ARGS:
            raise Unsupported("too many arguments")
        self._cur.append((OP[op], list(args)))

    def _p(self, t):
        return NULL if t is None else (t.ptr if isinstance(t, ST) else t)

    def _drop_args(self, site, rows_op, width):
        if site is None:
            return [K(0), K(0), K(_fbits(1.0)), K(-1), K(width)]
        keep, li = site
        return [K(1), self._key(li), K(_fbits(keep)), rows_op, K(width)]

    def _site(self, layer):
        keep = 1.0 - self.dropout
        if keep >= 1.0:
            return None
        return (keep, layer.index)

    def _materialize(self, x):
        """dense ST from a pending dropout / pending gather (layers.dense_of)"""
        if isinstance(x, SGather):
            if x._m is None:
                out = self._alloc(x.rows, x.src.cols)
                self._emit('GATHER_ROWS', [self._p(x.src), K(x.src.ld), x.idx, x.rows.op(), K(x.src.cols), self._p(out), K(out.ld)])
                x._m = out
            return x._m
        if isinstance(x, SDropped):
            if x._m is None:
                inner = self._materialize(x.x)
                out = self._alloc(inner.rows, inner.cols)
                self._emit('DROPOUT', [self._p(inner), K(inner.ld), inner.rows.op(), K(inner.cols)]
                           + self._drop_args(x.site, K(-1), inner.cols) + [self._p(out), K(out.ld)])
                x._m = out
            return x._m
        return x

    def _dense_fwd(self, x, W, off, sc, relu, x2=None, site=None):
        """ops.dense_fwd: returns (y, (xhat, rstd) or None)"""
        g1 = g2 = NULL
        if isinstance(x2, SGather):
            x2, g2, r2 = x2.src, x2.idx, x2.rows
        elif x2 is not None:
            r2 = x2.rows
        if isinstance(x, SGather):
            x, g1, r1 = x.src, x.idx, x.rows
        else:
            r1 = x.rows
        Kd, N = x.cols, W.cols
        M = r1 if x2 is None else r1 + r2
        self._ws_need = max(self._ws_need, GEMM_WS_BOUND)
        y = self._alloc(M, N)
        norm = off is not None
        xhat = self._alloc(M, N) if norm else None
        rstd = self._alloc_vec(M.cap)[0] if norm else None
        self._emit('DENSE_FWD', [M.op(), K(N), K(Kd), self._p(x), K(x.ld), self._p(x2), K(x2.ld if x2 is not None else 0),
                                 r1.op(), self._p(W), K(W.ld), self._p(off), self._p(sc), K(_fbits(1e-9)), K(int(bool(relu))),
                                 self._p(y), K(N), self._p(xhat), self._p(rstd)]
                   + self._drop_args(site, r1.op(), Kd) + [K(self._ws_gemm_ptr), K(self.gemm_ws_floats), g1, g2])
        return y, ((xhat, rstd) if norm else None)

    def _dense_bwd(self, dy, y, ctx, sc, relu, x, W, dW, doff, dsc, need_dx, site):
        """ops.dense_bwd: returns dx or None"""
        gidx = NULL
        if isinstance(x, SGather):
            x, gidx = x.src, x.idx
        n, N, Kd = dy.rows, dy.cols, x.cols
        norm = ctx is not None
        pre = norm or bool(relu)
        # LayerNorm-backward partials grow with the rows (monotonic: evaluated at the capacity); the split-K scratch
        # of either GEMM is bounded whatever the batch
        self._ws_need = max(self._ws_need, ((int(lib.sgcn_ln_act_bwd_ws_floats(n.cap, N)) + 3) // 4 * 4 if norm else 0)
                            + GEMM_WS_BOUND)
        g_tmp = self._alloc(n, N) if pre else None
        dx = self._alloc(n, Kd) if need_dx else None
        self._emit('DENSE_BWD', [n.op(), K(N), K(Kd), self._p(dy), K(dy.ld), self._p(y) if pre else NULL, K(y.ld if pre else 0),
                                 self._p(ctx[0]) if norm else NULL, self._p(ctx[1]) if norm else NULL,
                                 self._p(sc) if norm else NULL, K(int(bool(relu))), self._p(x), K(x.ld), self._p(W), K(W.ld),
                                 self._p(dW), K(dW.ld), self._p(doff), self._p(dsc), self._p(dx), K(Kd)]
                   + self._drop_args(site, K(-1), Kd) + [self._p(g_tmp), K(self._ws_gemm_ptr), K(self.gemm_ws_floats), gidx])
        return dx

    def _spmm(self, b, B, out, d, cscale=NULL, add=None, add_rows=NULL, vals=None, gidx=NULL, beta=0.0):
        """ops.spmm on the CSR described at descriptor base b (vals: other values on the same pattern and plan -- the
        squared adjacency, the medg-weighted one; gidx: B rows read through an index; beta: out = A.B + beta out)"""
        self._emit('SPMM', [self._ip(b + 3), self._ip(b + 4), self._fp(b + 5) if vals is None else vals, self._n(b), self._n(b + 1), K(d),
                            self._p(B), K(B.ld), gidx, NULL, cscale, self._p(out), K(out.ld), K(_fbits(beta))]
                   + self._plan(b, d) + [self._p(add), K(add.ld if add is not None else 0), add_rows])

    # ---- --det_dropout (layers.DetDropoutFC, the aggregators on (mean, variance), Gaussian re-sampling) ----------------
    def _nelem(self, t):
        """element count rows x pitch of a dense activation without a pitch gap, as an operand"""
        if t.ld != t.cols:
            raise Unsupported("det-dropout element-wise op on a column block")
        return (t.rows.mul * t.ld, t.rows.slot, t.rows.add * t.ld)

    def _gemm(self, A, B, out, ta=False, tb=False, accumulate=False):
        """ops.gemm: out = op(A) @ op(B) (+ out)"""
        M, Kd = ((K(A.cols), A.rows.op()) if ta else (A.rows.op(), K(A.cols)))
        N = B.rows.op() if tb else K(B.cols)
        self._ws_need = max(self._ws_need, GEMM_WS_BOUND)
        self._emit('GEMM', [K(int(ta)), K(int(tb)), M, N, Kd, self._p(A), K(A.ld), self._p(B), K(B.ld), self._p(out), K(out.ld),
                            K(int(accumulate)), K(self._ws_gemm_ptr), K(self.gemm_ws_floats)])
        return out

    def _sq_vals(self, b, l_rows_cap, c=1.0):
        """tf.square of a minibatch CSR's values (layers._squared): a vector in the arena, sized by a bound the batch is
        checked against (fits())"""
        cap = max(int(l_rows_cap), 1)
        self._cap_checks.append((b + 2, cap))
        out = self._alloc_vec(cap)[0]
        self._emit('SQUARE', [self._fp(b + 5), self._n(b + 2), K(_fbits(c)), out])
        return out

    def _nnz_cap(self, l, which):
        """upper bound of the nonzeros of layer l's adj / adj^T (0 / 1: a sampled row has its own vertex + `degree`
        neighbours) or fadj (2: bounded by the graph's longest rows)"""
        deg = int(FLAGS.degree if self.model.is_training else FLAGS.test_degree)
        if which != 2:
            return self.caps[l + 1] * (deg + 1)
        full = self.model.__dict__.get('_max_full_degree')
        if full is None:
            a = getattr(self.model, 'adj', None)
            full = int(np.diff(a.indptr).max()) + 1 if (a is not None and hasattr(a, 'indptr')) else None
            self.model.__dict__['_max_full_degree'] = full
        if full is None:
            raise Unsupported("det-dropout: no bound on the full-neighbour matrix")
        return self.caps[l + 1] * full

    def _owner_words(self, hist):
        """one zeroed int32 per history row (sgcn_hist_apply_f32's claim table), shared by the program's exchanges"""
        ow = self.__dict__.get('_owner')
        if ow is None:
            # sized ONCE, for the tallest history of the model: HIST_APPLY ops bake the table's address in, so it must
            # never be replaced by a larger one while an earlier op still points at it
            rows = max(int(h.shape[0]) for hs in self.model.history for h in hs)
            ow = self._owner = torch.zeros(max(rows, int(hist.shape[0])), dtype=torch.int32, device=self.dev)
        if ow.numel() < int(hist.shape[0]):
            raise Unsupported("history exchange: a history taller than the claim table")
        return ow

    def _native_exchange(self, l, nh, aux=0):
        """Policy H-a (parallel.py) as ops of the program: [cap ids | cap x d row bits] per rank, all-gathered on the
        library's communicator, applied in rank order.
          aux = 0: on the step's own stream behind the optimizer, where the reference orders the update
                   (gcn/models.py:186-194) -- three ops on the step's dependent chain.
          aux = 2 (round 6, DataParallel.exchange_overlap): on the library's EXCHANGE stream, issued right behind the
                   aggregator that read the history -- its payload (the aggregator's input) is final there and its first
                   reader is the NEXT step's aggregator, so pack + all-gather + apply run beside the loss, the backward pass,
                   the gradient all-reduce and the optimizer; the step's stream waits for them at the end of the run.  The
                   all-gather has a communicator of its own (sgcn_coll_init_exchange).
          aux = 1: the round-5 form (auxiliary stream, ONE communicator for both collectives): measured slower and erratic
                   -- 0.177 against 0.144 ms per Reddit step with a one-rank communicator; a communicator that alternates
                   between two streams synchronises them itself (profiles/HISTORY.md round 5).  Kept for the record."""
        hist = self.model.history[l][0]
        # the capacity of a rank's block: the job-wide bound the layer-by-layer path uses too (DataParallel.history_cap,
        # >= every field's own bound) -- a rank whose minibatch did not fit its program exchanges blocks of the same size
        hc = getattr(getattr(self.model, '_par', None), 'history_cap', None)
        d, cap = int(nh.cols), (max(int(hc or 0), self.caps[l]) + 3) // 4 * 4
        send = self._alloc_vec(cap * (d + 1))[0]
        recv = self._alloc_vec(self.native_world * cap * (d + 1))[0]
        self._emit('HIST_PACK', [self._field_ptr(l), self.rows[l].op(), self._p(nh), K(nh.ld), K(d), K(cap), send, K(aux)])
        self._emit('ALLGATHER_I32', [send, recv, K(cap * (d + 1)), K(aux)])
        owner = self._owner_words(hist)
        self._emit('HIST_APPLY', [K(hist.data_ptr()), K(hist.stride(0)), recv, K(self.native_world), K(cap), K(d),
                                  K(owner.data_ptr()), K(aux)])

    def _sparse_dropout(self, xs, site):
        """ops.dropout on the slice's value vector (one row of nnz elements)"""
        out = self._alloc_vec(xs.cap)[0]
        self._emit('DROPOUT', [xs.vals, xs.nnz, K(1), xs.nnz] + self._drop_args(site, K(-1), 0)[:4] + [xs.nnz, out, xs.nnz])
        return out

    def _sparse_fwd(self, xs, vals, W, off, sc, relu, post):
        """ops.spmm(slice, W) [+ ops.ln_act_fwd]: returns (y, (xhat, rstd) or None)"""
        n, N = xs.rows, W.cols
        y0 = self._alloc(n, N)
        self._emit('SPMM', [xs.o_p, xs.o_c, vals, n.op(), K(xs.ncols), K(N), self._p(W), K(W.ld), NULL, NULL, NULL,
                            self._p(y0), K(N), K(_fbits(0.0)), K(0), NULL, K(0), NULL, K(0), K(0), NULL, K(0), NULL, K(0), K(0)])
        if not post:
            return y0, None
        norm = off is not None
        y = self._alloc(n, N)
        xhat = self._alloc(n, N) if norm else None
        rstd = self._alloc_vec(n.cap)[0] if norm else None
        self._emit('LN_ACT_FWD', [self._p(y0), K(N), self._p(off), self._p(sc), n.op(), K(N), K(_fbits(1e-9)), K(int(bool(relu))),
                                  self._p(y), K(N), self._p(xhat), self._p(rstd)])
        return y, ((xhat, rstd) if norm else None)

    def _sparse_bwd(self, g, rec):
        """layers.Dense.backward with sparse_inputs: LayerNorm / ReLU backward, then dW += slice^T . g through the slice's
        transpose index (built once per step) and the values gathered into its order"""
        _, lay, xs, vals, y, ctx, relu, post = rec
        n, N = g.rows, g.cols
        if post:
            norm = ctx is not None
            need = (int(lib.sgcn_ln_act_bwd_ws_floats(n.cap, N)) + 3) // 4 * 4 if norm else 0
            self._ws_need = max(self._ws_need, need + GEMM_WS_BOUND)
            dx = self._alloc(n, N)
            self._emit('LN_ACT_BWD', [self._p(g), K(g.ld), self._p(y), K(y.ld), self._p(ctx[0]) if norm else NULL,
                                      self._p(ctx[1]) if norm else NULL, self._p(self._param(lay, 'scale')) if norm else NULL,
                                      n.op(), K(N), K(int(bool(relu))), self._p(dx), K(N),
                                      self._p(self._param(lay, 'offset', True)) if norm else NULL,
                                      self._p(self._param(lay, 'scale', True)) if norm else NULL,
                                      K(self._ws_gemm_ptr), K(self.gemm_ws_floats)])
            g = dx
        if xs.tbox[0] is None:
            ws_ints = int(lib.sgcn_csr_transpose_ws_ints(xs.ncols, xs.cap))
            trp, trow, tsrc = self._alloc_vec(xs.ncols + 1)[0], self._alloc_vec(xs.cap)[0], self._alloc_vec(xs.cap)[0]
            ws = self._alloc_vec(max(ws_ints, 1))[0]
            self._emit('CSR_TRANSPOSE', [K(xs.ncols), xs.nnz, xs.o_c, xs.o_r, trp, trow, tsrc, ws, K(ws_ints)])
            xs.tbox[0] = (trp, trow, tsrc)
        trp, trow, tsrc = xs.tbox[0]
        tv = self._alloc_vec(xs.cap)[0]
        self._emit('GATHER_F32', [vals, tsrc, xs.nnz, tv])
        dW = self._param(lay, 'weights', True)
        self._emit('SPMM', [trp, trow, tv, K(xs.ncols), n.op(), K(N), self._p(g), K(g.ld), NULL, NULL, NULL,
                            self._p(dW), K(dW.ld), K(_fbits(1.0)), K(0), NULL, K(0), NULL, K(0), K(0), NULL, K(0), NULL, K(0), K(0)])
        return None

    # ---- parameters -------------------------------------------------------------------------------
    def _param(self, layer, name, grad=False):
        t = (layer.grads if grad else layer.vars).get(name)
        if t is None:
            return None
        rows, cols = int(t.shape[0]), int(t.shape[1])
        return ST(K(t.data_ptr()), Rows(0, -1, rows, rows), cols, cols)

    # ---- the walk ---------------------------------------------------------------------------------
    def _build(self):
        m = self.model
        F = m.features_dev
        if self.sparse:
            # upload()'s csr_slice, as the program's first op: row pointer by a device prefix pass, then the copy
            cap = max(self.nnz_cap, 1)
            o_p = self._alloc_vec(self.caps[0] + 1)[0]
            o_c, o_d, o_r = self._alloc_vec(cap)[0], self._alloc_vec(cap)[0], self._alloc_vec(cap)[0]
            act = SSparse(o_p, o_c, o_d, o_r, self.rows[0], self._nnz(), int(F.shape[1]), cap)
            self._emit('CSR_SLICE', [self.rows[0].op(), self._field_ptr(0), K(F.val.data_ptr()), K(F.col.data_ptr()),
                                     K(F.rowptr.data_ptr()), o_p, o_d, o_c, o_r])
        else:
            feat = ST(K(F.data_ptr()), Rows(0, -1, int(F.shape[0]), int(F.shape[0])), int(F.shape[1]), int(F.stride(0)))
            act = SGather(feat, self._field_ptr(0), self.rows[0])
        tape = []
        concat = FLAGS.normalization != 'gcn'
        self.new_history = {}
        # Work that depends on nothing this step computes goes to the auxiliary stream first, beside the
        # dense layers: zeroing the gradient buffer, and every control-variate aggregator's history-only sum
        # P . Hbar[ffield] (the dominant gather of the step) -- sgcn_vr_aggregate_pre_f32.
        # the library's auxiliary stream is used only on request: with the weight gradients grouped and the memset /
        # statistics / scatter gone from it (group_dw, lean_sync) the two events around the aggregator's history half cost
        # more than the overlap returns
        # per-program state (SGCN_OP_MODE at the head of every run), not the process-wide knob: another program built later
        # -- the test model's, another model's -- does not change how this one runs
        self.overlap = int(bool(FLAGS.agg_overlap) or not (FLAGS.lean_sync and FLAGS.group_dw))
        # data parallel with the library's own RCCL communicator (parallel.DataParallel.native): the gradient all-reduce
        # and the history exchange are ops of THIS program -- one foreign call per step, as on one GPU -- instead of Python
        # calls between its phases
        self.native_world = int(getattr(m, 'native_coll', 0) or 0) if m.is_training else 0
        # ... the exchange on the library's exchange stream, right behind the aggregator (needs the second communicator)
        self.exchange_overlap = bool(self.native_world and getattr(getattr(m, '_par', None), 'exchange_overlap', False))
        self._exchanged = set()
        local_hist = m.history_hook is None and not self.native_world
        # the history scatter: beside the step on the auxiliary stream (one event pair, a barrier on the compute queue), or
        # -- lean_sync -- on the step's own stream after the optimizer, where it costs its 4 us and no synchronisation
        self._hist_last = bool(FLAGS.lean_sync) and local_hist
        self.det = bool(FLAGS.det_dropout)
        if self.det and (self.native_world or m.history_hook is not None):
            raise Unsupported("det-dropout with a multi-GPU history exchange")
        if m.is_training:
            if self.sparse or self.det:
                # the sparse layer's gradients are sums INTO the buffer (a transposed product with beta = 1, LayerNorm
                # parameter sums): zeroed first, on the step's own stream
                self._emit('MEMSET0', [K(m.grad.data_ptr()), K(m.grad.numel() * 4)])
            elif FLAGS.lean_sync:
                # gradient-STORE mode (include/sgcn.h SGCN_OP_GRAD_STORE): every layer of a supported stack writes each of
                # its parameter gradients exactly once per step, so nothing is zeroed (no memset, no join on it) and the
                # loss statistics ride in the optimizer's launch
                self._emit('GRAD_STORE', [])
            else:
                self._emit('AUX_MEMSET0', [K(m.grad.data_ptr()), K(m.grad.numel() * 4)])
        accP = {}
        two_phase = bool(FLAGS.agg_overlap) and not self.det      # (a det-dropout stack keeps its plain-input aggregator fused)
        for layer in m.layers:
            if two_phase and isinstance(layer, VRAggregator):
                l = layer.l
                bf = self._csr(l, 2)
                hist = m.history[l][0]
                d = int(hist.shape[1])
                ldw = (d + 3) // 4 * 4
                buf = self._alloc(self.rows[l + 1], ldw)
                self._emit('VR_AGG_PRE', [self._ip(bf + 3), self._ip(bf + 4), self._fp(bf + 5), self.rows[l + 1].op(), self._n(bf + 1),
                                          K(d), K(hist.data_ptr()), K(int(hist.stride(0))), self._ip(self._pb.o_ffields + 2 * l),
                                          self._p(buf)] + self._plan(bf, d))
                accP[l] = buf
        for layer in m.layers:
            if isinstance(layer, (AugmentedDropoutDense, Dense)) and layer.sparse_inputs:
                # layers.Dense / AugmentedDropoutDense with sparse_inputs, call for call: (dropout of the slice's values) ->
                # sparse product(s) with W -> LayerNorm + ReLU pass(es)
                if not isinstance(act, SSparse):
                    raise Unsupported("sparse layer behind a dense activation")
                xs = act
                aug = isinstance(layer, AugmentedDropoutDense)
                site = self._site(layer) if aug else None
                W = self._param(layer, 'weights')
                off = self._param(layer, 'offset') if layer.norm else None
                sc = self._param(layer, 'scale') if layer.norm else None
                relu = True if aug else bool(layer.act)
                vals = xs.vals
                if site is not None:
                    vals = self._sparse_dropout(xs, site)
                hx, ctx = self._sparse_fwd(xs, vals, W, off, sc, relu, aug or layer.norm or layer.act)
                tape.append(('sdense', layer, xs, vals, hx, ctx, relu, aug or layer.norm or layer.act))
                if aug:
                    hmu = hx if site is None else self._sparse_fwd(xs, xs.vals, W, off, sc, True, True)[0]
                    act = (hx, hmu)
                else:
                    act = hx
            elif isinstance(layer, AugmentedDropoutDense):
                if layer.output_dim > 128:
                    raise Unsupported("wide AugmentedDropoutDense")
                x, mu = act if isinstance(act, tuple) else (act, act)
                site = self._site(layer)
                W, off, sc = self._param(layer, 'weights'), self._param(layer, 'offset') if layer.norm else None, \
                    self._param(layer, 'scale') if layer.norm else None
                same = mu is x
                x = self._materialize(x) if isinstance(x, SDropped) else x
                mu = x if same else (self._materialize(mu) if isinstance(mu, SDropped) else mu)
                if mu is x and site is None:
                    h, ctx = self._dense_fwd(x, W, off, sc, True)
                    tape.append(('dense', layer, x, h, ctx, None, True))
                    act = (h, h)
                else:
                    n = x.rows
                    h2, ctx2 = self._dense_fwd(x, W, off, sc, True, x2=mu, site=site)
                    ctx = (ctx2[0].head(n), ctx2[1]) if ctx2 is not None else None
                    hx = h2.head(n)
                    tape.append(('dense', layer, x, hx, ctx, site, True))
                    act = (hx, h2.tail(n))
            elif isinstance(layer, Dropout) and isinstance(act, SSparse):
                # layers.Dropout on the sparse slice: same structure, dropped values; nothing on the way back
                site = self._site(layer)
                if site is not None:
                    act = act.with_vals(self._sparse_dropout(act, site))
                tape.append(('dropout', None, False, None))
            elif isinstance(layer, Dropout):
                site = self._site(layer)
                inp = act[0] if (layer.cvd and isinstance(act, tuple)) else act
                gauss = None
                if isinstance(inp, SDet):
                    # layers.Dropout on (mu, var): x ~ N(mu, var + 1e-10) by the counter hash (key of layer index + 4096),
                    # then the ordinary dropout below
                    key = self._key(layer.index + 4096)
                    x = self._alloc(inp.mu.rows, inp.mu.cols)
                    self._emit('GAUSS', [self._p(inp.mu), self._p(inp.var), self._nelem(inp.mu), key, self._p(x)])
                    gauss = (inp.var, key)
                    inp = x
                if isinstance(inp, tuple):
                    raise Unsupported("dropout of a tuple")
                if not (layer.fuse_next and isinstance(inp, SGather)):
                    inp = self._materialize(inp)
                if site is None:
                    act = inp
                    tape.append(('dropout', None, False, gauss))
                elif layer.fuse_next:
                    act = SDropped(inp, site)
                    tape.append(('dropout', site, True, gauss))
                else:
                    act = self._materialize(SDropped(inp, site))
                    tape.append(('dropout', site, False, gauss))
            elif isinstance(layer, DetDropoutFC):
                # layers.DetDropoutFC.forward, call for call
                if isinstance(act, SDet):
                    mu, var = act.mu, act.var
                else:
                    mu, var = self._materialize(act), None
                    if isinstance(mu, (tuple, SDropped)):
                        raise Unsupported("DetDropoutFC input")
                keep = 1.0 - self.dropout
                W = self._param(layer, 'weights')
                n, Kin, N = mu.rows, mu.cols, W.cols
                var_in = self._alloc(n, Kin)
                self._emit('DET_PRE', [self._p(mu), self._p(var), self._nelem(mu), K(_fbits(keep)), self._p(var_in)])
                W2 = self._alloc(W.rows, N)
                self._emit('SQUARE', [self._p(W), K(W.rows.cap * N), K(_fbits(1.2)), self._p(W2)])
                mu1 = self._gemm(mu, W, self._alloc(n, N))
                var1 = self._gemm(var_in, W2, self._alloc(n, N))
                ctx = None
                if layer.norm:
                    off, sc = self._param(layer, 'offset'), self._param(layer, 'scale')
                    mu2, xhat, rstd = self._alloc(n, N), self._alloc(n, N), self._alloc_vec(n.cap)[0]
                    self._emit('LN_ACT_FWD', [self._p(mu1), K(mu1.ld), self._p(off), self._p(sc), n.op(), K(N), K(_fbits(layer.LN_EPS)),
                                              K(0), self._p(mu2), K(mu2.ld), self._p(xhat), rstd])
                    var2 = self._alloc(n, N)
                    self._emit('DET_LNVAR_FWD', [self._p(var1), rstd, self._p(sc), n.op(), K(N), K(_fbits(layer.LN_EPS)), self._p(var2)])
                    ctx = (xhat, rstd)
                else:
                    mu2, var2 = mu1, var1
                mo, vo = self._alloc(n, N), self._alloc(n, N)
                self._emit('DET_RELU_FWD', [self._p(mu2), self._p(var2), self._nelem(mu2), self._p(mo), self._p(vo)])
                tape.append(('detfc', layer, mu, var is not None, var_in, W2, var1, mu2, var2, ctx, keep))
                act = SDet(mo, vo)
            elif isinstance(layer, Dense):
                x, site = act, None
                if isinstance(x, SDropped):
                    x, site = x.x, x.site
                if not (layer.output_dim <= 128 or not (layer.norm or layer.act)):
                    raise Unsupported("wide Dense with LayerNorm / ReLU")
                W = self._param(layer, 'weights')
                off = self._param(layer, 'offset') if layer.norm else None
                sc = self._param(layer, 'scale') if layer.norm else None
                y, ctx = self._dense_fwd(x, W, off, sc, layer.act, site=site)
                tape.append(('dense', layer, x, y, ctx, site, layer.act))
                act = y
            elif isinstance(layer, VRAggregator):
                l = layer.l
                ba, bf = self._csr(l, 0), self._csr(l, 2)
                hist = m.history[l][0]
                H = ST(K(hist.data_ptr()), Rows(0, -1, int(hist.shape[0]), int(hist.shape[0])), int(hist.shape[1]), int(hist.stride(0)))
                n1 = self.rows[l + 1]
                if isinstance(act, SDet):
                    # layers.VRAggregator._forward_det, call for call: seven products on the general kernel around two
                    # element-wise launches; the squared matrices and the medg-weighted one share the sampled matrices'
                    # patterns and plans
                    if layer.cvd or len(m.history[l]) < 2:
                        raise Unsupported("det-dropout aggregator without a mean and a variance history")
                    mu, var = act.mu, act.var
                    Hm, Hv = m.history[l][0], m.history[l][1]
                    d = mu.cols
                    if d != int(Hm.shape[1]) or int(Hm.stride(0)) != int(Hv.stride(0)):
                        raise Unsupported("det-dropout aggregator: history shape")
                    n0 = self.rows[l]
                    A2 = self._sq_vals(ba, self._nnz_cap(l, 0))
                    P2 = self._sq_vals(bf, self._nnz_cap(l, 2))
                    Mv = self._fp(self._pb.o_medg + 2 * l)
                    ffield = self._ip(self._pb.o_ffields + 2 * l)
                    dmu, ds2, msig2, ds, sbar = (self._alloc(n0, d) for _ in range(5))
                    self._emit('DET_AGG_PREP', [self._p(mu), self._p(var), K(Hm.data_ptr()), K(Hv.data_ptr()), K(int(Hm.stride(0))),
                                                self._field_ptr(l), n0.op(), K(d), self._p(dmu), self._p(ds2), self._p(msig2),
                                                self._p(ds), self._p(sbar)])
                    w = 2 * d if concat else d
                    om, ov = self._alloc(n1, w), self._alloc(n1, w)
                    nb = om.cols_from(d, d) if concat else om
                    HmT = ST(K(Hm.data_ptr()), Rows(0, -1, int(Hm.shape[0]), int(Hm.shape[0])), d, int(Hm.stride(0)))
                    HvT = ST(K(Hv.data_ptr()), Rows(0, -1, int(Hv.shape[0]), int(Hv.shape[0])), d, int(Hv.stride(0)))
                    self._spmm(ba, dmu, nb, d)
                    self._spmm(bf, HmT, nb, d, gidx=ffield, beta=1.0)
                    raw = self._alloc(n1, d)
                    self._spmm(ba, ds2, raw, d, vals=A2)
                    self._spmm(bf, HvT, raw, d, vals=P2, gidx=ffield, beta=1.0)
                    self._spmm(ba, msig2, raw, d, vals=Mv, beta=1.0)
                    ovn = ov.cols_from(d, d) if concat else ov
                    self._emit('RELU_EPS', [self._p(raw), K(raw.ld), n1.op(), K(d), K(_fbits(1e-10)), self._p(ovn), K(ovn.ld)])
                    if concat:
                        self._emit('COPY2D', [self._p(om), K(om.ld), self._p(mu), K(mu.ld), n1.op(), K(d)])
                        self._emit('COPY2D', [self._p(ov), K(ov.ld), self._p(var), K(var.ld), n1.op(), K(d)])
                    if local_hist and not self._hist_last:
                        for hist_t, src in ((Hm, mu), (Hv, var)):
                            self._emit('AUX_SCATTER_ROWS', [K(hist_t.data_ptr()), K(hist_t.stride(0)), self._field_ptr(l), n0.op(),
                                                            K(src.cols), self._p(src), K(src.ld)])
                    self.new_history[l] = [mu, var]
                    tape.append(('vagg_det', l, d, concat, var, ds, sbar, raw))
                    act = SDet(om, ov)
                elif layer.cvd:
                    h, mu = (self._materialize(t) for t in act)
                    d = h.cols
                    if mu.ld != h.ld:
                        raise Unsupported("h / mu pitch")
                    width = 2 * d if concat else d
                    out_h, out_mu = self._alloc(n1, width), self._alloc(n1, width)
                    sptr = self._fp(self._pb.o_scales + 2 * l)
                    if d != int(hist.shape[1]):
                        raise Unsupported("aggregator width differs from its history")
                    if two_phase:
                        self._emit('VR_AGG_POST', [self._ip(ba + 3), self._ip(ba + 4), self._fp(ba + 5), n1.op(), self.rows[l].op(), K(d),
                                                   self._p(h), self._p(mu), K(h.ld), self._p(H), K(H.ld), self._field_ptr(l), sptr,
                                                   self._p(out_h), self._p(out_mu), K(width), K(1), K(int(concat)), self._p(accP[l])])
                    else:
                        self._emit('VR_AGG', [self._ip(ba + 3), self._ip(ba + 4), self._fp(ba + 5), self._ip(bf + 3), self._ip(bf + 4),
                                              self._fp(bf + 5), n1.op(), self.rows[l].op(), self._n(bf + 1), K(d), self._p(h), self._p(mu),
                                              K(h.ld), self._p(H), K(H.ld), self._field_ptr(l), self._ip(self._pb.o_ffields + 2 * l), sptr,
                                              self._p(out_h), self._p(out_mu), K(width), K(1), K(int(concat))] + self._plan(bf, d))
                    if local_hist and not self._hist_last:
                        self._emit('AUX_SCATTER_ROWS', [K(hist.data_ptr()), K(hist.stride(0)), self._field_ptr(l), self.rows[l].op(),
                                                        K(mu.cols), self._p(mu), K(mu.ld)])
                    self.new_history[l] = mu
                    if self.exchange_overlap:
                        self._native_exchange(l, mu, aux=2)
                        self._exchanged.add(l)
                    tape.append(('agg', l, d, concat, sptr))
                    act = (out_h, out_mu)
                else:
                    x = self._materialize(act)
                    d = x.cols
                    width = 2 * d if concat else d
                    out_h = self._alloc(n1, width)
                    if d != int(hist.shape[1]):
                        raise Unsupported("aggregator width differs from its history")
                    if two_phase:
                        self._emit('VR_AGG_POST', [self._ip(ba + 3), self._ip(ba + 4), self._fp(ba + 5), n1.op(), self.rows[l].op(), K(d),
                                                   self._p(x), NULL, K(x.ld), self._p(H), K(H.ld), self._field_ptr(l), NULL,
                                                   self._p(out_h), NULL, K(width), K(0), K(int(concat)), self._p(accP[l])])
                    else:
                        self._emit('VR_AGG', [self._ip(ba + 3), self._ip(ba + 4), self._fp(ba + 5), self._ip(bf + 3), self._ip(bf + 4),
                                              self._fp(bf + 5), n1.op(), self.rows[l].op(), self._n(bf + 1), K(d), self._p(x), NULL,
                                              K(x.ld), self._p(H), K(H.ld), self._field_ptr(l), self._ip(self._pb.o_ffields + 2 * l), NULL,
                                              self._p(out_h), NULL, K(width), K(0), K(int(concat))] + self._plan(bf, d))
                    if local_hist and not self._hist_last:
                        self._emit('AUX_SCATTER_ROWS', [K(hist.data_ptr()), K(hist.stride(0)), self._field_ptr(l), self.rows[l].op(),
                                                        K(x.cols), self._p(x), K(x.ld)])
                    self.new_history[l] = x
                    if self.exchange_overlap:
                        self._native_exchange(l, x, aux=2)
                        self._exchanged.add(l)
                    tape.append(('agg', l, d, concat, NULL))
                    act = out_h
            elif isinstance(layer, PlainAggregator) and isinstance(act, SDet):
                # layers.PlainAggregator on (mu, var): A mu and A^2 var (gcn/layers.py:236-248)
                l = layer.l
                mu, var = act.mu, act.var
                d, n1, ba = mu.cols, self.rows[l + 1], self._csr(l, 0)
                A2 = self._sq_vals(ba, self._nnz_cap(l, 0))
                if not concat:
                    om, ov = self._alloc(n1, d), self._alloc(n1, d)
                    self._spmm(ba, mu, om, d)
                    self._spmm(ba, var, ov, d, vals=A2)
                else:
                    om, ov = self._alloc(n1, 2 * d), self._alloc(n1, 2 * d)
                    self._emit('COPY2D', [self._p(om), K(om.ld), self._p(mu), K(mu.ld), n1.op(), K(d)])
                    self._emit('COPY2D', [self._p(ov), K(ov.ld), self._p(var), K(var.ld), n1.op(), K(d)])
                    self._spmm(ba, mu, om.cols_from(d, d), d)
                    self._spmm(ba, var, ov.cols_from(d, d), d, vals=A2)
                tape.append(('pagg_det', l, d, concat))
                act = SDet(om, ov)
            elif isinstance(layer, PlainAggregator):
                l = layer.l
                x = self._materialize(act)
                d = x.cols
                n1 = self.rows[l + 1]
                ba = self._csr(l, 0)
                if not concat:
                    out = self._alloc(n1, d)
                    self._spmm(ba, x, out, d)
                else:
                    out = self._alloc(n1, 2 * d)
                    self._emit('COPY2D', [self._p(out), K(out.ld), self._p(x), K(x.ld), n1.op(), K(d)])
                    self._spmm(ba, x, out.cols_from(d, d), d)
                tape.append(('agg', l, d, concat, NULL))
                act = out
            else:
                raise Unsupported("layer %r" % type(layer).__name__)
        logits = act
        if isinstance(logits, (tuple, SDropped, SGather, SDet)):
            raise Unsupported("model output is not a plain activation")
        nL = self.rows[self.L]
        c = logits.cols
        lab_b = self._pb.o_labels
        train = m.is_training
        stats, self.stats_off = self._alloc_vec(4 + (2 if train else 3) * nL.cap)     # (evaluation: + the rows' classes)
        dz = self._alloc(nL, c) if train else None
        self.pred = self._alloc(nL, c) if not train else None
        rowstat = K(stats[2] + 16)
        self._emit('SIGMOID_CE' if m.multitask else 'SOFTMAX_CE',
                   [self._p(logits), K(logits.ld), self._fp(lab_b), K(c), nL.op(), K(c), self._p(dz), K(c), self._p(self.pred), K(c),
                    stats, rowstat])
        wd = float(FLAGS.weight_decay)
        lo, hi = m._wd_range
        if wd and hi > lo:
            self._emit('L2_PENALTY', [K(m.theta.data_ptr()), K(lo), K(hi), K(_fbits(wd)), NULL, K(stats[2] + 8)])
        if train:
            g = dz
            first = m._first_param
            for layer, rec in reversed(list(zip(m.layers, tape))[first:]):
                if rec[0] == 'sdense':
                    g = self._sparse_bwd(g, rec)
                elif rec[0] == 'dense':
                    _, lay, x, y, ctx, site, relu = rec
                    g = self._dense_bwd(g, y, ctx, self._param(lay, 'scale') if lay.norm else None, relu, x,
                                        self._param(lay, 'weights'), self._param(lay, 'weights', True),
                                        self._param(lay, 'offset', True) if lay.norm else None,
                                        self._param(lay, 'scale', True) if lay.norm else None, lay.need_dx, site)
                elif rec[0] == 'dropout':
                    _, site, fused, gauss = rec
                    if g is not None and site is not None and not fused:
                        out = self._alloc(g.rows, g.cols)
                        self._emit('DROPOUT', [self._p(g), K(g.ld), g.rows.op(), K(g.cols)] + self._drop_args(site, K(-1), g.cols)
                                   + [self._p(out), K(out.ld)])
                        g = out
                    if gauss is not None and g is not None:          # d/d(mu, var) of the Gaussian draw: (g, g eps / (2 sigma))
                        var, key = gauss
                        d_var = self._alloc(g.rows, g.cols)
                        self._emit('GAUSS_BWD', [self._p(var), self._p(g), self._nelem(g), key, self._p(d_var)])
                        g = (g, d_var)
                elif rec[0] == 'detfc':
                    # layers.DetDropoutFC.backward, call for call
                    _, lay, mu, had_var, var_in, W2, var1, mu2, var2, ctx, keep = rec
                    g0, g1 = g
                    n, N, Kin = mu2.rows, mu2.cols, mu.cols
                    W, dW = self._param(lay, 'weights'), self._param(lay, 'weights', True)
                    g_mu, g_var = self._alloc(n, N), self._alloc(n, N)
                    self._emit('DET_RELU_BWD', [self._p(mu2), self._p(var2), self._p(g0), self._p(g1), self._nelem(mu2),
                                                self._p(g_mu), self._p(g_var)])
                    if g0.ld != g0.cols or g1.ld != g1.cols:
                        raise Unsupported("det-dropout gradient with a pitch")
                    if lay.norm:
                        sc = self._param(lay, 'scale')
                        self._ws_need = max(self._ws_need, (int(lib.sgcn_ln_act_bwd_ws_floats(n.cap, N)) + 3) // 4 * 4 + GEMM_WS_BOUND)
                        g_mu1 = self._alloc(n, N)
                        self._emit('LN_ACT_BWD', [self._p(g_mu), K(g_mu.ld), self._p(mu2), K(mu2.ld), self._p(ctx[0]), ctx[1], self._p(sc),
                                                  n.op(), K(N), K(0), self._p(g_mu1), K(g_mu1.ld),
                                                  self._p(self._param(lay, 'offset', True)), self._p(self._param(lay, 'scale', True)),
                                                  K(self._ws_gemm_ptr), K(self.gemm_ws_floats)])
                        d_var1, tmp = self._alloc(n, N), self._alloc(n, N)
                        self._emit('DET_LNVAR_BWD', [self._p(g_var), self._p(var1), self._p(ctx[0]), ctx[1], self._p(sc), n.op(), K(N),
                                                     K(_fbits(lay.LN_EPS)), self._p(d_var1), self._p(g_mu1),
                                                     self._p(self._param(lay, 'scale', True)), self._p(tmp)])
                        g_mu, g_var = g_mu1, d_var1
                    # dW = mu^T d_mu1 + 2.4 W (.) (var_in^T d_var1)
                    self._gemm(mu, g_mu, dW, ta=True, accumulate=True)
                    t = self._gemm(var_in, g_var, self._alloc(W.rows, N), ta=True)
                    self._emit('ADDMUL', [self._p(dW), self._p(W), self._p(t), K(W.rows.cap * N), K(_fbits(2.4))])
                    if not lay.need_dx:
                        g = None
                    else:
                        d_mu = self._gemm(g_mu, W, self._alloc(n, Kin), tb=True)
                        t2 = self._gemm(g_var, W2, self._alloc(n, Kin), tb=True)
                        d_var = self._alloc(n, Kin) if had_var else None
                        self._emit('DET_PRE_BWD', [self._p(mu), self._p(t2), self._nelem(mu), K(_fbits(keep)), self._p(d_mu), self._p(d_var)])
                        g = (d_mu, d_var) if had_var else d_mu
                elif rec[0] == 'pagg_det':
                    _, l, d, cc = rec
                    bt, n1 = self._csr(l, 1), self.rows[l + 1]
                    A2t = self._sq_vals(bt, self._nnz_cap(l, 1))
                    dm, dv = self._alloc(self.rows[l], d), self._alloc(self.rows[l], d)
                    if not cc:
                        self._spmm(bt, g[0], dm, d)
                        self._spmm(bt, g[1], dv, d, vals=A2t)
                    else:
                        self._spmm(bt, g[0].cols_from(d, d), dm, d, add=g[0].cols_from(0, d), add_rows=n1.op())
                        self._spmm(bt, g[1].cols_from(d, d), dv, d, vals=A2t, add=g[1].cols_from(0, d), add_rows=n1.op())
                    g = (dm, dv)
                elif rec[0] == 'vagg_det':
                    # layers.VRAggregator._backward_det
                    _, l, d, cc, var, ds, sbar, raw = rec
                    bt, n1, n0 = self._csr(l, 1), self.rows[l + 1], self.rows[l]
                    gm, gv = (g[0].cols_from(d, d), g[1].cols_from(d, d)) if cc else g
                    gvg = self._alloc(n1, d)
                    self._emit('GATE', [self._p(raw), K(raw.ld), self._p(gv), K(gv.ld), n1.op(), K(d), self._p(gvg)])
                    d_mu = self._alloc(n0, d)
                    if cc:
                        self._spmm(bt, gm, d_mu, d, add=g[0].cols_from(0, d), add_rows=n1.op())
                    else:
                        self._spmm(bt, gm, d_mu, d)
                    A2t = self._sq_vals(bt, self._nnz_cap(l, 1))
                    g_ds2, g_msig2 = self._alloc(n0, d), self._alloc(n0, d)
                    self._spmm(bt, gvg, g_ds2, d, vals=A2t)
                    self._spmm(bt, gvg, g_msig2, d, vals=self._fp(self._pb.o_tmedg + 2 * l))
                    d_var = self._alloc(n0, d)
                    addv = g[1].cols_from(0, d) if cc else None
                    self._emit('DET_AGG_PREP_BWD', [self._p(var), self._p(ds), self._p(sbar), self._p(g_ds2), self._p(g_msig2), n0.op(), K(d),
                                                    self._p(addv), K(addv.ld if addv is not None else 0), n1.op() if cc else K(0),
                                                    self._p(d_var)])
                    g = (d_mu, d_var)
                elif rec[0] == 'agg':
                    _, l, d, cc, sptr = rec
                    bt = self._csr(l, 1)
                    dx = self._alloc(self.rows[l], d)
                    if not cc:
                        self._spmm(bt, g, dx, d, cscale=sptr)
                    else:
                        self._spmm(bt, g.cols_from(d, d), dx, d, cscale=sptr, add=g.cols_from(0, d), add_rows=self.rows[l + 1].op())
                    g = dx
            if FLAGS.group_dw:
                # deferred weight-gradient mode (include/sgcn.h SGCN_OP_DW_FLUSH): the DENSE_BWD ops above only recorded their
                # dW GEMMs; all of them + their reductions run here as two launches
                self._emit('DW_FLUSH', [])
            if wd and hi > lo:
                self._emit('L2_PENALTY', [K(m.theta.data_ptr()), K(lo), K(hi), K(_fbits(wd)), K(m.grad.data_ptr()), NULL])
            if self.native_world:
                self._emit('ALLREDUCE_AVG', [K(m.grad.data_ptr()), K(m.grad.numel())])
            self._cur = self.ops_opt
            self._emit('ADAM', [K(m.theta.data_ptr()), K(m.grad.data_ptr()), K(m.adam_m.data_ptr()), K(m.adam_v.data_ptr()),
                                K(m.theta.numel()), self._lr(), K(_fbits(FLAGS.beta1)), K(_fbits(FLAGS.beta2)), K(_fbits(1e-8))])
        self._cur = self.ops_hist
        for l, nh in ({} if (local_hist and not self._hist_last) else self.new_history).items():
            if self.native_world:
                if l not in self._exchanged:
                    self._native_exchange(l, nh)
                continue
            # (a det-dropout aggregator updates a mean AND a variance history: models.update_history's zip)
            for hist, src in zip(m.history[l], nh if isinstance(nh, list) else [nh]):
                self._emit('SCATTER_ROWS', [K(hist.data_ptr()), K(hist.stride(0)), self._field_ptr(l), self.rows[l].op(),
                                            K(src.cols), self._p(src), K(src.ld)])

    # ---- build the ctypes program ------------------------------------------------------------------
    def _finalize(self):
        self._key_slots = {li: self._n_meta + i for i, li in enumerate(self._key_layers)}
        self.lr_slot = self._n_meta + len(self._key_layers)
        self.nnz_slot = self.lr_slot + 1
        self.nslots = self.lr_slot + (2 if self.sparse else 1)

        def pack(lst):
            if lst:
                lst = [(OP['MODE'], [K(self.overlap), K(-1)])] + list(lst)
            arr = (StepOp * max(len(lst), 1))()
            for k, (opc, args) in enumerate(lst):
                o = arr[k]
                o.op, o.nargs = opc, len(args)
                for j, a in enumerate(args):
                    o.mul[j], o.slot[j], o.add[j] = int(a[0]), int(a[1]), int(a[2])
            return arr
        self.c_fb, self.c_opt, self.c_hist = pack(self.ops_fb), pack(self.ops_opt), pack(self.ops_hist)
        self.n_fb, self.n_opt, self.n_hist = (len(x) + (1 if x else 0) for x in (self.ops_fb, self.ops_opt, self.ops_hist))
        allops = self.ops_fb + self.ops_opt + self.ops_hist       # one contiguous program for the single-GPU case
        self.c_all, self.n_all = pack(allops), len(allops) + (1 if allops else 0)
        f = np.array(self._fill, dtype=np.int64).reshape(-1, 3)
        self._f_idx, self._f_mul = f[:, 0].copy(), f[:, 1].copy()
        self._f_ip, self._f_fp = (f[:, 2] == 1).astype(np.int64), (f[:, 2] == 2).astype(np.int64)
        self.slots = np.zeros(self.nslots, dtype=np.int64)
        self._slots_ptr = self.slots.ctypes.data
        chk = np.array(self._cap_checks, dtype=np.int64).reshape(-1, 2)
        self._cap_idx, self._cap_max = chk[:, 0].copy(), chk[:, 1].copy()
        ns = np.array(self._nslot_checks, dtype=np.int64).reshape(-1, 2) if self._nslot_checks else np.zeros((0, 2), np.int64)
        self._ns_idx, self._ns_ldw = ns[:, 0].copy(), ns[:, 1].copy()
        so = self.stats_off
        self.loss_t, self.acc_t = self.arena[so + 2], self.arena[so + 3]
        self._keys = sorted(self._key_slots.items())
        # the same tables for sgcn_step_fill (the arrays above stay the owners: the struct holds their addresses)
        self._f_base = f[:, 2].copy()
        self._key_slot = np.array([sl for _, sl in self._keys], dtype=np.int64)
        self._key_layer = np.array([li for li, _ in self._keys], dtype=np.int64)
        self._c_fill = StepFill(
            n=self._f_idx.shape[0], idx=self._f_idx.ctypes.data, mul=self._f_mul.ctypes.data, base=self._f_base.ctypes.data,
            n_cap=self._cap_idx.shape[0], cap_idx=self._cap_idx.ctypes.data, cap_max=self._cap_max.ctypes.data,
            n_ws=self._ns_idx.shape[0], ws_idx=self._ns_idx.ctypes.data, ws_ld=self._ns_ldw.ctypes.data,
            ws_floats=int(self.plan_ws_floats), n_keys=len(self._keys), key_slot=self._key_slot.ctypes.data,
            key_layer=self._key_layer.ctypes.data, lr_slot=self.lr_slot)
        self._c_fill_ref = C.byref(self._c_fill)
        self._runs = {'all': (self.c_all, self.n_all), 'fb': (self.c_fb, self.n_fb), 'opt': (self.c_opt, self.n_opt),
                      'hist': (self.c_hist, self.n_hist)}

    # ---- per step ---------------------------------------------------------------------------------
    def fits(self, pb):
        meta = pb.meta
        if np.any(meta[self._cap_idx] > self._cap_max):
            return False
        if self._ns_idx.size and np.any(meta[self._ns_idx] * self._ns_ldw > self.plan_ws_floats):
            return False
        return True

    def fill(self, pb, ip, fp, step, lr_t):
        """The minibatch's slot table (sgcn_step_fill: capacity checks, sizes and addresses, the step's dropout keys, the
        Adam step size); False when the minibatch does not fit the program."""
        rc = lib.sgcn_step_fill(self._c_fill_ref, pb.meta_ptr, pb.meta.shape[0], ip, fp, self.model.dropout_seed, step, lr_t,
                                self._slots_ptr, self.nslots)
        if rc < 0:
            check(rc)
        if rc == 0 and self.sparse:
            m = pb.m
            nnz = int(self._rowlen[pb.ibuf[m[4]:m[4] + m[5]]].sum())
            if nnz > self.nnz_cap:
                return False
            self.slots[self.nnz_slot] = nnz
        return rc == 0

    def run(self, which, stream):
        arr, n = self._runs[which]
        if n:
            rc = lib.sgcn_step_run(arr, n, self._slots_ptr, self.nslots, stream)
            if rc != 0 and self.native_world:
                # a step that failed on THIS rank may have stopped ahead of a collective op its peers are in: abort the
                # library's communicator so that they fail too instead of blocking for good (no watchdog on it)
                par = getattr(self.model, '_par', None)
                if par is not None:
                    par.abort()
                else:
                    lib.sgcn_coll_abort()
            check(rc)

    def tensor_of(self, t, n):
        """torch view of the first n rows of an arena activation under the CURRENT slot table (multi-GPU history
        exchange)"""
        mul, slot, add = t.ptr
        addr = add + (mul * int(self.slots[slot]) if slot >= 0 else 0)
        off = (addr - self.arena.data_ptr()) // 4
        return self.arena[off:off + n * t.ld].view(n, t.ld)[:, :t.cols]
