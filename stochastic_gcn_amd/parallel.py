"""Multi-GPU data parallelism for the hot path: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the box; "gloo" in the CPU tests).

The reference is single-process (one tf.Session, gcn/train.py:130); this module is new
 and follows BASELINE.json's north star:

  * vertex-range sharding: rank g owns vertices [g*N/W, (g+1)*N/W) -- its slice of the train
    ids and one sampler instance seeded ``seed + g`` (rank 0 of a 1-rank job reproduces the
    reference's exact sample sequence);
  * features, the CSR and the history are replicated per GPU (Reddit: 1.12 GB + 0.2 GB + 119 MB
    of 288 GB), so every gather of the step stays local;
  * ONE collective on the gradient path: all-reduce(sum) of the flat fp32 gradient buffer
    (~0.21 M floats for Reddit = 0.84 MB: latency-bound on xGMI, so a single flat call rather
    than per-tensor buckets), divided by W to keep the mean-loss semantics of
    gcn/models.py:82-83;
  * history consistency (policy H-a): after the optimizer step the ranks all-gather their
    ``(fields[l], new_history rows)`` and every replica applies all W updates in rank order
    (deterministic; a vertex updated by two ranks in one step keeps the higher rank's row).
    The all-gather is ASYNCHRONOUS: it is issued behind the optimizer and joined in front of the
    history's next reader -- the next step's aggregator (``join_history``) -- so it rides beside the
    host's work on the next batch instead of on the step's dependent chain.

On RCCL the two collectives of a step run on the LIBRARY's own communicator (``native``: ``sgcn_coll_*``, created from an
id that rank 0 draws and the process group broadcasts) as ops of the compiled step program -- one foreign call per step,
as on one GPU.  As torch.distributed calls between the program's phases they cost the launching thread ~80 us per step
(0.130 -> 0.211 ms per Reddit step with a one-rank group, host-bound: profiles/r44_epoch_fixed_cost.jsonl); the
torch path stays for gloo (the CPU tests) and behind ``SGCN_NATIVE_COLL=0``.

``SGCN_FORCE_PG=1`` makes a one-rank job take the collective paths as well (a real process group
of one rank): the RCCL smoke test, and ``bench.py --gpus 1`` printing its all-reduce time.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class DataParallel(object):
    HISTORY_FIXED_LIMIT_BYTES = 16 << 20     # per-rank payload above which the bound is too loose

    def __init__(self, backend=None, device=None, init=True):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.history_cap = None          # upper bound of |fields[l]| per step (rows), or None
        self._hist_bufs = {}
        self._pending = []               # history all-gathers in flight: (work, recv, history, cap, d, scatter_fn)
        self.force = os.environ.get("SGCN_FORCE_PG", "0") not in ("", "0")
        self.backend = None
        if (self.world > 1 or self.force) and init and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            backend = backend or os.environ.get("SGCN_DIST_BACKEND") or \
                ("nccl" if (device is not None and device.type == "cuda") else "gloo")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
        if dist.is_initialized():
            self.backend = dist.get_backend()
        self.native = False              # the library's own RCCL communicator carries the step's collectives
        self._native_owner = False
        self.exchange_overlap = False    # ... and a second one the history exchange, on the library's exchange stream
        if self.active and self.backend == "nccl" and os.environ.get("SGCN_NATIVE_COLL", "1") != "0" and \
                device is not None and device.type == "cuda":
            from ._ffi import lib
            if lib.sgcn_coll_world() == self.world:       # one communicator per process: a later object of the job shares it
                self.native = lib.sgcn_coll_retain() == 0  # (by reference count: the last shutdown() destroys it)
                self.exchange_overlap = self.native and bool(lib.sgcn_coll_has_exchange()) and self._want_exchange_overlap()
            elif init:
                self._init_native()
        self.native_history = self.native                  # the history exchange too (set_history_cap may say no)

    def _init_native(self):
        """The library's communicator: every rank probes that it can load RCCL (a rank that cannot would leave the others
        hanging in the collective initialisation), rank 0's id travels over the process group, every rank initialises.  Any
        failure, on any rank, leaves ALL ranks on the torch.distributed path."""
        import sys
        from ._ffi import lib
        dev = self.device
        ident = torch.zeros(128, dtype=torch.uint8)
        # every rank PROBES (library, symbols, NCCL >= 2.10); only rank 0 draws an id -- ncclGetUniqueId leaves a bootstrap
        # listener thread and socket behind for the life of the process
        ok = 1.0 if lib.sgcn_coll_available(None) == 0 else 0.0
        if ok and self.rank == 0:
            ok = 1.0 if lib.sgcn_coll_unique_id(ident.data_ptr()) == 0 else 0.0
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            if self.rank == 0:
                print("stochastic_gcn_amd: RCCL (>= 2.10) is not loadable from libsgcn.so on every rank; collectives "
                      "through torch.distributed", file=sys.stderr)
            return
        t = ident.to(dev)
        dist.broadcast(t, src=0)
        ident = t.cpu().contiguous()
        if dev is not None and dev.type == "cuda":
            torch.cuda.set_device(dev)
        rc = lib.sgcn_coll_init(ident.data_ptr(), self.world, self.rank)
        flag = torch.tensor([1.0 if rc == 0 else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            lib.sgcn_coll_destroy()
            if self.rank == 0:
                print("stochastic_gcn_amd: the library's RCCL communicator did not come up on every rank; collectives "
                      "through torch.distributed", file=sys.stderr)
            return
        self.native = self._native_owner = True
        # A second communicator for the history exchange, which the step program issues on the library's exchange stream
        # beside the backward pass and the gradient all-reduce (step_program._native_exchange): one communicator used from
        # two streams in turn makes RCCL order the streams itself (round 5: 0.177 ms per step, erratic).  Optional -- a job
        # where it does not come up on every rank keeps the exchange behind the optimizer on the step's own stream.
        if self._want_exchange_overlap():
            buf = torch.zeros(129, dtype=torch.uint8)            # [id (128 bytes) | rank 0 drew it]
            if self.rank == 0 and lib.sgcn_coll_unique_id(buf.data_ptr()) == 0:
                buf[128] = 1
            t = buf.to(dev)
            dist.broadcast(t, src=0)
            buf = t.cpu().contiguous()
            if int(buf[128]) == 1:
                ident2 = buf[:128].contiguous()
                rc = lib.sgcn_coll_init_exchange(ident2.data_ptr())
                flag = torch.tensor([1.0 if rc == 0 else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                # (a rank whose second communicator failed while others' came up: those keep theirs unused -- the
                # decision below is job-wide, and sgcn_coll_destroy takes both down)
                self.exchange_overlap = float(flag.item()) >= 1.0

    def _want_exchange_overlap(self):
        """SGCN_EXCHANGE_OVERLAP = 1 / 0, default: with two ranks or more.  With ONE rank (SGCN_FORCE_PG) the all-gather is a
        4.6 us local copy and taking the exchange off the chain does not pay: 128.2 us per step in order, 132 - 135 beside
        the step (the record on the step's stream costs 8 us, the wait at the end 3, two active queues ~8 more:
        profiles/r63_exchange_chain_probe.jsonl); a real all-gather among eight ranks is worth more than those ~7 us."""
        v = os.environ.get("SGCN_EXCHANGE_OVERLAP", "auto")
        return v == "1" or (v not in ("0", "1") and self.world >= 2)

    @property
    def active(self):
        return self.world > 1 or (self.force and dist.is_initialized())

    # ---- sharding ---------------------------------------------------------------------------
    def vertex_range(self, n):
        lo = (n * self.rank) // self.world
        hi = (n * (self.rank + 1)) // self.world
        return lo, hi

    def shard_ids(self, ids, n):
        """The ids of this rank's vertex range, order preserved."""
        lo, hi = self.vertex_range(n)
        ids = np.asarray(ids)
        return ids[(ids >= lo) & (ids < hi)]

    def sampler_seed(self, seed):
        return int(seed) + self.rank

    # ---- collectives ------------------------------------------------------------------------
    def allreduce_mean_(self, flat):
        """Mean over the ranks, in place.  RCCL averages inside the collective (ReduceOp.AVG: no extra kernel on the
        step's dependent chain); gloo has no AVG: sum, then one division."""
        if self.active:
            if self.native:               # (a step that runs layer by layer: the same communicator as the programs' ops)
                from ._ffi import check, lib
                check(lib.sgcn_coll_allreduce_avg_f32(flat.data_ptr(), flat.numel(), torch.cuda.current_stream().cuda_stream))
            elif self.backend == "nccl":
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat.div_(self.world)
        return flat

    def broadcast_(self, flat, src=0):
        if self.active:
            dist.broadcast(flat, src=src)
        return flat

    def max_scalar(self, x):
        if not self.active:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_scalar(self, x):
        if not self.active:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def broadcast_scalar(self, x, src=0):
        """Rank ``src``'s value on every rank (control decisions must be taken once per job)."""
        if not self.active:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device or "cpu")
        dist.broadcast(t, src=src)
        return float(t.item())

    def barrier(self):
        if self.active:
            dist.barrier()

    def attach(self, model):
        """Install the gradient all-reduce into a model's step and align the replicas' weights."""
        model.grad_hook = self.allreduce_mean_
        model.history_hook = self.sync_history
        model.history_join = self.join_history
        # step programs carry the collectives themselves -- unless the history exchange's fixed block is too large for
        # that (set_history_cap): then every rank runs both collectives between the program's phases
        model.native_coll = self.world if (self.native and self.native_history) else 0
        model._par = self                                         # (for the exchange's block capacity: history_cap)
        model.dropout_seed = int(getattr(model, "dropout_seed", 0)) + 7919 * self.rank   # independent masks per rank
        self.broadcast_(model.theta)

    def set_history_cap(self, cap, d_max):
        """The job-wide bound of the rows a step can update (``cap``, or None) and the widest history (``d_max`` floats).
        The library's exchange (HIST_PACK / ALLGATHER_I32 / HIST_APPLY) always moves a FIXED block of cap x (d + 1) words
        per rank and layer; above HISTORY_FIXED_LIMIT_BYTES per rank -- a bound that saturates at the graph's size: Reddit
        with 8 GPUs and d = 128 would receive ~1 GB per layer and step -- or without a bound, the history exchange of
        EVERY rank goes through torch.distributed instead (a size exchange + a gather padded to the largest block).  The
        decision depends on job-wide constants only, so all ranks take it alike."""
        self.history_cap = None if cap is None else int(cap)
        fits = cap is not None and (int(cap) + 3) // 4 * 4 * (int(d_max) + 1) * 4 <= self.HISTORY_FIXED_LIMIT_BYTES
        self.native_history = bool(self.native and fits)
        return self.native_history

    def abort(self):
        """A rank-local failure ahead of a collective: abort the library's communicator so that the peers' collectives
        fail instead of blocking for good (there is no watchdog on it; torch's process group has its own timeout)."""
        if self.native:
            from ._ffi import lib
            lib.sgcn_coll_abort()
            self.native = self.native_history = self._native_owner = False

    def sync_history(self, history, idx, rows, scatter_fn):
        """All-gather this step's (idx[n], rows[n x d]) and apply every rank's update to the local
        replica ``history`` in rank order -- the apply happens in ``join_history``, which every reader
        of the history calls first (Model.run_one_step, save, the trainer's copy into the test model).

        Fixed-capacity path (``history_cap`` rows, set by the trainer from batch size and
        degrees): ONE asynchronous all-gather of a preallocated int32 buffer ``[cap ids | cap x d
        row bits]`` per rank, ids padded with -1 (the scatter kernel skips them) -- no size exchange,
        no host synchronisation.  Without a bound: a size exchange (one host sync) followed by a
        padded all-gather."""
        if not self.active:
            scatter_fn(history, idx, rows)
            return
        dev = rows.device
        n, d = int(rows.shape[0]), int(rows.shape[1])
        if self.native and self.native_history:
            try:
                return self._sync_history_native(history, idx, rows, n, d, dev)
            except BaseException:
                self.abort()                 # (this rank will not reach the collective: do not leave the peers in it)
                raise
        cap = self.history_cap
        if cap is not None and n > cap:      # a rank-local branch here would desynchronise the ranks
            raise RuntimeError("history exchange: %d rows exceed history_cap=%d" % (n, cap))
        if cap is None or cap * (d + 1) * 4 > self.HISTORY_FIXED_LIMIT_BYTES:
            sizes = torch.zeros(self.world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
            cap = int(sizes.max().item())
            bufs = None
        else:
            # one send / receive pair per history tensor in flight (a model with two aggregator layers has two
            # exchanges pending between a step's optimizer and the next step's first aggregator)
            key = (cap, d, dev, len(self._pending))
            bufs = self._hist_bufs.get(key)
        if bufs is None:
            send = torch.empty(cap * (d + 1), dtype=torch.int32, device=dev)
            recv = torch.empty(self.world * cap * (d + 1), dtype=torch.int32, device=dev)
            bufs = (send, recv)
            if self.history_cap is not None and cap == self.history_cap:
                self._hist_bufs[(cap, d, dev, len(self._pending))] = bufs
        send, recv = bufs
        send[:n] = idx.tensor() if hasattr(idx, "tensor") and not torch.is_tensor(idx) else idx
        send[n:cap] = -1
        send[cap:].view(torch.float32).view(cap, d)[:n] = rows
        work = dist.all_gather_into_tensor(recv, send, async_op=True)
        self._pending.append((work, recv, history, cap, d, scatter_fn))

    def _sync_history_native(self, history, idx, rows, n, d, dev):
        """The exchange of a step that runs layer by layer, on the library's communicator and the step's stream (what the
        compiled program does with its HIST_PACK / ALLGATHER_I32 / HIST_APPLY ops): nothing stays pending."""
        from ._ffi import check, lib
        cap = self.history_cap
        if cap is None:
            raise RuntimeError("history exchange: the trainer sets history_cap (the rows a step can update) first")
        if n > cap:
            raise RuntimeError("history exchange: %d rows exceed history_cap=%d" % (n, cap))
        cap = (cap + 3) // 4 * 4
        key = ("native", cap, d, dev)
        bufs = self._hist_bufs.get(key)
        if bufs is None:
            bufs = self._hist_bufs[key] = (torch.empty(cap * (d + 1), dtype=torch.int32, device=dev),
                                           torch.empty(self.world * cap * (d + 1), dtype=torch.int32, device=dev))
        send, recv = bufs
        ids = idx.tensor() if hasattr(idx, "tensor") and not torch.is_tensor(idx) else idx
        ids = ids.to(torch.int32)
        rows = rows if rows.stride(1) == 1 else rows.contiguous()
        st = torch.cuda.current_stream().cuda_stream
        check(lib.sgcn_hist_pack_f32(ids.data_ptr(), n, rows.data_ptr(), int(rows.stride(0)), d, cap, send.data_ptr(), st))
        check(lib.sgcn_coll_allgather_i32(send.data_ptr(), recv.data_ptr(), cap * (d + 1), st))
        okey = ("owner", int(history.shape[0]), dev)
        owner = self._hist_bufs.get(okey)
        if owner is None:
            owner = self._hist_bufs[okey] = torch.zeros(int(history.shape[0]), dtype=torch.int32, device=dev)
        check(lib.sgcn_hist_apply_f32(history.data_ptr(), int(history.stride(0)), recv.data_ptr(), self.world, cap, d,
                                      owner.data_ptr(), st))

    def join_history(self):
        """Wait for the history exchanges in flight and apply them (rank order, issue order).  On RCCL the wait is a
        stream dependency, not a host block."""
        if not self._pending:
            return
        pend, self._pending = self._pending, []
        for work, recv, history, cap, d, scatter_fn in pend:
            work.wait()
            per = cap * (d + 1)
            for r in range(self.world):
                blk = recv[r * per:(r + 1) * per]
                scatter_fn(history, blk[:cap], blk[cap:].view(torch.float32).view(cap, d))

    def shutdown(self):
        self.join_history()
        if self.native:                   # (every user drops its reference; the last one destroys the communicator)
            from ._ffi import lib
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            lib.sgcn_coll_destroy()
        self.native = self.native_history = self._native_owner = False
        if self.active and dist.is_initialized():
            dist.destroy_process_group()


# ---- row-block sharded full-graph SpMM (SURVEY.md §8e "Full-graph SpMM", BASELINE configs 4/5) ----
def partition_rows_by_nnz(indptr, world):
    """Row boundaries ``b[0..world]`` of contiguous vertex ranges with (nearly) equal nonzero
    counts: power-law rows make equal-vertex ranges badly unbalanced, equal-nnz ranges are what
    keeps the ranks' SpMM times equal.  Empty ranges are allowed (more ranks than rows)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    M = indptr.shape[0] - 1
    nnz = int(indptr[-1])
    b = np.zeros(world + 1, dtype=np.int64)
    b[world] = M
    for r in range(1, world):
        # first row boundary whose prefix reaches r/world of the nonzeros
        b[r] = min(M, max(b[r - 1], int(np.searchsorted(indptr, (nnz * r + world - 1) // world, side='left'))))
    return b


class ShardedSpMM(object):
    """``C = A . B`` and ``dB = A^T . dC`` with the vertex set split into contiguous ranges, one
    per GPU, balanced by nonzeros (of A and A^T together, so forward and backward are both
    even).  Rank g owns rows ``[b[g], b[g+1])`` of A and C -- and the same rows of A^T and dB,
    which keeps a layer's activations and their gradients sharded identically.

    Both products are "row block of a static sparse matrix x full dense matrix" on the
    single-GPU kernels, so the only data-path question is where the dense operand lives (§8e):
      * ``forward(B)`` / ``backward(dC)``          -- operand replicated / resident: no collective;
      * ``forward_allgather(B_local)`` / ``backward_allgather(dC_local)`` -- the operand is a
        layer activation sharded like C: the ranks all-gather their row blocks first
        (xGMI-bound for large operands; bench.py reports both variants)."""

    ROW_WEIGHT = 7       # a row of a block costs the column sweep what ~7 of its nonzeros do (its line of C, its share of
                         # the rounds of resident tiles): blocks of S-RMAT 10 M with EQUAL nonzeros took 4.1 ms per fwd + bwd
                         # (74 k hub rows) to 9.2 ms (3.7 M sparse rows); least squares over 16 blocks of two partitions:
                         # t = 0.193 ms per M nonzeros + 1.32 ms per M rows (profiles/r43_rmat10m_blocks.json)

    def __init__(self, par, adj, device, kernel="cs", with_transpose=True, d=None, G=None, plan_kw=None, row_weight=None):
        """d: width of the dense operand, when known up front -- picks the sweep's lane-group count for it
        (ops.ColumnSweepCSR.choose_g); None: one group per wavefront.  G: that count given outright.  plan_kw: further
        arguments of the blocks' ops.ColumnSweepCSR plans (align, warp).  row_weight: nonzero-equivalents a row adds to
        its block's load (None: ROW_WEIGHT for the column sweep, 0 otherwise)."""
        from . import ops
        adj = adj.tocsr()
        if adj.shape[0] != adj.shape[1]:
            raise ValueError("ShardedSpMM shards one vertex set: the matrix must be square")
        self.par, self.device, self.kernel = par, device, kernel
        self.shape = (int(adj.shape[0]), int(adj.shape[1]))
        adj_t = ops.transpose_host(adj) if with_transpose else None      # (parallel counting sort; SciPy's pass until round 5)
        load = adj.indptr.astype(np.int64) + (adj_t.indptr.astype(np.int64) if with_transpose else 0)
        if row_weight is None:
            row_weight = self.ROW_WEIGHT if kernel == "cs" else 0
        self.row_weight = int(row_weight)
        if self.row_weight:              # (prefix sums: row r adds its nonzeros in both directions + the weight per direction)
            load = load + np.arange(load.shape[0], dtype=np.int64) * (self.row_weight * (2 if with_transpose else 1))
        self.bounds = partition_rows_by_nnz(load, par.world)
        self.lo, self.hi = int(self.bounds[par.rank]), int(self.bounds[par.rank + 1])
        self.row_counts = [int(self.bounds[r + 1] - self.bounds[r]) for r in range(par.world)]
        blk = adj[self.lo:self.hi]
        blk_t = adj_t[self.lo:self.hi] if with_transpose else None
        self.local_nnz = int(blk.nnz)
        self.distinct_cols = int(np.unique(blk.indices).shape[0])     # rows of B this block touches
        self.A = self.AT = self._mm = None
        if self.hi == self.lo or kernel is None:   # more ranks than row blocks (this rank only joins collectives), or the
            pass                                   # partition + collectives alone (kernel=None: the CPU dry run of bench.py)
        elif kernel == "cs":
            if G is None:
                G = ops.ColumnSweepCSR.choose_g(d, blk.nnz / max(blk.shape[0], 1), blk.shape[0]) if d else 1
            kw = dict(plan_kw or {})
            if G == 1:
                kw.pop('align', None)
            # a block small enough that its rows are split anyway to fill one round of resident tiles (an eighth of
            # S-Reddit): split them by COLUMN RANGE, a range per half of the XCDs (ops.ColumnSweepCSR.choose_ranges)
            def plan(m):
                k2 = dict(kw)
                nr = ops.ColumnSweepCSR.choose_ranges(m.shape[0], m.nnz, m.shape[1], G) if k2.pop('col_ranges', 'auto') == 'auto' else 0
                if nr:
                    k2.pop('warp', None)
                    return ops.ColumnSweepCSR(m, device, G=G, col_ranges=nr, T=k2.get('T', 0))
                return ops.ColumnSweepCSR(m, device, G=G, **k2)
            self.A = plan(blk)
            self.AT = plan(blk_t) if with_transpose else None
            self._mm = ops.spmm_cs
        else:
            self.A = ops.DeviceCSR.from_scipy(blk, device)
            self.AT = ops.DeviceCSR.from_scipy(blk_t, device) if with_transpose else None
            self._mm = ops.spmm

    def autotune(self, B, dC=None):
        if self.kernel == "cs" and self.A is not None:
            self.A.autotune(B)
            if self.AT is not None and dC is not None:
                self.AT.autotune(dC)

    def _local(self, A, X, out):
        if A is None:
            return torch.empty((0, X.shape[1]), dtype=torch.float32, device=X.device)
        return self._mm(A, X, out=out)

    def forward(self, B, out=None):
        """C[lo:hi] = A[lo:hi, :] . B        (B: all rows, resident on this GPU)."""
        return self._local(self.A, B, out)

    def backward(self, dC, out=None):
        """dB[lo:hi] = A^T[lo:hi, :] . dC    (dC: all rows, resident on this GPU)."""
        return self._local(self.AT, dC, out)

    def allgather_rows(self, X_local):
        """All rows from the ranks' blocks (ragged blocks travel padded to the largest one).
        The result keeps 16-byte aligned rows (pitch = d rounded up to 4 floats), which the
        column-sweep kernel requires of its dense operand."""
        par = self.par
        d = int(X_local.shape[1])
        pitch = (d + 3) // 4 * 4
        if not par.active:
            if X_local.stride(0) % 4 == 0 and X_local.stride(1) == 1:
                return X_local
            full = torch.zeros((X_local.shape[0], pitch), dtype=torch.float32, device=X_local.device)
            full[:, :d] = X_local
            return full[:, :d]
        cap = max(self.row_counts)
        send = torch.zeros((cap, pitch), dtype=torch.float32, device=X_local.device)
        send[:X_local.shape[0], :d] = X_local
        recv = torch.empty((par.world, cap, pitch), dtype=torch.float32, device=X_local.device)
        dist.all_gather_into_tensor(recv.view(par.world * cap, pitch), send)
        if all(c == cap for c in self.row_counts):
            return recv.view(par.world * cap, pitch)[:, :d]
        full = torch.empty((self.shape[0], pitch), dtype=torch.float32, device=X_local.device)
        for r in range(par.world):
            full[int(self.bounds[r]):int(self.bounds[r + 1])] = recv[r, :self.row_counts[r]]
        return full[:, :d]

    def forward_allgather(self, B_local, out=None):
        return self.forward(self.allgather_rows(B_local), out=out)

    def backward_allgather(self, dC_local, out=None):
        return self.backward(self.allgather_rows(dC_local), out=out)
