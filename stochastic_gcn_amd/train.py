"""Training driver -- drop-in for gcn/train.py (same flags, same flow, same log lines).

    python -m stochastic_gcn_amd.train --dataset reddit --normalization graphsage --weight_decay 0 \\
        --dropout 0.2 --layer_norm --hidden1 128 --num_fc_layers 2 --epochs 30 --early_stopping 30 \\
        --batch_size=512 --test_batch_size=512 --cv --cvd --test_cv --degree=1 --test_degree=1

(= gcn/config/reddit.config:2 + the CVD+PP switches of README.md:46-55.)  Multi-GPU: launch with
``python -m torch.distributed.run --nproc-per-node N -m stochastic_gcn_amd.train ...``.

Flow (gcn/train.py:73-383): load_data -> PP products on the GPU (K11) -> placeholders ->
train/test models from one template (shared weights) -> two schedulers -> SGDTrain epochs with
validation, the reference's two log lines per epoch, early stopping -> Test().
"""
from __future__ import division, print_function

import queue
import sys
import threading
from time import time

import numpy as np
import scipy.sparse as sp
import torch

from . import ops
from .flags import FLAGS
from .models import make_template
from .parallel import DataParallel
from .plaingcn import PlainGCN
from .scheduler import NativePrefetcher, PyScheduler
from .utils import Averager, calc_f1, f1_from_classes, load_data
from .vrgcn import VRGCN


class Placeholder(object):
    """Opaque feed-dict key (the reference uses tf.placeholder objects the same way)."""

    def __init__(self, name, shape=None):
        self.name, self.shape = name, shape

    def __repr__(self):
        return "<ph %s>" % self.name


def make_placeholders(L, num_classes):
    """gcn/train.py:89-103."""
    return {
        'adj': [Placeholder('adj_%d' % l) for l in range(L)],
        'madj': [Placeholder('madj_%d' % l) for l in range(L)],
        'fadj': [Placeholder('fadj_%d' % l) for l in range(L)],
        'fields': [Placeholder('field_%d' % l) for l in range(L + 1)],
        'ffields': [Placeholder('ffield_%d' % l) for l in range(L)],
        'scales': [Placeholder('scale_%d' % l) for l in range(L)],
        'labels': Placeholder('labels', (None, num_classes)),
        'dropout': Placeholder('dropout', ()),
    }


def plan_cache_paths(dataset):
    """Where the column-sweep plans of a dataset's two adjacency matrices are cached: beside the
    reference-schema ``.npz`` dataset cache (utils.cache_path), or under $SGCN_PLAN_CACHE_DIR; None when
    neither exists (synthetic stand-ins without a cache directory rebuild their plans, ~1 s)."""
    import os
    from .utils import cache_path
    root = os.environ.get("SGCN_PLAN_CACHE_DIR")
    if root:
        base = os.path.join(root, os.path.basename(cache_path(dataset))[:-4])
    elif os.path.exists(cache_path(dataset)):
        base = cache_path(dataset)[:-4]
    else:
        return None, None
    return base + ".csplan.train.npz", base + ".csplan.full.npz"


# What the static-graph kernels cost end to end on the MI355X (profiles/r60_bench_setup.json, S-Reddit: 23.2 M nonzeros,
# d = 602, 16 host cores): the row-gather kernel needs the CSR in HBM and a row-pointer pass (5 ms) and takes 7.66 ms per
# product; the column sweep needs its host plan + upload (0.185 s: 8 ns per nonzero), a clock autotune worth ~62 products,
# and takes 0.40 of the row kernel's time per product.
CS_PLAN_S_PER_NNZ = 8.0e-9
CS_AUTOTUNE_PRODUCTS = 62
CS_TIME_RATIO = 0.40
ROWS_S_PER_NNZ_FLOAT = 7.66e-3 / (23173306 * 602.0)


def static_kernel_for(nnz, d, products):
    """'rows' or 'cs': the kernel with the lower expected END-TO-END time for `products` products of one static matrix
    with a d-wide dense operand -- setup included.  The reference computes each PP product once (gcn/utils.py:321-322, the
    result cached in the dataset's .npz), and for one product no plan pays: the column sweep breaks even at ~83 products of
    S-Reddit (a full-batch model's layers over a few epochs), which is what bench.py reports as
    setup.products_to_break_even_vs_rows_kernel."""
    t_rows = ROWS_S_PER_NNZ_FLOAT * nnz * d
    rows = products * t_rows
    cs = CS_PLAN_S_PER_NNZ * nnz + (CS_AUTOTUNE_PRODUCTS + products) * CS_TIME_RATIO * t_rows
    return 'cs' if cs < rows else 'rows'


def pp_products(train_adj, full_adj, features, device, cache=(None, None), stats=None, products=1):
    """train_feats = train_adj . feats, test_feats = full_adj . feats (gcn/utils.py:169-170,
    321-322).  Dense features: an SpMM kernel on the GPU (K11), chosen by expected end-to-end time for the number of
    times the product will run with one plan (``products``; ``static_kernel_for``): once, as in the reference -- the
    row-gather kernel sgcn_spmm_csr_f32, no plan; many times -- the column sweep (sgcn_spmm_cs_f32, the kernel bench.py
    times), its host plan cached beside the dataset, or for a graph with communities the LDS-staged sweep
    (ops.LdsSweepCSR.for_graph).  Sparse features: a sparse x sparse product, done once on the host with SciPy exactly
    like the reference."""
    if sp.issparse(features):
        return train_adj.dot(features).tocsr(), full_adj.dot(features).tocsr()
    X = features.to(device) if isinstance(features, torch.Tensor) else \
        torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(device)
    d = int(X.shape[1])
    if X.stride(0) % 4 != 0:            # the sweep reads 16-byte vectors: pad the row pitch once
        Xp = torch.zeros((X.shape[0], (d + 3) // 4 * 4), dtype=torch.float32, device=device)
        Xp[:, :d] = X
        X = Xp[:, :d]
    out = []
    for a, path in zip((train_adj, full_adj), cache):
        t0 = time()
        if static_kernel_for(a.nnz, d, products) == 'rows':
            R = ops.DeviceCSR.from_scipy(a, device)
            out.append(ops.spmm(R, X).contiguous())
            if stats is not None:
                torch.cuda.synchronize()
                stats.append(dict(plan_from_cache=False, pace=None, kernel="sgcn::spmm_seg_kernel", products=products,
                                  end_to_end_s=time() - t0))
            continue
        # a large graph WITH communities (>= 90 % of its nonzeros inside tiles that share their columns): the LDS-staged
        # sweep + the column sweep on the rest; anything else: the column sweep alone
        L = ops.LdsSweepCSR.for_graph(a, device) if (path is None and a.nnz >= 2000000 and d >= 128) else None
        if L is not None:
            L.autotune(X)
            if stats is not None:
                stats.append(dict(plan_from_cache=False, pace=None, kernel=L.variant(d), products=products))
            out.append(ops.spmm_lds(L, X).contiguous())
            continue
        A, hit = ops.ColumnSweepCSR.cached(a, device, path, G=ops.ColumnSweepCSR.choose_g(d, a.nnz / max(a.shape[0], 1), a.shape[0]))
        if d not in A.pace:
            A.autotune(X)               # once per plan and width; stored with the cached plan
        A.store_if_cached()
        if stats is not None:
            stats.append(dict(plan_from_cache=hit, pace=A.pace.get(d), kernel=A.variant(d), products=products))
        out.append(ops.spmm_cs(A, X).contiguous())
    return out[0], out[1]


class Prefetcher(object):
    """Runs ``sch.minibatch`` on a host thread a few batches ahead of the GPU.  The sampler is
    stateful, so there is exactly one producer and batches are consumed in call order -- the
    sample sequence is identical to the synchronous loop."""

    def __init__(self, sch, batch_size, depth, n_steps, slots=None):
        self.q = queue.Queue(maxsize=max(1, depth))
        self.t = threading.Thread(target=self._run, args=(sch, batch_size, n_steps, slots), daemon=True)
        self.t.start()

    def _run(self, sch, batch_size, n_steps, slots):
        t_make = 0.0
        for i in range(n_steps):
            t0 = time()
            b = next_minibatch(sch, batch_size, slots[i % len(slots)] if slots else None)
            t_make += time() - t0
            self.q.put(b)
        self.make_s = t_make          # producer time spent building batches (excludes queue waits)
        self.q.put(None)

    def next(self):
        return self.q.get()


class ParallelPrefetcher(object):
    """NON-PARITY fast mode (``--sampler_threads N``): N sampler instances,
    each with its own private CSR copy and RNG stream (seed + 1000 * k), build batches
    ``k, k + N, k + 2N, ...`` concurrently -- the packed C call runs without the GIL -- and the
    consumer receives them in batch order.  Every batch is a valid draw of the same sampling
    distribution, but the sequence is no longer the reference's single mt19937 stream (that is
    what N = 1, the default, reproduces bit for bit).  Non-PP / NS / Exact runs are sampler-bound
    (S-Reddit NS, L = 2, degree 20: 8.4 ms of host sampling per 512-vertex batch against < 1 ms of
    GPU work), which is what this mode is for."""

    def __init__(self, schedulers, batches, plan_t, slots, depth):
        self.sch, self.batches, self.plan_t, self.slots = schedulers, batches, plan_t, slots
        n = len(schedulers)
        self.window = max(1, depth) * n               # batches allowed in flight ahead of the consumer
        assert len(slots) >= self.window + 2, "staging ring too small for the prefetch window"
        self.cv = threading.Condition()
        self.done, self.consumed, self.pos, self.error = {}, 0, 0, None
        self.threads = [threading.Thread(target=self._run, args=(k,), daemon=True) for k in range(n)]
        for t in self.threads:
            t.start()

    def _run(self, k):
        n = len(self.sch)
        try:
            for i in range(k, len(self.batches), n):
                with self.cv:
                    while i >= self.consumed + self.window and self.error is None:
                        self.cv.wait()
                    if self.error is not None:
                        return
                pb = self.sch[k].batch_packed(self.batches[i], self.plan_t, self.slots[i % len(self.slots)])
                with self.cv:
                    self.done[i] = pb
                    self.cv.notify_all()
        except BaseException as e:          # surface producer failures in the consumer
            with self.cv:
                self.error = e
                self.cv.notify_all()

    def next(self):
        i = self.pos
        if i >= len(self.batches):
            return None
        with self.cv:
            while i not in self.done and self.error is None:
                self.cv.wait()
            if self.error is not None:
                raise self.error
            pb = self.done.pop(i)
            self.consumed = i + 1
            self.cv.notify_all()
        self.pos += 1
        return pb


def epoch_batches(data, batch_size, n_steps):
    """The id slices ``minibatch`` would walk (wrapping like next_minibatch), as a list."""
    out, pos, n = [], 0, len(data)
    for _ in range(n_steps):
        if pos == n:
            pos = 0
        end = min(n, pos + batch_size)
        out.append(data[pos:end])
        pos = end
    return out


def next_minibatch(sch, batch_size, slot=None):
    """The next minibatch as a PackedBatch (one C call: sampler + CSR/plan packing into a pinned
    staging slot), wrapping to the start of the (already shuffled) shard when it runs out: in a
    multi-GPU job every rank must take the same number of steps per epoch, and vertex-range
    shards hold slightly different numbers of train ids."""
    fd = sch.minibatch_packed(batch_size, FLAGS.plan_t, slot)
    if fd is None:
        sch.start = 0
        fd = sch.minibatch_packed(batch_size, FLAGS.plan_t, slot)
    return fd


def stop_verdict(par, epoch, cost_val, amt_data):
    """0 = go on, 1 = early stopping, 2 = epoch / data budget reached (gcn/train.py:231-235), the
    same value on every rank: rank 0's validation history and the job-wide amount of data decide."""
    amt = par.sum_scalar(amt_data)
    stop = 0
    if epoch > FLAGS.early_stopping and cost_val[-1] > np.mean(cost_val[-(FLAGS.early_stopping + 1):-1]):
        stop = 1
    elif amt >= FLAGS.data and epoch > FLAGS.epochs:
        stop = 2
    return int(par.broadcast_scalar(stop))


class Trainer(object):
    """Everything gcn/train.py does between flag parsing and the end of training, as an object
    (so bench.py can time epochs of the real training path).  FLAGS must be set before."""

    def __init__(self, data=None, verbose=True):
        np.random.seed(FLAGS.seed)
        torch.manual_seed(FLAGS.seed)
        if not torch.cuda.is_available():
            raise RuntimeError("training needs an MI355X (no CPU fallback for the SpMM/history path)")
        par = DataParallel(device=None, init=False)
        dev_index = par.local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        self.device = device = torch.device('cuda', dev_index)
        self.par = par = DataParallel(device=device)
        if par.world > torch.cuda.device_count():
            # several ranks time-slice ONE GPU (the gloo smoke configuration): every cross-stream event of
            # the step program's auxiliary stream then costs a context switch between the processes
            # (measured: 95 ms per step instead of 1.2) -- keep the program's launches on one stream
            FLAGS.update(agg_overlap=False)
        self.log = log = print if (par.rank == 0 and verbose) else (lambda *a, **k: None)

        (num_data, train_adj, full_adj, features, train_features, test_features, labels,
         train_d, val_d, test_d) = data if data is not None else load_data(FLAGS.dataset)
        self.pp_stats = []
        if train_features is None:
            train_features, test_features = pp_products(train_adj, full_adj, features, device,
                                                        cache=plan_cache_paths(FLAGS.dataset), stats=self.pp_stats,
                                                        products=FLAGS.pp_products)
        if FLAGS.gradvar:
            log('Analyze mode...')
            full_adj = train_adj.copy()
            test_features = train_features
        log('Features shape = {}, training edges = {}, testing edges = {}'.format(
            features.shape, train_adj.nnz, full_adj.nnz))
        log('{} training data, {} validation data, {} testing data.'.format(len(train_d), len(val_d), len(test_d)))
        self.num_data, self.labels = num_data, labels
        self.train_d, self.val_d, self.test_d = train_d, val_d, test_d

        self.multitask = multitask = FLAGS.dataset == 'ppi'
        L = FLAGS.num_layers - 1 if FLAGS.preprocess else FLAGS.num_layers
        test_L = FLAGS.num_layers - 1 if FLAGS.test_preprocess else FLAGS.num_layers
        self.placeholders = placeholders = make_placeholders(max(L, test_L), labels.shape[1])

        t = time()
        log('Building model...')
        train_cls = VRGCN if FLAGS.cv else PlainGCN
        test_cls = VRGCN if FLAGS.test_cv else PlainGCN

        def model_func(model, nbr_features, adj, preprocess, is_training, cvd, _store=None):
            return model(FLAGS.num_layers, preprocess, placeholders, features, nbr_features, adj, cvd,
                         multitask=multitask, is_training=is_training, device=device, _store=_store)
        create_model = make_template('model', model_func)
        self.train_model = create_model(train_cls, nbr_features=train_features, adj=train_adj,
                                        preprocess=FLAGS.preprocess, is_training=True, cvd=FLAGS.cvd)
        self.test_model = create_model(test_cls, nbr_features=test_features, adj=full_adj,
                                       preprocess=FLAGS.test_preprocess, is_training=False, cvd=FLAGS.test_cvd)
        log('Finised in {} seconds'.format(time() - t))
        if par.active:
            par.attach(self.train_model)
            self.train_d = par.shard_ids(train_d, num_data).astype(np.int32)
            # a vertex range without a single train id (tiny or skewed sets, many ranks) would feed
            # empty batches into kernels that reject n == 0 while the peers wait in the all-reduce:
            # every rank falls back to round-robin id sharding when ANY range is empty
            if par.max_scalar(1.0 if len(self.train_d) == 0 else 0.0) > 0:
                self.train_d = np.ascontiguousarray(np.asarray(train_d)[par.rank::par.world], dtype=np.int32)
                if len(self.train_d) == 0:
                    raise RuntimeError("data-parallel training: %d train ids cannot be sharded over %d ranks"
                                       % (len(train_d), par.world))
            # |fields[l]| <= batch * prod(1 + degree): lets the history exchange run without a
            # size round-trip (parallel.DataParallel.sync_history)
            bound = int(FLAGS.batch_size)
            for _ in range(L):
                bound = min(int(num_data), bound * (1 + int(FLAGS.degree)))
            par.set_history_cap(bound, max([int(h.shape[1]) for hs in self.train_model.history for h in hs] or [1]))
            self.train_model.native_coll = par.world if (par.native and par.native_history) else 0

        train_degrees = np.array([FLAGS.degree] * L, dtype=np.int32)
        test_degrees = np.array([FLAGS.test_degree] * test_L, dtype=np.int32)
        self.train_sch = PyScheduler(train_adj, labels, L, train_degrees, placeholders,
                                     par.sampler_seed(FLAGS.seed), self.train_d, cv=FLAGS.cv,
                                     importance=FLAGS.importance)
        # the evaluation sampler is seeded IDENTICALLY on every rank: all ranks evaluate the same
        # vertices with the same (all-reduced) weights, so validation cost -- and with it the
        # early-stopping decision -- agrees across the job
        self.eval_sch = PyScheduler(full_adj, labels, test_L, test_degrees, placeholders,
                                    int(FLAGS.seed), cv=FLAGS.test_cv,
                                    importance=FLAGS.test_importance)
        self.sess = None   # API compatibility: run_one_step(sess, feed_dict)
        from .scheduler import StagingSlot
        nthr = max(1, int(FLAGS.sampler_threads))
        self.train_schs, self.eval_schs = [self.train_sch], [self.eval_sch]
        for k in range(1, nthr):           # non-parity fast mode: extra sampler instances
            self.train_schs.append(PyScheduler(train_adj, labels, L, train_degrees, placeholders,
                                               par.sampler_seed(FLAGS.seed) + 1000 * k, cv=FLAGS.cv,
                                               importance=FLAGS.importance))
            self.eval_schs.append(PyScheduler(full_adj, labels, test_L, test_degrees, placeholders,
                                              int(FLAGS.seed) + 1000 * k, cv=FLAGS.test_cv,
                                              importance=FLAGS.test_importance))
        ring = max(FLAGS.prefetch, 1) * nthr + 3
        self.slots = [StagingSlot(pin=True) for _ in range(ring)]
        self.eval_slots = [StagingSlot(pin=True) for _ in range(ring if nthr > 1 else 4)]
        self.cost_val = []
        self.avg_loss = Averager(1)
        self.avg_acc = Averager(1)
        self.last_epoch = {}

    # ---- evaluation (gcn/train.py:133-160) ------------------------------------------------------
    class _EvalSink(object):
        """Pinned host memory the evaluation batches' result vectors are copied into as they are produced (copy engine,
        stream order): no torch arithmetic or concatenation kernel on the evaluation path, one synchronisation per sweep."""

        def __init__(self, floats):
            self.buf = torch.empty(max(int(floats), 1), dtype=torch.float32).pin_memory()
            self.pos = 0

        def push(self, vec):
            n = int(vec.numel())
            if self.pos + n > self.buf.numel():
                raise RuntimeError("evaluation sink overflow")
            out = self.buf[self.pos:self.pos + n]
            out.copy_(vec, non_blocking=True)
            self.pos += n
            return out

    def evaluate(self, data):
        total_pred, total_labs, total_cls, stats = [], [], [], []
        t_test = time()
        N = len(data)
        chunks = [data[st:min(st + FLAGS.test_batch_size, N)] for st in range(0, N, FLAGS.test_batch_size)]
        sink = None
        if not self.multitask:
            sink = getattr(self, '_eval_sink', None)
            need = 4 * len(chunks) + 3 * N
            if sink is None or sink.buf.numel() < need:
                sink = self._eval_sink = Trainer._EvalSink(need)
            sink.pos = 0
        if FLAGS.native_prefetch and (FLAGS.prefetch > 0 or len(self.eval_schs) > 1):
            pre = NativePrefetcher(self.eval_schs if len(self.eval_schs) > 1 else self.eval_sch, chunks,
                                   FLAGS.plan_t, depth=max(FLAGS.prefetch, 1))
        elif len(self.eval_schs) > 1:
            pre = ParallelPrefetcher(self.eval_schs, chunks, FLAGS.plan_t, self.eval_slots, max(FLAGS.prefetch, 1))
        else:
            pre = None
        def fetch(i):
            return pre.next() if pre else \
                self.eval_sch.batch_packed(chunks[i], FLAGS.plan_t, self.eval_slots[i % len(self.eval_slots)])
        nxt = fetch(0) if chunks else None
        vecs, rows, slow = [], [], None
        for k in range(len(chunks)):
            batch = nxt
            nxt = fetch(k + 1) if k + 1 < len(chunks) else None
            if nxt is not None and pre is not None:      # its H2D copy starts one batch early (train_epoch)
                self.test_model.stage(nxt)
            self.test_model.eval_light = not self.multitask       # (for this call only: models.py _run_program / loss)
            self.test_model.eval_sink = sink
            try:
                los, acc, prd = self.test_model.run_one_step(self.sess, batch, sync=False)
            finally:
                self.test_model.eval_light = False
                self.test_model.eval_sink = None
            vec = self.test_model.__dict__.pop('eval_vec', None)
            if vec is not None:                 # single-label: ONE vector per batch [stats | CE | hit | classes per row]
                vecs.append(vec)
                rows.append(self.test_model.eval_rows)
                continue
            # the other result forms (multi-label, or a step that ran layer by layer): loss, accuracy and the rows'
            # classes -- or predictions and labels -- go to pinned host memory as they are produced (copy engine, stream
            # order; los / acc are views the next batch overwrites); the sums are taken on the host after ONE
            # synchronisation.  No torch arithmetic or concatenation kernel on this path either.
            cls = getattr(self.test_model, 'eval_classes', None)
            r = int(prd.shape[0])
            if slow is None:
                per_row = 1 if cls is not None else 2 * int(prd.shape[1])
                slow = getattr(self, '_eval_sink_slow', None)
                need = 2 * len(chunks) + per_row * N
                if slow is None or slow.buf.numel() < need:
                    slow = self._eval_sink_slow = Trainer._EvalSink(need)
                slow.pos = 0
            stats.append((slow.push(los.reshape(1)), slow.push(acc.reshape(1)), r))
            if cls is not None:                 # single-label: the loss kernel's class indices, 4 bytes per row
                total_cls.append(slow.push(cls))
            else:
                total_pred.append(slow.push(prd.reshape(-1)).view(r, -1))
                total_labs.append(slow.push(self.test_model.cur.labels.reshape(-1)).view(r, -1))
        if pre is not None and hasattr(pre, 'close'):
            pre.close()
        assert not (vecs and stats), "evaluation batches took both result forms"
        if vecs:
            # the vectors are already on their way to pinned host memory: ONE synchronisation, then (loss, accuracy) x rows
            # per batch summed over the batches in fp32 on the host (no torch kernel on this path)
            torch.cuda.current_stream().synchronize()
            tot = np.zeros(2, dtype=np.float32)
            v = []
            for vec, r in zip(vecs, rows):
                hv = vec.numpy() if not vec.is_cuda else vec.cpu().numpy()
                tot += hv[2:4] * np.float32(r)
                v.append(hv[4 + 2 * r:4 + 3 * r])
            tot = tot / np.float32(max(N, 1))
            v = np.concatenate(v).astype(np.int64)
            micro, macro = f1_from_classes(v // 4096, v % 4096)
            return float(tot[0]), float(tot[1]), micro, macro, (time() - t_test)
        if not stats:
            return 0.0, 0.0, 0.0, 0.0, time() - t_test
        torch.cuda.current_stream().synchronize()                           # the only host sync
        tot = np.zeros(2, dtype=np.float32)
        for los, acc, r in stats:
            tot += np.array([los.numpy()[0], acc.numpy()[0]], dtype=np.float32) * np.float32(r)
        tot = tot / np.float32(max(N, 1))
        if total_cls and not total_pred:
            v = np.concatenate([c.numpy() for c in total_cls]).astype(np.int64)   # argmax(pred) + 4096 * argmax(labels) per row
            micro, macro = f1_from_classes(v // 4096, v % 4096)
        else:
            total_pred = np.concatenate([x.numpy() for x in total_pred])
            total_labs = np.concatenate([x.numpy() for x in total_labs])
            micro, macro = calc_f1(total_pred, total_labs, self.multitask)
        return float(tot[0]), float(tot[1]), micro, macro, (time() - t_test)

    # ---- one training epoch (gcn/train.py:182-209) ----------------------------------------------
    def train_epoch(self):
        par, train_model, train_sch = self.par, self.train_model, self.train_sch
        train_sch.shuffle()
        t = time()
        train_model.init_counts()
        tsch = 0
        n_steps = -(-len(self.train_d) // FLAGS.batch_size)
        if FLAGS.max_steps:
            n_steps = min(n_steps, FLAGS.max_steps)
        n_steps = int(par.max_scalar(n_steps))       # same step count on every rank
        slots = self.slots
        if FLAGS.native_prefetch and (FLAGS.prefetch > 0 or len(self.train_schs) > 1):
            # C++ sampler thread(s): one = the reference's sample sequence, N = the non-parity fast mode
            pre = NativePrefetcher(self.train_schs if len(self.train_schs) > 1 else train_sch,
                                   epoch_batches(train_sch.data, FLAGS.batch_size, n_steps),
                                   FLAGS.plan_t, depth=max(FLAGS.prefetch, 1))
        elif len(self.train_schs) > 1:
            pre = ParallelPrefetcher(self.train_schs, epoch_batches(train_sch.data, FLAGS.batch_size, n_steps),
                                     FLAGS.plan_t, slots, max(FLAGS.prefetch, 1))
        else:
            pre = Prefetcher(train_sch, FLAGS.batch_size, FLAGS.prefetch, n_steps, slots) \
                if FLAGS.prefetch > 0 else None
        outs = None
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        def fetch(it):
            return pre.next() if pre else next_minibatch(train_sch, FLAGS.batch_size, slots[it % len(slots)])
        t1 = time()
        nxt = fetch(1) if n_steps > 0 else None
        tsch += time() - t1
        for it in range(1, n_steps + 1):
            batch = nxt
            t1 = time()
            # one batch of lookahead: the NEXT minibatch's H2D copy is started before this step is queued, so that by
            # the time its own step is queued the copy has completed and the step needs no device-side wait for it
            nxt = fetch(it + 1) if it < n_steps else None
            tsch += time() - t1
            if nxt is not None and pre is not None:
                train_model.stage(nxt)
            batch.dropout = FLAGS.dropout
            # no host sync inside the epoch: loss / accuracy stay on the device
            outs = train_model.run_one_step(self.sess, batch, sync=False)
        if pre:
            assert pre.next() is None
            self.producer_s = getattr(pre, 'make_s', None)
            if hasattr(pre, 'close'):
                pre.close()
                self.producer_s = getattr(pre, 'stats', None)
        host_loop_s = time() - t        # everything queued: below the epoch time = the host runs ahead of the GPU
        ev1.record()
        torch.cuda.synchronize()
        # 'TF time' of the epoch line (gcn/train.py:227-229; scripts/analyze-time.py reads it) is the time spent
        # inside sess.run, i.e. INCLUDING the device work.  The steps here are launched asynchronously, so the
        # host-side timer of run_one_step only sees launch time: report the epoch's device-elapsed time
        # (HIP events around the step loop) when that is the larger of the two.
        train_model.run_t = max(train_model.run_t, ev0.elapsed_time(ev1) * 1e-3)
        if outs is not None:      # Averager(1) of the reference = the last step's values
            self.avg_loss.add(float(outs[1]))
            self.avg_acc.add(float(outs[2]))
        self.last_epoch = dict(train_wall_s=time() - t, steps=n_steps, sch_wait_s=tsch, host_loop_s=host_loop_s,
                               producer_s=getattr(self, 'producer_s', None),
                               sampled_edges=float(train_model.adj_sizes.sum()),
                               full_edges=float(train_model.fadj_sizes.sum()),
                               field0=float(train_model.field_sizes[0]))
        return t, tsch

    def SGDTrain(self):
        log, par, train_model = self.log, self.par, self.train_model
        if FLAGS.load:
            train_model.load(self.sess, load_history=FLAGS.gradvar)
            for h1, h2 in zip(train_model.history_vars, self.test_model.history_vars):
                h2.copy_(h1)
            return
        log('Start training...')
        for epoch in range(100000000):
            t, tsch = self.train_epoch()
            cost, acc, micro, macro, duration = self.evaluate(self.val_d)
            self.cost_val.append(cost)
            cost_val = self.cost_val
            # the reference's epoch lines (gcn/train.py:217-229); token positions are consumed by
            # scripts/analyze-time.py:39-54 and scripts/plot-convergence.py:78-86
            log("Epoch:", '%04d' % (epoch + 1),
                "train_loss=", "{:.5f}".format(self.avg_loss.mean()),
                "train_acc=", "{:.5f}".format(self.avg_acc.mean()),
                "val_loss=", "{:.5f}".format(cost),
                "val_acc=", "{:.5f}".format(acc),
                "mi F1={:.5f} ma F1={:.5f} ".format(micro, macro),
                "time=", "{:.5f}".format(time() - t),
                "ttime=", "{:.5f}".format(duration),
                "(sch {:.5f} s)".format(tsch),
                "data = {}".format(train_model.amt_data))
            G = float(2 ** 30)
            log('TF time = {}, g time = {}, G GFLOPS = {}, NN GFLOPS = {}, field sizes = {}, adj sizes = {}, fadj sizes = {}'.format(
                train_model.run_t, train_model.g_t, train_model.g_ops / G, train_model.nn_ops / G,
                train_model.field_sizes, train_model.adj_sizes, train_model.fadj_sizes))
            log('[sgcn] epoch train wall = {:.5f} s over {} steps ({} GPU(s))'.format(
                self.last_epoch['train_wall_s'], self.last_epoch['steps'], par.world))
            # the two exits of gcn/train.py:231-235, verbatim conditions (`epoch > FLAGS.epochs`:
            # the reference trains epochs + 2 epochs).  In a multi-GPU job the decision is COLLECTIVE:
            # rank 0 decides on its validation cost and the job-wide amount of data, and broadcasts
            # the verdict -- a rank-local `break` would leave the peers blocked in the next epoch's
            # all-reduce.
            stop = stop_verdict(par, epoch, cost_val, train_model.amt_data)
            if stop == 1:
                log("Early stopping...")
            if stop:
                break
        log("Optimization Finished!")
        if par.rank == 0:
            train_model.save(self.sess)

    def Test(self):
        test_cost, test_acc, micro, macro, test_duration = self.evaluate(self.test_d)
        self.log("Test set results:", "cost=", "{:.5f}".format(test_cost),
                 "accuracy=", "{:.5f}".format(test_acc),
                 "mi F1={:.5f} ma F1={:.5f} ".format(micro, macro),
                 "time=", "{:.5f}".format(test_duration))
        remaining = np.array(sorted(set(range(self.num_data)) - set(self.test_d.tolist())), dtype=np.int32)
        if FLAGS.test_cv:
            self.evaluate(remaining)


def main(argv=None):
    FLAGS.parse(argv)
    tr = Trainer()
    tr.SGDTrain()
    if FLAGS.gradvar:
        # --gradvar keeps what belongs to the path (the test graph = the training graph, histories restored by
        # --load: models.py, Trainer.__init__); the bias / variance study it drove in the reference
        # (gcn/train.py:241-276) is an analysis tool, out of scope
        tr.log('--gradvar: analysis mode set up (test graph = training graph); the bias / variance study is '
               'not part of this package -- use model.get_pred_and_grad(sess, feed) directly')
    num_runs = FLAGS.num_layers + 1 if FLAGS.test_cv else 1
    for _ in range(num_runs):
        tr.Test()
    tr.par.shutdown()


if __name__ == '__main__':
    main(sys.argv[1:])
