"""Model shell around the hot path (mirror of gcn/models.py: ``Model`` / ``GCN``).

Keeps the reference's API -- ``GCN(L, preprocess, placeholders, features, nbr_features, adj,
cvd, multitask=, is_training=)``, ``run_one_step(sess, feed_dict)`` (``sess`` accepted and
ignored), ``get_data``, ``get_pred_and_grad``, ``init_counts``, ``save`` / ``load`` and the
counters ``run_t, g_t, g_ops, nn_ops, field_sizes, adj_sizes, fadj_sizes, amt_data,
history_vars`` (gcn/models.py:339-347) -- but executes eagerly on one MI355X:

  * features (N x F*dim_s) and every N x d history live in HBM for the whole run (1.12 GB +
    119 MB for Reddit; the box has 288 GB), so the per-step host gather + feed copy of the
    reference (gcn/vrgcn.py:39-47, 3.2 ms/step measured in SURVEY.md §6) becomes a device
    row gather;
  * a minibatch arrives as ONE pinned staging buffer ``[int32 section | fp32 section]``
    (``DevFeed``) holding fields / ffields / scales / labels and the CSR, transposed-CSR and row
    plan of each layer: one H2D copy per step, addressed through ``ops.DevArray`` windows;
  * parameters, gradients and Adam moments are single flat fp32 buffers (one fused update,
    and exactly one RCCL all-reduce per step in the multi-GPU driver, parallel.py);
  * history rows are scattered after the optimizer step, as the reference orders it
    (gcn/models.py:186-194).
"""
import math
import os
from time import time

import numpy as np
import scipy.sparse as sp
import torch

from . import ops
from ._ffi import check, lib as _lib
from .flags import FLAGS
from .layers import (AugmentedDropoutDense, Dense, DetDropoutFC, Dropout, SparseInput)
from .scheduler import PackedBatch, build_plan


from .scheduler import CSR_DESC as _CSR_DESC          # noqa: E402

AGG_PLAN_T = 128       # include/sgcn.h SGCN_AGG_PLAN_T: the aggregator gives a workgroup to a plan segment of the fadj matrix


class _LazyHostFields(object):
    """``cur.host_fields[l]`` for a packed batch without touching NumPy unless somebody asks (only
    the sparse-feature input slice does)."""

    def __init__(self, pb):
        self.pb = pb

    def __getitem__(self, l):
        return self.pb.field(l)


class VariableStore(object):
    """What ``tf.make_template`` provides in the reference (gcn/train.py:115-119): the second
    model instance created through the same template reuses the first one's weights."""

    def __init__(self):
        self.theta = None
        self.layout = None


def make_template(name, func):
    store = VariableStore()

    def create(*args, **kwargs):
        kwargs['_store'] = store
        return func(*args, **kwargs)
    return create


class _EvalCur(object):
    """what Trainer.evaluate reads of `model.cur` after an evaluation step run as a program: the batch's labels"""
    __slots__ = ("labels",)

    def __init__(self, labels):
        self.labels = labels


class DevFeed(object):
    """One minibatch on device."""

    def __init__(self, feed_dict, placeholders, L, cv, device, plan_T=0):
        ph = placeholders
        ints, flts = [], []

        def add_i(a):
            a = np.ascontiguousarray(a, dtype=np.int32).ravel()
            off = sum(x.shape[0] for x in ints)
            pad = (-a.shape[0]) % 4                      # keep every sub-array 16-byte aligned
            ints.append(np.concatenate([a, np.zeros(pad, np.int32)]) if pad else a)
            return (off, a.shape[0])

        def add_f(a):
            a = np.ascontiguousarray(a, dtype=np.float32).ravel()
            off = sum(x.shape[0] for x in flts)
            pad = (-a.shape[0]) % 4
            flts.append(np.concatenate([a, np.zeros(pad, np.float32)]) if pad else a)
            return (off, a.shape[0])

        self.host_fields = [feed_dict[ph['fields'][l]] for l in range(L + 1)]
        f_slots = [add_i(f) for f in self.host_fields]
        labels = np.asarray(feed_dict[ph['labels']], dtype=np.float32)
        lab_slot = add_f(labels)
        sc_slots = [add_f(feed_dict[ph['scales'][l]]) for l in range(L)]
        ff_slots, csr_slots = [], []
        for l in range(L):
            entry = {}
            a = feed_dict[('csr', ph['adj'][l])]
            entry['a'] = self._pack_csr(a, add_i, add_f, plan_T, transpose=True)
            if cv:
                ff_slots.append(add_i(feed_dict[ph['ffields'][l]]))
                p = feed_dict[('csr', ph['fadj'][l])]
                entry['f'] = self._pack_csr(p, add_i, add_f, max(plan_T, AGG_PLAN_T), transpose=False)
            csr_slots.append(entry)

        ibuf = np.concatenate(ints) if ints else np.zeros(0, np.int32)
        fbuf = np.concatenate(flts) if flts else np.zeros(0, np.float32)
        pin = device.type == 'cuda'
        it = torch.from_numpy(ibuf)
        ft = torch.from_numpy(fbuf)
        if pin:
            it, ft = it.pin_memory(), ft.pin_memory()
        self.ibuf = it.to(device, non_blocking=True)
        self.fbuf = ft.to(device, non_blocking=True)
        iv = lambda s: self.ibuf[s[0]:s[0] + s[1]]     # noqa: E731
        fv = lambda s: self.fbuf[s[0]:s[0] + s[1]]     # noqa: E731

        self.fields = [iv(s) for s in f_slots]
        self.labels = fv(lab_slot).view(labels.shape)
        self.scales = [fv(s) for s in sc_slots]
        self.ffields = [iv(s) for s in ff_slots]
        self.adj, self.fadj = [], []
        for entry in csr_slots:
            self.adj.append(self._unpack_csr(entry['a'], iv, fv, device))
            if cv:
                self.fadj.append(self._unpack_csr(entry['f'], iv, fv, device))

    @classmethod
    def from_packed(cls, pb, device):
        """Fast path: the sampler already laid the batch out (sgcn_sched_batch_packed); two H2D
        copies out of the (pinned) staging slot, then views."""
        self = cls.__new__(cls)
        if pb.slot is not None:           # one H2D copy: [int32 section | fp32 section]
            n_i = max(pb.n_i, 1)
            words = pb.slot.buf[:n_i + max(pb.n_f, 1)].to(device, non_blocking=True)
            self.ibuf, self.fbuf = words[:n_i], words[n_i:].view(torch.float32)
        else:
            self.ibuf = torch.from_numpy(pb.ibuf[:max(pb.n_i, 1)]).to(device, non_blocking=True)
            self.fbuf = torch.from_numpy(pb.fbuf[:max(pb.n_f, 1)]).to(device, non_blocking=True)
        if pb.slot is not None and device.type == 'cuda':
            ev = torch.cuda.Event()
            ev.record()
            pb.slot.event = ev                       # the producer waits on it before reusing the slot
        # windows into the two sections, not tensor views (ops.DevArray): address + length is all
        # the kernels need; offsets come from the descriptor table as plain Python ints (pb.m)
        ibuf, fbuf, ip, fp = self.ibuf, self.fbuf, self.ibuf.data_ptr(), self.fbuf.data_ptr()
        DA, m, L = ops.DevArray, pb.m, pb.L
        self.host_fields = _LazyHostFields(pb)
        self.fields = [DA(ibuf, ip, m[4 + 2 * l], m[5 + 2 * l]) for l in range(L + 1)]
        o = pb.o_labels
        lo, lr, lc = m[o], m[o + 1], m[o + 2]
        self.labels = fbuf[lo:lo + lr * lc].view(lr, lc)
        o = pb.o_scales
        self.scales = [DA(fbuf, fp, m[o + 2 * l], m[o + 2 * l + 1]) for l in range(L)]
        o = pb.o_ffields
        self.ffields = [DA(ibuf, ip, m[o + 2 * l], m[o + 2 * l + 1]) for l in range(L)] if pb.cv else []
        ND = _CSR_DESC

        def csr(b):     # descriptor at m[b : b + ND]: rows, cols, nnz, rowptr, col, val, seg, nseg, fix, nfix, nslots
            plan = ops.DevicePlan.__new__(ops.DevicePlan)
            plan.nseg, plan.nfix, plan.nslots = m[b + 7], m[b + 9], m[b + 10]
            plan.seg = DA(ibuf, ip, m[b + 6], 4 * m[b + 7])
            plan.fix = DA(ibuf, ip, m[b + 8], 3 * m[b + 9]) if m[b + 9] else None
            plan.ws, plan.device = None, device
            return ops.DeviceCSR((m[b], m[b + 1]), DA(ibuf, ip, m[b + 3], m[b] + 1), DA(ibuf, ip, m[b + 4], m[b + 2]),
                                 DA(fbuf, fp, m[b + 5], m[b + 2]), plan)
        self.adj, self.fadj = [], []
        base = pb.o_csr
        for l in range(L):
            b = base + 3 * l * ND
            a = csr(b)
            a.transpose = csr(b + ND)
            self.adj.append(a)
            if pb.cv:
                self.fadj.append(csr(b + 2 * ND))
        self.sizes = dict(adj=[m[base + 3 * l * ND + 2] for l in range(L)],
                          fadj=[m[base + (3 * l + 2) * ND + 2] for l in range(L)],
                          fields=[m[5 + 2 * l] for l in range(L + 1)])
        return self

    @staticmethod
    def _pack_csr(h, add_i, add_f, plan_T, transpose):
        seg, fix, nslots = build_plan(h.rowptr, plan_T)
        e = dict(shape=h.shape, rowptr=add_i(h.rowptr), col=add_i(h.col), val=add_f(h.val),
                 seg=add_i(seg), nseg=seg.shape[0], fix=add_i(fix), nfix=fix.shape[0], nslots=nslots)
        if transpose and h.t_rowptr is not None:
            tseg, tfix, tns = build_plan(h.t_rowptr, plan_T)
            e['t'] = dict(shape=(h.shape[1], h.shape[0]), rowptr=add_i(h.t_rowptr), col=add_i(h.t_col),
                          val=add_f(h.t_val), seg=add_i(tseg), nseg=tseg.shape[0], fix=add_i(tfix),
                          nfix=tfix.shape[0], nslots=tns)
        return e

    @staticmethod
    def _unpack_csr(e, iv, fv, device):
        plan = ops.DevicePlan.__new__(ops.DevicePlan)
        plan.nseg, plan.nfix, plan.nslots = e['nseg'], e['nfix'], e['nslots']
        plan.seg = iv(e['seg'])
        plan.fix = iv(e['fix']) if e['nfix'] else None
        plan.ws, plan.device = None, device
        m = ops.DeviceCSR(e['shape'], iv(e['rowptr']), iv(e['col']), fv(e['val']), plan)
        if 't' in e:
            m.transpose = DevFeed._unpack_csr(e['t'], iv, fv, device)
        return m


class Model(object):
    def __init__(self, **kwargs):
        allowed_kwargs = {'name', 'logging', 'multitask', 'is_training', 'device', '_store'}
        for kwarg in kwargs.keys():
            assert kwarg in allowed_kwargs, 'Invalid keyword argument: ' + kwarg
        self.name = kwargs.get('name') or 'model'
        self.logging = kwargs.get('logging', False)
        self.vars = []
        self.placeholders = {}
        self.layers = []
        self.activations = []
        self.inputs = None
        self.outputs = None
        self.loss = 0
        self.accuracy = 0
        self.multitask = kwargs.get('multitask', False)
        self.aggregators = []
        self.is_training = kwargs.get('is_training', True)
        dev = kwargs.get('device')
        if dev is None:
            if not torch.cuda.is_available():
                raise RuntimeError("stochastic_gcn_amd models need a GPU (no CPU fallback)")
            dev = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(dev)
        self._store = kwargs.get('_store') or VariableStore()
        self.dropout = 0.0
        self.cur = None
        self._want_grad = False
        self.grad_hook = None      # parallel.py installs the RCCL all-reduce here
        self.history_hook = None   # parallel.py: all-gather + apply every rank's history rows
        self.history_join = None   # parallel.py: the exchange is asynchronous; every reader of the history joins it first

    # -- reference API --------------------------------------------------------------------
    def _eval_out(self, vec):
        """An evaluation batch's result vector [stats(4) | CE per row | hit per row | classes per row] on its way to the
        host: straight into the caller's pinned buffer (``eval_sink``: a copy-engine transfer in stream order, no kernel --
        the next batch may overwrite the program's arena behind it), else a device copy."""
        sink = self.__dict__.get('eval_sink')
        return sink.push(vec) if sink is not None else vec.clone()

    def join_history(self):
        """Data parallel: the history rows the ranks exchanged behind the last step land now (a no-op otherwise)."""
        if self.history_join is not None:
            self.history_join()

    # The history exchange of a data-parallel step is asynchronous (parallel.DataParallel.sync_history): until it is joined
    # a replica's history lacks the rows of that step -- its own included.  So that no reader can forget the join, the
    # public names ARE the join: ``history`` / ``history_vars`` hand the tensors out behind it (a no-op without a pending
    # exchange).  Only the two places that ISSUE the exchange (update_history, _run_program) go to ``_history`` directly.
    @property
    def history(self):
        self.join_history()
        return self._history

    @history.setter
    def history(self, value):
        self._history = value

    @property
    def history_vars(self):
        self.join_history()
        return self._history_vars

    @history_vars.setter
    def history_vars(self, value):
        self._history_vars = value

    def save(self, sess=None, path=None):
        """Weights + history (gcn/models.py:204-209 saves self.vars + self.history_vars)."""
        self.join_history()
        path = path or "tmp/%s.ckpt.npz" % self.name
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        blob = {"var/" + n: v.detach().cpu().numpy() for n, v in self.named_vars()}
        for i, h in enumerate(self.history_vars):
            blob["history/%d" % i] = h.detach().cpu().numpy()
        np.savez(path, **blob)
        print("Model saved in file: %s" % path)
        return path

    def load(self, sess=None, load_history=False, path=None):
        path = path or "tmp/%s.ckpt.npz" % self.name
        z = np.load(path)
        for n, v in self.named_vars():
            v.copy_(torch.from_numpy(z["var/" + n]).to(self.device))
        if load_history:
            self.join_history()
            for i, h in enumerate(self.history_vars):
                h.copy_(torch.from_numpy(z["history/%d" % i]).to(self.device))
        print("Model restored from file: %s" % path)


class GCN(Model):
    def __init__(self, L, preprocess, placeholders, features, nbr_features, adj, cvd, **kwargs):
        super(GCN, self).__init__(**kwargs)
        self.L = L
        self.preprocess = preprocess
        self.placeholders = placeholders
        self.sparse_input = sp.issparse(features) or isinstance(features, ops.DeviceCSR)
        self.input_dim = features.shape[1]
        self_dim = 0 if FLAGS.normalization == 'gcn' else self.input_dim
        self._set_features(features, nbr_features, self_dim, preprocess and FLAGS.pp_nbr)
        self.adj = adj
        self.cvd = cvd
        self.build()
        self.init_counts()

    def init(self, sess):
        pass

    # ---- feature residency --------------------------------------------------------------
    def _set_features(self, features, nbr_features, self_dim, stack):
        dev = self.device
        if self.sparse_input:
            if stack:                                             # gcn/models.py:235-239
                f = sp.hstack((features[:, :self_dim], nbr_features)).tocsr().astype(np.float32)
            else:
                f = features.tocsr().astype(np.float32)
            f.sort_indices()
            self.features = f
            self.features_dev = ops.DeviceCSR.from_scipy(f, dev, with_plan=False)
            return

        def to_dev(x):
            if isinstance(x, torch.Tensor):
                return x.to(dev)
            return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        if stack:
            nb = to_dev(nbr_features)
            n, fn = nb.shape
            out = torch.empty((n, self_dim + fn), dtype=torch.float32, device=dev)
            if self_dim:
                out[:, :self_dim] = to_dev(features)[:, :self_dim]
            out[:, self_dim:] = nb
            self.features_dev = out
        else:
            self.features_dev = to_dev(features)
        self.features = self.features_dev

    def _preprocess(self):
        if self.preprocess:
            self.L -= 1
        self.agg0_dim = FLAGS.hidden1 if self.preprocess else self.input_dim

    def _build_history(self):
        self.history = []

    def _build_aggregators(self):
        pass

    # ---- graph assembly (gcn/models.py:124-196, :258-337) -------------------------------
    def build(self):
        self.sparse_mm = self.sparse_input
        if self.sparse_input and not self.preprocess:
            print('Warning: we do not support sparse input without pre-processing. Converting to dense...')
            self.features_dev = torch.from_numpy(
                np.asarray(self.features.todense(), dtype=np.float32)).to(self.device)
            self.sparse_mm = False
        self.num_data = self.adj.shape[0]
        lab = self.placeholders['labels']
        self.output_dim = int(lab.shape[1]) if hasattr(lab, 'shape') else int(self.placeholders['labels_dim'])
        self._preprocess()
        self._build_history()
        self._build_aggregators()
        self._build()
        self._allocate_variables()
        self.history_vars = [h[0] for h in self.history]
        if self.is_training:
            self.adam_t = 0
        # dropout sites: key = hash(seed, layer index, step); a Dropout feeding a dense-input Dense
        # layer hands its mask over instead of applying it (layers.Dropped)
        self.dropout_step = 0
        self.dropout_seed = int(FLAGS.seed)       # parallel.DataParallel.attach adds the rank
        from .layers import Dropout as _Dropout, Dense as _Dense
        for i, layer in enumerate(self.layers):
            layer.index = i
            layer.key_fn = (lambda idx=i: ops.dropout_key(self.dropout_seed, idx, self.dropout_step))
            if isinstance(layer, _Dropout):
                layer.noise_key_fn = (lambda idx=i: ops.dropout_key(self.dropout_seed, idx + 4096, self.dropout_step))
            if isinstance(layer, _Dropout) and i + 1 < len(self.layers):
                nxt = self.layers[i + 1]
                layer.fuse_next = isinstance(nxt, _Dense) and not nxt.sparse_inputs
        # nothing upstream of the first parametrised layer needs a gradient
        self._first_param = next((i for i, l in enumerate(self.layers) if l.param_shapes()), len(self.layers))
        if self._first_param < len(self.layers):
            self.layers[self._first_param].need_dx = False

    def _keep_prob(self):
        return 1.0 - self.dropout

    def _build(self):
        if FLAGS.det_dropout and (self.cvd or FLAGS.reverse or self.L == 0):
            # the reference builds these stacks but cannot run them: with --cvd the aggregator reads (mu, var) as (h, mu)
            # (gcn/layers.py:299), with --reverse / one layer a Dense layer or the loss receives the tuple
            raise NotImplementedError("det_dropout with cvd / reverse / a single pre-processed layer has no working form "
                                      "in the reference")
        dim_s = 1 if FLAGS.normalization == 'gcn' else 2
        cnt = 0
        self.layer_comp = []
        kp = self._keep_prob
        if self.preprocess:
            for l in range(FLAGS.num_fc_layers):
                input_dim = self.input_dim * dim_s if l == 0 else FLAGS.hidden1
                sparse_inputs = self.sparse_mm if l == 0 else False
                last_layer = self.L == 0 and l + 1 == FLAGS.num_fc_layers
                output_dim = self.output_dim if last_layer else FLAGS.hidden1
                layer_norm = False if last_layer else FLAGS.layer_norm
                if FLAGS.det_dropout:                                   # gcn/models.py:275-282
                    self.layers.append(DetDropoutFC(kp, input_dim, FLAGS.hidden1, sparse_inputs=sparse_inputs,
                                                    norm=FLAGS.layer_norm, name='dense%d' % cnt))
                elif self.cvd:
                    self.layers.append(AugmentedDropoutDense(kp, input_dim, FLAGS.hidden1,
                                                             sparse_inputs=sparse_inputs,
                                                             norm=FLAGS.layer_norm, name='dense%d' % cnt))
                else:
                    self.layers.append(Dropout(kp, self.cvd, name='dropout_pp%d' % cnt))
                    self.layers.append(Dense(input_dim, output_dim, self.placeholders,
                                             sparse_inputs=sparse_inputs, act=not last_layer,
                                             norm=layer_norm, name='dense%d' % cnt))
                self.layer_comp.append((input_dim * FLAGS.hidden1, 0))
                cnt += 1
        for l in range(self.L):
            self.layers.append(self.aggregators[l])
            for l2 in range(FLAGS.num_fc_layers):
                dim = self.agg0_dim if l == 0 else FLAGS.hidden1
                input_dim = dim * dim_s if l2 == 0 else FLAGS.hidden1
                last_layer = l2 + 1 == FLAGS.num_fc_layers and l + 1 == self.L
                output_dim = self.output_dim if last_layer else FLAGS.hidden1
                layer_norm = False if last_layer else FLAGS.layer_norm
                if FLAGS.det_dropout and l + 1 != self.L:               # gcn/models.py:312-318
                    self.layers.append(DetDropoutFC(kp, input_dim, output_dim, norm=layer_norm, name='dense%d' % cnt))
                elif self.cvd and l + 1 != self.L:
                    self.layers.append(AugmentedDropoutDense(kp, input_dim, output_dim,
                                                             norm=layer_norm, name='dense%d' % cnt))
                else:
                    if not FLAGS.reverse:
                        self.layers.append(Dropout(kp, self.cvd, name='dropout%d' % cnt))
                    self.layers.append(Dense(input_dim, output_dim, self.placeholders,
                                             act=not last_layer, norm=layer_norm, name='dense%d' % cnt))
                    if FLAGS.reverse and not last_layer:
                        self.layers.append(Dropout(kp, self.cvd, name='dropout_r%d' % cnt))
                self.layer_comp.append((input_dim * output_dim, l + 1))
                cnt += 1

    # ---- flat parameter / gradient / Adam buffers ---------------------------------------
    def _allocate_variables(self):
        layout, total = [], 0
        for layer in self.layers:
            for pname, shape, init in layer.param_shapes():
                n = int(np.prod(shape))
                layout.append((layer.name + '/' + pname, shape, init, total, n))
                total += (n + 3) // 4 * 4
        st = self._store
        if st.theta is None:
            st.theta = torch.zeros(total, dtype=torch.float32, device=self.device)
            st.layout = [(n, s) for n, s, _, _, _ in layout]
            gen = torch.Generator(device='cpu')
            gen.manual_seed(int(FLAGS.seed))
            for name, shape, init, off, n in layout:
                if init == 'glorot':         # tf.get_variable default = glorot_uniform
                    lim = float(np.sqrt(6.0 / (shape[0] + shape[1])))
                    w = (torch.rand(shape, generator=gen) * 2 - 1) * lim
                elif init == 'ones':
                    w = torch.ones(shape)
                else:
                    w = torch.zeros(shape)
                st.theta[off:off + n] = w.reshape(-1).to(self.device)
        else:
            assert st.layout == [(n, s) for n, s, _, _, _ in layout], \
                "template re-used with a different variable layout"
        self.theta = st.theta
        self.grad = torch.zeros_like(self.theta)
        self._layout = layout
        by_layer = {l.name: l for l in self.layers}
        for name, shape, _, off, n in layout:
            lname, pname = name.rsplit('/', 1)
            by_layer[lname].vars[pname] = self.theta[off:off + n].view(shape)
            by_layer[lname].grads[pname] = self.grad[off:off + n].view(shape)
        if self.is_training:
            self.adam_m = torch.zeros_like(self.theta)
            self.adam_v = torch.zeros_like(self.theta)
        # weight decay: the vars of the first parametrised layer (gcn/models.py:68-75) are one contiguous
        # range of the flat buffer (a layer's parameters are laid out back to back, wd_vars() first)
        self._wd_range = (0, 0)
        for layer in self.layers:
            if layer.param_shapes():
                spans = [(off, off + n) for name, shape, _, off, n in layout
                         if name.rsplit('/', 1)[0] == layer.name and name.rsplit('/', 1)[1] in layer.wd_vars()]
                lo, hi = min(a for a, _ in spans), max(b for _, b in spans)
                # padding floats between parameters are zero and stay zero (zero gradient), so they add nothing
                assert all(name.rsplit('/', 1)[0] == layer.name and name.rsplit('/', 1)[1] in layer.wd_vars()
                           for name, shape, _, off, n in layout if lo <= off < hi), "wd vars must be contiguous"
                self._wd_range = (lo, hi)
                break
        self.vars = [v for _, v in self.named_vars()]

    def named_vars(self):
        out = []
        by_layer = {l.name: l for l in self.layers}
        for name, shape, _, off, n in self._layout:
            lname, pname = name.rsplit('/', 1)
            out.append((name, by_layer[lname].vars[pname]))
        return out

    def set_params(self, params):
        """Load a {'denseK/weights': ndarray, ...} dict (parity tests share weights with the oracle)."""
        for name, v in self.named_vars():
            v.copy_(torch.from_numpy(np.ascontiguousarray(params[name], dtype=np.float32)).to(self.device))

    def get_params(self):
        return {name: v.detach().cpu().numpy().copy() for name, v in self.named_vars()}

    def get_grads(self):
        by_layer = {l.name: l for l in self.layers}
        return {name: by_layer[name.rsplit('/', 1)[0]].grads[name.rsplit('/', 1)[1]].detach().cpu().numpy().copy()
                for name, _ in self.named_vars()}

    # ---- one step -----------------------------------------------------------------------
    # The epoch counters (gcn/vrgcn.py:50-69) are linear in the minibatches' sizes: the step-program path adds the
    # minibatch's descriptor table to a pending sum (one vector add per step) and the counters take it in when they are read.
    def _counter(name):
        def get(self):
            if self.__dict__.get('_pend_n'):
                self._flush_counts()
            return self.__dict__['_c_' + name]

        def put(self, v):
            if self.__dict__.get('_pend_n'):
                self._flush_counts()
            self.__dict__['_c_' + name] = v
        return property(get, put)
    g_ops, nn_ops, amt_data = _counter('g_ops'), _counter('nn_ops'), _counter('amt_data')
    field_sizes, adj_sizes, fadj_sizes = _counter('field_sizes'), _counter('adj_sizes'), _counter('fadj_sizes')
    del _counter

    def _flush_counts(self):
        m, base, ND = self._pend.tolist(), self._pend_o_csr, _CSR_DESC
        self._pend_n, self._pend = 0, None
        self._count_sizes(dict(adj=[m[base + 3 * l * ND + 2] for l in range(self.L)],
                               fadj=[m[base + (3 * l + 2) * ND + 2] for l in range(self.L)],
                               fields=[m[5 + 2 * l] for l in range(self.L + 1)]))

    def init_counts(self):
        self._pend_n, self._pend = 0, None
        self.run_t = 0
        self.g_t = 0
        self.g_ops = 0
        self.nn_ops = 0
        self.field_sizes = np.zeros(self.L + 1)
        self.adj_sizes = np.zeros(self.L)
        self.fadj_sizes = np.zeros(self.L)
        self.amt_data = 0

    def upload(self, feed_dict):
        """feed-dict -> DevFeed (two H2D copies) and the input feature rows."""
        cv = bool(self._history)        # (structure only: no join, see ``history``)
        if isinstance(feed_dict, PackedBatch):
            cur = DevFeed.from_packed(feed_dict, self.device)
        else:
            cur = DevFeed(feed_dict, self.placeholders, self.L, cv, self.device)
        if FLAGS.det_dropout and cv:
            cur.madj = [self._madj(feed_dict, l) for l in range(self.L)]
        f0 = cur.fields[0]
        if self.sparse_input and self.sparse_mm:
            sl = ops.csr_slice(self.features_dev, cur.host_fields[0], rows_dev=f0, with_coo_rows=True)
            cur.inputs = SparseInput(sl)
        else:       # gathered by the first dense layer's GEMMs, or on demand (ops.GatheredRows)
            cur.inputs = ops.GatheredRows(self.features_dev, f0)
        return cur

    def _madj(self, feed_dict, l):
        """The det-dropout aggregator's third matrix (placeholders['madj'], gcn/train.py:92): the minibatch adjacency's
        pattern carrying the sampler's medg weights (gcn/scheduler.cpp:164), with its transpose for the backward."""
        if isinstance(feed_dict, PackedBatch):
            a, w = feed_dict.csr(l, 0), feed_dict._f(*feed_dict._medg[l])
        else:
            a, w = feed_dict[('csr', self.placeholders['adj'][l])], feed_dict[self.placeholders['madj'][l]][1]
        m = sp.csr_matrix((np.asarray(w, np.float32), np.asarray(a.col), np.asarray(a.rowptr)), shape=tuple(a.shape))
        return ops.DeviceCSR.from_scipy(m, self.device, with_plan=True, with_transpose=True)

    def forward(self, cur):
        self.cur = cur
        self.activations = [cur.inputs]
        for layer in self.layers:
            self.activations.append(layer(self.activations[-1]))
        self.outputs = self.activations[-1]
        return self.outputs

    def loss_and_grad(self, labels):
        """gcn/models.py:68-94.  Returns (loss, accuracy, pred, dlogits) as device tensors; the weight-decay
        term of the first parametrised layer is added to the loss slot by sgcn_l2_penalty_f32."""
        z = self.outputs
        want_grad = self.is_training or self._want_grad
        want_pred = not self.is_training or self._want_grad
        ce = ops.sigmoid_ce if self.multitask else ops.softmax_ce
        stats, dlogits, pred = ce(z, labels, want_grad=want_grad, want_pred=want_pred)
        # the rows' classes for the F1 scores (sgcn_softmax_ce_f32: argmax(pred) + 4096 * argmax(labels)), single-label only
        n = int(z.shape[0])
        self.eval_classes = stats[4 + 2 * n:4 + 3 * n] if (want_pred and not self.multitask) else None
        if FLAGS.weight_decay and self._wd_range[1] > self._wd_range[0]:
            ops.l2_penalty(self.theta, self._wd_range[0], self._wd_range[1], FLAGS.weight_decay, loss=stats[2:3])
        # the snapshot Trainer.evaluate reads is taken BEHIND the weight-decay term (gcn/models.py:75: the reported loss
        # includes it), as the step program's L2_PENALTY into stats[2] does
        if self.__dict__.get('eval_light') and self.eval_classes is not None:
            self.eval_vec, self.eval_rows = self._eval_out(stats), n      # (Trainer.evaluate: the vector the step program hands out)
        return stats[2], stats[3], pred, dlogits

    def backward(self, dlogits):
        self.grad.zero_()
        g = dlogits
        for layer in reversed(self.layers[self._first_param:]):
            g = layer.backward(g)
        if FLAGS.weight_decay and self._wd_range[1] > self._wd_range[0]:
            ops.l2_penalty(self.theta, self._wd_range[0], self._wd_range[1], FLAGS.weight_decay, grad=self.grad)
        return self.grad

    def adam_step(self):
        """tf.train.AdamOptimizer(lr, beta1, beta2, eps=1e-8) on the flat buffers."""
        self.adam_t += 1
        b1, b2 = float(FLAGS.beta1), float(FLAGS.beta2)
        ops.adam_step(self.theta, self.grad, self.adam_m, self.adam_v, self._adam_lr(self.adam_t), b1, b2, 1e-8)

    @staticmethod
    def _adam_lr(t):
        """the bias-corrected step size of Adam's step t (both step paths: the same double, rounded to fp32 by the kernel)"""
        b1, b2 = float(FLAGS.beta1), float(FLAGS.beta2)
        return float(FLAGS.learning_rate) * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)

    def update_history(self, cur):
        """tf.scatter_update(history, fields[l], new_history) (gcn/models.py:160-166)."""
        for l in range(self.L):
            agg = self.aggregators[l]
            nh = getattr(agg, 'new_history', None)
            if nh is not None and self._history:
                for h, v in zip(self._history[l], nh):      # (issuing side: no join, see ``history``)
                    if self.history_hook is not None:
                        self.history_hook(h, cur.fields[l], v, ops.scatter_rows)
                    else:
                        ops.scatter_rows(h, cur.fields[l], v)

    def _count(self, feed_dict):
        raise NotImplementedError

    def get_data(self, feed_dict):
        cur = self.upload(feed_dict)
        if isinstance(feed_dict, PackedBatch):
            self._count_sizes(cur.sizes)
        else:
            self._count(feed_dict)
        return cur

    def _count_sizes(self, sizes):
        """The epoch counters of gcn/vrgcn.py:50-69 / gcn/plaingcn.py:41-50 from batch sizes."""
        cv = bool(self._history)        # (structure only: no join, see ``history``)
        for l in range(self.L):
            dim = self.agg0_dim if l == 0 else FLAGS.hidden1
            g_ops = ((sizes['fadj'][l] if cv else 0) + sizes['adj'][l]) * dim * 4
            if self.cvd and cv:
                g_ops *= 2
            self.g_ops += g_ops
            self.adj_sizes[l] += sizes['adj'][l]
            if cv:
                self.fadj_sizes[l] += sizes['fadj'][l]
            self.amt_data += sizes['adj'][l]
        for l in range(self.L + 1):
            self.field_sizes[l] += sizes['fields'][l]
        for c, l in self.layer_comp:
            nn_ops = c * sizes['fields'][l] * 4
            if self.cvd and cv:
                nn_ops *= 2
            self.nn_ops += nn_ops

    # ---- the step as one call of the native launch loop (step_program.py) -----------------------------
    def _program(self, feed_dict, dropout):
        """The compiled step program for this minibatch, or None (eager path): packed batches only, a
        supported layer stack, and a minibatch that fits the program's buffers."""
        if not (FLAGS.native_step and isinstance(feed_dict, PackedBatch)):
            return None
        # a program bakes in raw addresses (weights, gradients, Adam moments, history, features) and whether the
        # history update is local: all of that is part of its identity, so a tensor re-allocated or a hook
        # attached after the first step builds a NEW program instead of leaving a stale one in use
        key = (round(float(dropout), 9), self.history_hook is None, int(getattr(self, 'native_coll', 0) or 0),
               bool(getattr(getattr(self, '_par', None), 'exchange_overlap', False)),
               self.theta.data_ptr(), self.grad.data_ptr(),
               self.adam_m.data_ptr() if self.is_training else 0, self.adam_v.data_ptr() if self.is_training else 0,
               self.features_dev.data_ptr() if isinstance(self.features_dev, torch.Tensor) else 0,
               tuple(h.data_ptr() for hs in self._history for h in hs),
               bool(FLAGS.group_dw), bool(FLAGS.lean_sync), bool(FLAGS.agg_overlap))
        progs = self.__dict__.setdefault('_programs', {})
        if key not in progs:
            while len(progs) >= 2:         # superseded programs (arenas of up to 2 GB each) go: oldest first
                progs.pop(next(iter(progs)))
            from .step_program import StepProgram, Unsupported
            try:
                progs[key] = StepProgram(self, dropout)
            except Unsupported as e:
                progs[key] = None
                self._program_note = str(e)
        return progs[key]          # (whether THIS minibatch fits its buffers: StepProgram.fill, in _run_program)

    def stage(self, pb):
        """Start the H2D copy of a packed minibatch that lives in a pinned staging slot -- ONE copy [int32 section | fp32
        section] on a COPY stream: the host runs a step or two ahead of the GPU, so the next minibatch crosses PCIe while the
        current step computes instead of queueing behind it (the copy is ~1 MB: 20-25 us of an otherwise idle compute stream
        per step).  run_one_step calls it itself; a loop that knows its next batch calls it one batch EARLY (train.py), so
        that the copy is complete, and known to be, when the step that reads it is queued."""
        if pb.slot is None:
            return None
        if getattr(pb, '_staged', None) is not None and getattr(pb, '_ring_owner', None) is self:
            return pb._staged
        # only the step-program path reads the staged copy: a model that steps eagerly (sparse input features, det-dropout,
        # native_step off) would copy every batch twice and keep a ring of dead device buffers
        drop = float(getattr(pb, 'dropout', 0.0) or 0.0) if self.is_training else 0.0
        if self._program(pb, drop) is None:
            return None
        dev = self.device
        n_i = max(pb.n_i, 1)
        cs = self.__dict__.get('_copy_stream')
        if cs is None:
            cs = self._copy_stream = torch.cuda.Stream(device=dev)
            self._copy_stream_h = cs.cuda_stream
        nw = n_i + max(pb.n_f, 1)
        words = self._ring_buffer(pb, nw, cs)
        pb._words_ptr = words.data_ptr()
        check(_lib.sgcn_copy_h2d_async(pb._words_ptr, pb.slot.ptr, 4 * nw, self._copy_stream_h))
        ev = torch.cuda.Event()
        ev.record(cs)
        pb.slot.event = ev                # the producer waits on it before reusing the slot
        pb._staged = (words, ev)
        return pb._staged

    # The staged batches' device buffers are a RING this model owns, not allocations: a tensor that is allocated on the
    # copy stream and read on the step's stream makes the caching allocator record an event on the step's stream when it is
    # freed -- a marker packet between two steps' kernels, 1.5-3 us of every step (0.0408 -> 0.0402 s per Reddit epoch, A/B in one box).  The
    # ring needs the step's stream to say "done" only once per _RING_GROUP steps: the buffers of a group are written again
    # _RING_GROUPS groups later (128 batches, ~100 MB for Reddit: a launching thread that far ahead of the GPU is held back
    # by its copies; with 48 the hold set in every 16 steps and the epoch was 1-2 % slower, A/B in one box), and the copy
    # stream (never the step's) waits for the event recorded behind the group's last
    # step.
    _RING_GROUP, _RING_GROUPS = 16, 8

    def _ring_buffer(self, pb, nw, cs):
        G, NG = self._RING_GROUP, self._RING_GROUPS
        ring = self.__dict__.get('_ring')
        if ring is None:
            ring = self._ring = dict(bufs=[None] * (G * NG), n=0, done={})
        i = ring['n']
        ring['n'] = i + 1
        if i % G == 0 and i >= G * NG:
            main = torch.cuda.current_stream()
            g = i // G - NG                       # the group whose buffers this one takes over
            ev = ring['done'].pop(g, None)
            for stale in [k for k in ring['done'] if k < g]:
                del ring['done'][stale]
            if ev is not None:
                cs.wait_event(ev)
            else:                                 # a batch of that group was staged but never run: everything queued so far
                cs.wait_stream(main)
        buf = ring['bufs'][i % (G * NG)]
        if buf is None or buf.numel() < nw:
            if buf is not None:                   # steps in flight may still read the smaller one
                buf.record_stream(cs)
            buf = ring['bufs'][i % (G * NG)] = torch.empty(nw + nw // 4, dtype=pb.slot.buf.dtype, device=self.device)
            cs.wait_stream(torch.cuda.current_stream())                  # (allocated on the step's stream: ordered behind whatever used the block)
        pb._ring_i, pb._ring_owner = i, self
        return buf              # (the whole buffer: the minibatch is its first nw words)

    def _ring_step_queued(self, pb):
        """behind the step that read batch `pb`'s ring buffer: the last step of a group marks the group reusable"""
        i = getattr(pb, '_ring_i', None)
        if i is not None and i % self._RING_GROUP == self._RING_GROUP - 1:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._ring['done'][i // self._RING_GROUP] = ev

    def _run_program(self, prog, pb, sync):
        """One step as a program; None when the minibatch does not fit it (the caller then runs it layer by layer)."""
        t = time()
        dev = self.device
        n_i = max(pb.n_i, 1)
        main = torch.cuda.current_stream()
        if pb.slot is not None:
            words, ev = self.stage(pb)
            if self._ring['n'] - pb._ring_i > self._RING_GROUP * (self._RING_GROUPS - 1):
                raise RuntimeError("this minibatch was staged %d batches ago: its device buffer has been handed on "
                                   "(Model.stage keeps %d batches)" % (self._ring['n'] - pb._ring_i,
                                                                       self._RING_GROUP * (self._RING_GROUPS - 1)))
            # a copy the caller staged one batch ahead (stage()) has completed by the time its step is queued: the step's
            # queue then needs no barrier on the copy engine's signal (3-10 us per step, and most of the step-to-step jitter).
            # When it has not -- a launching thread two ring groups ahead of the GPU, its copies held back by the ring's
            # events -- it is the HOST that waits: that is the back-pressure of the loop, and the queue (dozens of steps
            # deep at that point) still gets no barrier
            if not ev.query():
                ev.synchronize()
            ip = pb._words_ptr
            fp = ip + 4 * n_i
            live = words
        else:
            ib = torch.from_numpy(pb.ibuf[:n_i]).to(dev, non_blocking=True)
            fb = torch.from_numpy(pb.fbuf[:max(pb.n_f, 1)]).to(dev, non_blocking=True)
            ip, fp = ib.data_ptr(), fb.data_ptr()
            live = (ib, fb)
        # sizes, addresses, dropout keys and the step size into the program's slot table (one foreign call, which also
        # checks the minibatch against the program's buffers)
        lr_t = self._adam_lr(self.adam_t + 1) if self.is_training else 0.0
        if not prog.fill(pb, ip, fp, self.dropout_step, lr_t):
            return None
        self._live_batch = live
        if self.is_training:
            self.adam_t += 1
        if self.__dict__.get('_pend') is None:
            self._pend, self._pend_o_csr = pb.meta.copy(), pb.o_csr
        else:
            self._pend += pb.meta
        self._pend_n = self.__dict__.get('_pend_n', 0) + 1
        self.g_t += time() - t
        t = time()
        stream = main.cuda_stream
        if not self.is_training:
            # evaluation (gcn/train.py:133-160): forward + loss + prediction + the test model's history scatter as ONE
            # foreign call.  pred is copied out of the program's arena (the next batch overwrites it); the labels stay a
            # view of this batch's own staging copy
            prog.run('all', stream)
            m = pb.m
            nL, c = m[5 + 2 * self.L], int(prog.pred.cols)
            if self.__dict__.get('eval_light') and not self.multitask:
                # Trainer.evaluate's form: ONE copy out of the arena per batch -- [stats(4) | CE per row | hit per row |
                # class indices per row] -- instead of prediction, labels and statistics as three tensors
                so = prog.stats_off
                self.eval_vec, self.eval_rows = self._eval_out(prog.arena[so:so + 4 + 3 * nL]), nL
                if pb.slot is not None:
                    self._ring_step_queued(pb)
                self.cur = self.eval_classes = None
                self.dropout_step += 1
                self.run_t += time() - t
                return [None, None, None]
            pred = prog.tensor_of(prog.pred, nL).clone()
            off, r_, c_ = m[pb.o_labels], m[pb.o_labels + 1], m[pb.o_labels + 2]
            if pb.slot is not None:           # (a copy: the ring buffer is written again a few dozen batches on)
                src = words[n_i + off:n_i + off + r_ * c_].clone()
                self._ring_step_queued(pb)
            else:
                src = fb[off:off + r_ * c_].view(torch.int32)
            self.cur = _EvalCur(src.view(torch.float32).view(r_, c_))
            so = prog.stats_off + 4 + 2 * nL
            self.eval_classes = None if self.multitask else prog.arena[so:so + nL].clone()
            self.dropout_step += 1
            loss, acc = prog.loss_t, prog.acc_t
            if sync:
                loss, acc, pred = float(loss), float(acc), pred.cpu().numpy()
            self.run_t += time() - t
            return [loss, acc, pred]
        if (self.grad_hook is None and self.history_hook is None) or prog.native_world:
            prog.run('all', stream)       # (data parallel on the library's own communicator: the collectives are ops of the program)
        else:                                 # data parallel: collectives between the program's phases
            prog.run('fb', stream)
            if self.grad_hook is not None:
                self.grad_hook(self.grad)
            prog.run('opt', stream)
            if self.history_hook is None:
                prog.run('hist', stream)
            else:
                m = pb.m
                for l, nh in prog.new_history.items():
                    n = m[5 + 2 * l]
                    idx = words[m[4 + 2 * l]:m[4 + 2 * l] + n] if pb.slot is not None else ib[m[4 + 2 * l]:m[4 + 2 * l] + n]
                    self.history_hook(self._history[l][0], idx, prog.tensor_of(nh, n), ops.scatter_rows)
        if pb.slot is not None:
            self._ring_step_queued(pb)
        self.dropout_step += 1
        loss, acc = prog.loss_t, prog.acc_t
        if sync:
            loss, acc = float(loss), float(acc)
        self.run_t += time() - t
        return [None, loss, acc]

    def run_one_step(self, sess, feed_dict, sync=True):
        """One step (gcn/vrgcn.py:72-84): [_, loss, acc] in training, [loss, acc, pred] in evaluation.

        ``sync=False`` leaves loss / acc on the device.  On the step-program path they are VIEWS of the
        program's statistics slot, which the next step overwrites: read (or ``.clone()``) them before the
        next ``run_one_step`` -- the Trainer reads the last step's only (tests/test_step_program_gpu.py
        asserts the aliasing).  The eager path returns fresh tensors."""
        self.join_history()                   # (the previous step's exchange: its first reader is this step's aggregator)
        if isinstance(feed_dict, PackedBatch):
            drop = float(getattr(feed_dict, 'dropout', 0.0) or 0.0) if self.is_training else 0.0
            prog = self._program(feed_dict, drop)
            if prog is not None:
                self.dropout = prog.dropout
                out = self._run_program(prog, feed_dict, sync)
                if out is not None:
                    return out
        t = time()
        if not self.is_training:
            self.dropout = 0.0
        elif isinstance(feed_dict, PackedBatch):
            self.dropout = float(getattr(feed_dict, 'dropout', 0.0) or 0.0)
        else:
            self.dropout = float(feed_dict.get(self.placeholders['dropout'], 0.0))
        cur = self.get_data(feed_dict)
        self.g_t += time() - t

        t = time()
        ops.pin_stream()
        try:
            self.forward(cur)
            loss, acc, pred, dlogits = self.loss_and_grad(cur.labels)
            if self.is_training:
                self.backward(dlogits)
                if self.grad_hook is not None:
                    self.grad_hook(self.grad)
                self.adam_step()
            self.update_history(cur)
        finally:
            ops.unpin_stream()
        self.dropout_step += 1          # the next step draws fresh masks at every dropout site
        if sync:
            loss, acc = float(loss), float(acc)
            outs = [None, loss, acc] if self.is_training else [loss, acc, pred.cpu().numpy()]
        else:
            outs = [None, loss, acc] if self.is_training else [loss, acc, pred]
        self.run_t += time() - t
        return outs

    def get_pred_and_grad(self, sess, feed_dict):
        """Prediction and the gradient of the loss wrt the first variable (gcn/models.py:196),
        without touching weights or history."""
        self.join_history()
        self.dropout = float(feed_dict.get(self.placeholders['dropout'], 0.0))
        cur = self.get_data(feed_dict)
        self.forward(cur)
        self._want_grad = True
        try:
            loss, acc, pred, dlogits = self.loss_and_grad(cur.labels)
        finally:
            self._want_grad = False
        self.backward(dlogits)
        first = self.named_vars()[0][0]
        self.dropout_step += 1
        return pred.cpu().numpy(), [self.get_grads()[first]]
