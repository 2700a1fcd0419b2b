"""oracle/model_np.py -- TEST INFRASTRUCTURE ONLY (the parity checker; never the product).

NumPy fp32 restatement of the reference model stack around the hot path, forward AND
backward, so that layer activations, loss, gradients, the Adam step and the history update
of the HIP path can be compared on the same seeded inputs:

  layer stack assembly   gcn/models.py:258-337 (GCN._build), flag effects gcn/train.py:85-87
  PP input features      gcn/models.py:231-241
  AugmentedDropoutDense  gcn/layers.py:365-412      Dense / MyLayerNorm  gcn/layers.py:87-138
  Dropout                gcn/layers.py:415-433      sparse_dropout       gcn/layers.py:23-28
  Plain / VR aggregator  gcn/layers.py:214-257, 282-362   (oracle/oracle_np.py)
  history alloc / update gcn/vrgcn.py:23-36, gcn/models.py:160-166,186-194
  loss / accuracy / Adam gcn/models.py:50-51,68-94,186-196

PARITY UNPINNED: these are TensorFlow-1 ops in the reference (un-vendored, un-pinned), and
the reference has no test that pins their numbers; tests/test_model_oracle.py therefore
checks this restatement's backward against float64 torch.autograd of the same forward.
Dropout masks are inputs (a `masks` callable) so that the device path and the oracle can be
fed the identical randomness.
"""
import numpy as np
import scipy.sparse as sp

from . import oracle_np as onp
from . import det_np

f32 = np.float32

DEFAULT_FLAGS = dict(   # gcn/train.py:25-67
    learning_rate=0.01, hidden1=32, dropout=0.5, weight_decay=5e-4, degree=20, batch_size=1000,
    cv=False, preprocess=True, num_layers=2, num_fc_layers=1, beta1=0.9, beta2=0.999,
    normalization='gcn', layer_norm=False, det_dropout=False, cvd=False, reverse=False,
    pp_nbr=True)


def make_flags(**kw):
    f = dict(DEFAULT_FLAGS)
    f.update(kw)
    return f


def layer_specs(flags, L_after_pp, preprocess, cvd, input_dim, output_dim, sparse_mm):
    """The layer list GCN._build assembles (gcn/models.py:258-337), as plain tuples.
    L_after_pp = number of aggregator layers (model.L after _preprocess, gcn/models.py:251-253)."""
    specs = []
    dim_s = 1 if flags['normalization'] == 'gcn' else 2
    H, nfc = flags['hidden1'], flags['num_fc_layers']
    agg0_dim = H if preprocess else input_dim
    cnt = 0
    if preprocess:
        for l in range(nfc):
            in_dim = input_dim * dim_s if l == 0 else H
            sparse_in = sparse_mm if l == 0 else False
            last = L_after_pp == 0 and l + 1 == nfc
            out_dim = output_dim if last else H
            if flags['det_dropout']:                                # gcn/models.py:275-282
                specs.append(('det', 'dense%d' % cnt, in_dim, H, sparse_in, flags['layer_norm']))
            elif cvd:
                specs.append(('add', 'dense%d' % cnt, in_dim, H, sparse_in, flags['layer_norm']))
            else:
                specs.append(('dropout',))
                specs.append(('dense', 'dense%d' % cnt, in_dim, out_dim, sparse_in, not last,
                              False if last else flags['layer_norm']))
            cnt += 1
    for l in range(L_after_pp):
        specs.append(('agg', l))
        for l2 in range(nfc):
            dim = agg0_dim if l == 0 else H
            in_dim = dim * dim_s if l2 == 0 else H
            last = l2 + 1 == nfc and l + 1 == L_after_pp
            out_dim = output_dim if last else H
            norm = False if last else flags['layer_norm']
            if flags['det_dropout'] and l + 1 != L_after_pp:        # gcn/models.py:312-318
                specs.append(('det', 'dense%d' % cnt, in_dim, out_dim, False, norm))
            elif cvd and l + 1 != L_after_pp:
                specs.append(('add', 'dense%d' % cnt, in_dim, out_dim, False, norm))
            else:
                if not flags['reverse']:
                    specs.append(('dropout',))
                specs.append(('dense', 'dense%d' % cnt, in_dim, out_dim, False, not last, norm))
                if flags['reverse'] and not last:
                    specs.append(('dropout',))
            cnt += 1
    return specs


def init_params(specs, seed):
    """glorot-uniform weights (tf.get_variable default, gcn/inits.py:10-12), zeros/ones LN."""
    rng = np.random.RandomState(seed)
    params = {}
    for s in specs:
        if s[0] in ('add', 'dense', 'det'):
            name, fin, fout = s[1], s[2], s[3]
            lim = np.sqrt(6.0 / (fin + fout))
            params[name + '/weights'] = rng.uniform(-lim, lim, (fin, fout)).astype(f32)
            norm = s[6] if s[0] == 'dense' else s[5]
            if norm:
                params[name + '/offset'] = np.zeros((1, fout), f32)
                params[name + '/scale'] = np.ones((1, fout), f32)
    return params


# ---- primitive ops with explicit backward -----------------------------------------------------
def layer_norm_fwd(x, offset, scale, eps=1e-9):
    """MyLayerNorm2 = tf.nn.moments(axes=[1]) + tf.nn.batch_normalization (gcn/layers.py:95-97)."""
    mean = x.mean(axis=1, keepdims=True, dtype=f32)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True, dtype=f32)
    rstd = (1.0 / np.sqrt(var + f32(eps))).astype(f32)
    xhat = ((x - mean) * rstd).astype(f32)
    return (xhat * scale + offset).astype(f32), (xhat, rstd)


def layer_norm_bwd(dy, ctx, scale):
    xhat, rstd = ctx
    dscale = (dy * xhat).sum(axis=0, keepdims=True, dtype=f32)
    doffset = dy.sum(axis=0, keepdims=True, dtype=f32)
    dxhat = (dy * scale).astype(f32)
    dx = rstd * (dxhat - dxhat.mean(axis=1, keepdims=True, dtype=f32)
                 - xhat * (dxhat * xhat).mean(axis=1, keepdims=True, dtype=f32))
    return dx.astype(f32), doffset, dscale


# ---- the product's counter-based dropout masks, restated (include/sgcn.h sgcn_dropout_t) ---------
def _fmix32(h):
    """murmur3 finaliser on uint64 arrays holding 32-bit values."""
    M = np.uint64(0xFFFFFFFF)
    h = h & M
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & M
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & M
    h = h ^ (h >> np.uint64(16))
    return h


def dropout_key(seed, layer_index, step):
    f = lambda v: int(_fmix32(np.array([v & 0xFFFFFFFF], dtype=np.uint64))[0])      # noqa: E731
    return f(f(int(seed) * 0x9E3779B1 + int(layer_index) * 0x85EBCA77 + 0x27D4EB2F) + int(step) * 0xC2B2AE3D)


def hash_mask(key, shape, keep):
    """{0,1} mask of an activation of `shape` (row-major element index) for one dropout site."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    h = _fmix32((idx * np.uint64(0x9E3779B1) + np.uint64(key)) & np.uint64(0xFFFFFFFF))
    t = float(np.float32(keep)) * 4294967296.0
    thr = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    return (h < np.uint64(thr)).astype(f32).reshape(shape)


class HashMasks(object):
    """`masks(tag, shape)` callback for Model.forward that replays the product's masks: the key
    of layer i at a given step is dropout_key(seed, i, step); tags are 'L<i>'."""

    def __init__(self, seed, step, keep):
        self.seed, self.step, self.keep, self.calls = seed, step, keep, 0

    def __call__(self, tag, shape):
        self.calls += 1
        return hash_mask(dropout_key(self.seed, int(tag[1:]), self.step), tuple(shape), self.keep)

    def noise(self, tag, shape):
        """N(0, 1) of the Gaussian re-sampling in front of a det-dropout model's last Dropout (layer index + 4096
        keeps its stream apart from the layer's dropout mask)."""
        from . import det_np
        return det_np.gauss_noise(dropout_key(self.seed, int(tag[1:]) + 4096, self.step), tuple(shape))


def dropout_fwd(x, keep_prob, mask):
    """tf.nn.dropout: x * mask / keep_prob with mask in {0,1}."""
    if mask is None:
        return x
    return (x * (mask * f32(1.0 / keep_prob))).astype(f32)


def sparse_dropout(x_csr, keep_prob, mask):
    """sparse_dropout (gcn/layers.py:23-28): tf.sparse_retain by the mask, times 1/keep.
    The mask is indexed in CSR storage order (row-major = tf.sparse_reorder order,
    gcn/models.py:127)."""
    if mask is None:
        return x_csr
    keep = mask.astype(bool)
    rows = np.repeat(np.arange(x_csr.shape[0]), np.diff(x_csr.indptr))
    cnt = np.bincount(rows[keep], minlength=x_csr.shape[0])
    out = sp.csr_matrix(x_csr.shape, dtype=f32)
    out.indptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    out.indices = x_csr.indices[keep].astype(np.int32)
    out.data = (x_csr.data[keep] * f32(1.0 / keep_prob)).astype(f32)
    return out


class Model(object):
    """Forward/backward/update of one model instance (train or test) in NumPy fp32."""

    def __init__(self, flags, num_layers, preprocess, cvd, cv, features, nbr_features, num_data,
                 output_dim, params, multitask=False, is_training=True):
        self.flags, self.cvd, self.cv = flags, bool(cvd), bool(cv)
        self.preprocess, self.multitask, self.is_training = preprocess, multitask, is_training
        self.sparse_input = sp.issparse(features)
        self.input_dim = features.shape[1]
        self_dim = 0 if flags['normalization'] == 'gcn' else self.input_dim
        if preprocess and flags['pp_nbr']:                         # gcn/models.py:235-239
            if self.sparse_input:
                self.features = sp.hstack((features[:, :self_dim], nbr_features)).tocsr().astype(f32)
            else:
                self.features = np.hstack((features[:, :self_dim], nbr_features)).astype(f32)
        else:
            self.features = features
        self.L = num_layers - 1 if preprocess else num_layers      # gcn/models.py:251-253
        self.sparse_mm = self.sparse_input
        if self.sparse_input and not preprocess:                   # gcn/models.py:128-133
            self.features = np.asarray(self.features.todense(), dtype=f32)
            self.sparse_mm = False
        self.specs = layer_specs(flags, self.L, preprocess, self.cvd, self.input_dim, output_dim,
                                 self.sparse_mm)
        self.params = params
        H = flags['hidden1']
        agg0 = H if preprocess else self.input_dim
        # zero-initialised N x dims history per aggregator layer (gcn/vrgcn.py:23-36)
        self.history = [np.zeros((num_data, agg0 if i == 0 else H), f32) for i in range(self.L)] \
            if self.cv else []
        # det-dropout keeps (mean, variance) histories: n_history = 2 (gcn/vrgcn.py:28)
        self.history_var = [np.zeros_like(h) for h in self.history] if flags['det_dropout'] else []
        if flags['det_dropout'] and (self.cvd or flags['reverse'] or self.sparse_mm or self.L == 0):
            raise ValueError("det_dropout: the reference has no working cvd / reverse / sparse-input / one-layer form")
        self.adam_t = 0
        self.adam_m = {k: np.zeros_like(v) for k, v in params.items()}
        self.adam_v = {k: np.zeros_like(v) for k, v in params.items()}

    # ---- forward --------------------------------------------------------------------------
    def forward(self, feed, ph, dropout, masks):
        """feed: scheduler feed-dict; masks(tag, shape) -> {0,1} float array or None."""
        fl = self.flags
        keep = 1.0 - dropout
        concat = fl['normalization'] != 'gcn'
        f0 = feed[ph['fields'][0]]
        act = self.features[f0].tocsr() if sp.issparse(self.features) else self.features[f0]
        tape, acts, new_hist = [], [], {}
        for li, s in enumerate(self.specs):
            kind = s[0]
            if kind == 'add':
                _, name, fin, fout, sparse_in, norm = s
                W = self.params[name + '/weights']
                xin, muin = act if isinstance(act, tuple) else (act, act)
                if sparse_in:
                    m = masks('L%d' % li, (xin.nnz,)) if dropout > 0 else None
                    xd = sparse_dropout(xin, keep, m)
                    xs = onp.spmm(xd.indptr, xd.indices, xd.data, W) if xd.nnz else np.zeros((xd.shape[0], fout), f32)
                    mus = onp.spmm(xin.indptr, xin.indices, xin.data, W) if xin.nnz else np.zeros((xin.shape[0], fout), f32)
                else:
                    m = masks('L%d' % li, xin.shape) if dropout > 0 else None
                    xd = dropout_fwd(xin, keep, m)
                    xs = (xd @ W).astype(f32)
                    mus = (muin @ W).astype(f32)
                ctx = None
                if norm:
                    off, sc = self.params[name + '/offset'], self.params[name + '/scale']
                    xs_n, ctx = layer_norm_fwd(xs, off, sc)
                    mus_n, _ = layer_norm_fwd(mus, off, sc)
                else:
                    xs_n, mus_n = xs, mus
                out = (np.maximum(xs_n, 0).astype(f32), np.maximum(mus_n, 0).astype(f32))
                tape.append(('add', s, xd, m, keep, ctx, xs_n))
                act = out
            elif kind == 'det':                                    # DetDropoutFC, gcn/layers.py:141-202
                _, name, fin, fout, sparse_in, norm = s
                off = self.params[name + '/offset'] if norm else None
                sc = self.params[name + '/scale'] if norm else None
                act, ctx = det_np.fc_fwd(act, self.params[name + '/weights'], off, sc, keep)
                tape.append(('det', s, ctx))
            elif kind == 'dropout' and isinstance(act, tuple) and not self.cvd:   # gcn/layers.py:425-428
                mu_, var_ = act
                eps_ = masks.noise('L%d' % li, mu_.shape)
                x_ = det_np.sample_fwd(mu_, var_, eps_)
                m = masks('L%d' % li, x_.shape) if dropout > 0 else None
                act = dropout_fwd(x_, keep, m)
                tape.append(('gauss', m, keep, var_, eps_))
            elif kind == 'dropout':
                if self.cvd and isinstance(act, tuple):            # gcn/layers.py:423-425
                    h = act[0]
                    m = masks('L%d' % li, h.shape) if dropout > 0 else None
                    act = dropout_fwd(h, keep, m)
                    tape.append(('dropout', m, keep, True))
                elif sp.issparse(act):
                    m = masks('L%d' % li, (act.nnz,)) if dropout > 0 else None
                    act = sparse_dropout(act, keep, m)
                    tape.append(('dropout', None, keep, False))    # no grad wrt sparse input
                else:
                    m = masks('L%d' % li, act.shape) if dropout > 0 else None
                    act = dropout_fwd(act, keep, m)
                    tape.append(('dropout', m, keep, False))
            elif kind == 'dense':
                _, name, fin, fout, sparse_in, relu, norm = s
                W = self.params[name + '/weights']
                if sparse_in:
                    y = onp.spmm(act.indptr, act.indices, act.data, W) if act.nnz else np.zeros((act.shape[0], fout), f32)
                else:
                    y = (act @ W).astype(f32)
                ctx = None
                if norm:                                           # MyLayerNorm, own offset/scale
                    y, ctx = layer_norm_fwd(y, self.params[name + '/offset'], self.params[name + '/scale'])
                pre = y
                if relu:
                    y = np.maximum(y, 0).astype(f32)
                tape.append(('dense', s, act, ctx, pre))
                act = y
            elif kind == 'agg':
                l = s[1]
                adj = onp.coo_to_csr(feed[ph['adj'][l]])
                if isinstance(act, tuple) and not self.cvd:        # (mu, var): gcn/layers.py:236-248, 320-349
                    if self.cv:
                        fadj = onp.coo_to_csr(feed[ph['fadj'][l]])
                        madj = onp.coo_to_csr(feed[ph['madj'][l]])
                        new_hist[l] = act                          # self.new_history = (mu, var), :341
                        act, ctx = det_np.vr_agg_fwd(adj, fadj, madj, act[0], act[1], self.history[l],
                                                     self.history_var[l], feed[ph['fields'][l]],
                                                     feed[ph['ffields'][l]], concat)
                    else:
                        act, ctx = det_np.plain_agg_fwd(adj, act[0], act[1], concat)
                    tape.append(('detagg', ctx, self.cv))
                elif self.cv:
                    fadj = onp.coo_to_csr(feed[ph['fadj'][l]])
                    ifield, ffield = feed[ph['fields'][l]], feed[ph['ffields'][l]]
                    scale = feed[ph['scales'][l]]
                    if self.cvd:
                        h, mu = act
                        oh, om, nh = onp.vr_aggregate(adj, fadj, h, mu, self.history[l], ifield,
                                                      ffield, scale, True, concat)
                        act = (oh, om)
                    else:
                        oh, _, nh = onp.vr_aggregate(adj, fadj, act, None, self.history[l], ifield,
                                                     ffield, scale, False, concat)
                        act = oh
                    new_hist[l] = nh[0]
                    tape.append(('agg', adj, scale if self.cvd else None, concat))
                else:
                    act = onp.plain_aggregate(adj, act, concat)
                    tape.append(('agg', adj, None, concat))
            acts.append(act)
        self._tape, self._new_hist = tape, new_hist
        return act, acts

    # ---- loss -----------------------------------------------------------------------------
    def loss_and_grad(self, logits, labels):
        """gcn/models.py:68-94: weight decay on the first parametrised layer's vars + mean CE."""
        fl = self.flags
        n = logits.shape[0]
        wd_names = self._wd_names()
        loss = f32(0)
        for k in wd_names:
            loss += f32(fl['weight_decay']) * f32(0.5) * f32((self.params[k].astype(np.float64) ** 2).sum())
        if self.multitask:
            z = logits.astype(np.float64)
            ce = np.maximum(z, 0) - z * labels + np.log1p(np.exp(-np.abs(z)))
            loss += f32(ce.mean())
            dlogits = ((1.0 / (1.0 + np.exp(-z)) - labels) / ce.size).astype(f32)
            pred = (1.0 / (1.0 + np.exp(-z))).astype(f32)
            acc = f32(((logits > 0) == (labels > 0.5)).mean())
        else:
            z = logits.astype(np.float64)
            z = z - z.max(axis=1, keepdims=True)
            p = np.exp(z)
            p /= p.sum(axis=1, keepdims=True)
            ce = -(labels * np.log(p + 1e-300)).sum(axis=1)
            loss += f32(ce.mean())
            dlogits = ((p * labels.sum(axis=1, keepdims=True) - labels) / n).astype(f32)
            pred = p.astype(f32)
            acc = f32((logits.argmax(1) == labels.argmax(1)).mean())
        return loss, acc, pred, dlogits

    def _wd_names(self):
        for s in self.specs:
            if s[0] in ('add', 'det'):
                return [k for k in (s[1] + '/weights', s[1] + '/offset', s[1] + '/scale') if k in self.params]
            if s[0] == 'dense':
                return [s[1] + '/weights']       # MyLayerNorm vars are not in Dense.vars
        return []

    # ---- backward -------------------------------------------------------------------------
    def _gate(self, name, pre):
        """ReLU gate.  `relu_gate_hook(name, pre) -> bool array` lets a test enumerate the assignments of
        the gates whose input sits within fp32 rounding of the kink (there the sign, and with it a whole
        gradient contribution, is decided by summation order): tests/test_model_gpu.py _gate_interval."""
        hook = getattr(self, 'relu_gate_hook', None)
        return (pre > 0) if hook is None else hook(name, pre)

    def backward(self, dout):
        grads = {k: np.zeros_like(v) for k, v in self.params.items()}
        g = dout
        for rec in reversed(self._tape):
            kind = rec[0]
            if kind == 'dense':
                _, s, xin, ctx, pre = rec
                _, name, fin, fout, sparse_in, relu, norm = s
                if relu:
                    g = (g * self._gate(name, pre)).astype(f32)
                if norm:
                    g, doff, dsc = layer_norm_bwd(g, ctx, self.params[name + '/scale'])
                    grads[name + '/offset'] += doff
                    grads[name + '/scale'] += dsc
                W = self.params[name + '/weights']
                if sparse_in:
                    grads[name + '/weights'] += np.asarray(xin.T.dot(g), dtype=f32)
                    g = None
                else:
                    grads[name + '/weights'] += (xin.T @ g).astype(f32)
                    g = (g @ W.T).astype(f32)
            elif kind == 'det':
                _, s, ctx = rec
                name, norm = s[1], s[5]
                g, dW, doff, dsc = det_np.fc_bwd(g, ctx)
                grads[name + '/weights'] += dW
                if norm:
                    grads[name + '/offset'] += doff
                    grads[name + '/scale'] += dsc
                if g[1] is None:                                   # plain (first-layer) input
                    g = g[0]
            elif kind == 'gauss':
                _, m, keep, var_, eps_ = rec
                if m is not None:
                    g = (g * (m * f32(1.0 / keep))).astype(f32)
                g = det_np.sample_bwd(g, var_, eps_)
            elif kind == 'detagg':
                _, ctx, cv = rec
                g = det_np.vr_agg_bwd(g, ctx) if cv else det_np.plain_agg_bwd(g, ctx)
            elif kind == 'dropout':
                _, m, keep, _ = rec
                if g is not None and m is not None:
                    g = (g * (m * f32(1.0 / keep))).astype(f32)
            elif kind == 'add':
                _, s, xd, m, keep, ctx, xs_n = rec
                _, name, fin, fout, sparse_in, norm = s
                g = (g * self._gate(name, xs_n)).astype(f32)       # only the x stream carries grad
                if norm:
                    g, doff, dsc = layer_norm_bwd(g, ctx, self.params[name + '/scale'])
                    grads[name + '/offset'] += doff
                    grads[name + '/scale'] += dsc
                W = self.params[name + '/weights']
                if sparse_in:
                    grads[name + '/weights'] += np.asarray(xd.T.dot(g), dtype=f32)
                    g = None
                else:
                    grads[name + '/weights'] += (xd.T @ g).astype(f32)
                    g = (g @ W.T).astype(f32)
                    if m is not None:
                        g = (g * (m * f32(1.0 / keep))).astype(f32)
            elif kind == 'agg':
                _, adj, scale, concat = rec
                d = g.shape[1] // 2 if concat else g.shape[1]
                g_nbr = g[:, d:] if concat else g
                at = adj.T.tocsr()
                # stable transpose keeps ascending output-row order inside each column
                dx = onp.spmm(at.indptr, at.indices, at.data, np.ascontiguousarray(g_nbr),
                              cscale=scale)
                if concat:
                    dx[:adj.shape[0]] += g[:, :d]
                g = dx.astype(f32)
            if getattr(self, 'gtrace', None) is not None:          # test hook: dL/d(input) per record
                self.gtrace.append((kind, g))
        wd = f32(self.flags['weight_decay'])
        for k in self._wd_names():
            grads[k] += wd * self.params[k]
        return grads

    # ---- optimiser + history --------------------------------------------------------------
    def adam_step(self, grads):
        """tf.train.AdamOptimizer(lr, beta1, beta2, epsilon=1e-8) (gcn/models.py:50-51)."""
        fl = self.flags
        self.adam_t += 1
        b1, b2 = fl['beta1'], fl['beta2']
        lr_t = fl['learning_rate'] * np.sqrt(1 - b2 ** self.adam_t) / (1 - b1 ** self.adam_t)
        for k, g in grads.items():
            self.adam_m[k] = (b1 * self.adam_m[k] + (1 - b1) * g).astype(f32)
            self.adam_v[k] = (b2 * self.adam_v[k] + (1 - b2) * g * g).astype(f32)
            self.params[k] = (self.params[k] - f32(lr_t) * self.adam_m[k]
                              / (np.sqrt(self.adam_v[k]) + f32(1e-8))).astype(f32)

    def update_history(self, feed, ph):
        """tf.scatter_update(history, fields[l], new_history) after the optimizer step
        (gcn/models.py:160-166,186-194)."""
        for l, nh in self._new_hist.items():
            if isinstance(nh, tuple):
                onp.scatter_rows(self.history[l], feed[ph['fields'][l]], nh[0])
                onp.scatter_rows(self.history_var[l], feed[ph['fields'][l]], nh[1])
            else:
                onp.scatter_rows(self.history[l], feed[ph['fields'][l]], nh)

    def run_one_step(self, feed, ph, dropout, masks):
        logits, acts = self.forward(feed, ph, dropout if self.is_training else 0.0, masks)
        loss, acc, pred, dlogits = self.loss_and_grad(logits, feed[ph['labels']])
        grads = None
        if self.is_training:
            grads = self.backward(dlogits)
            self.adam_step(grads)
        self.update_history(feed, ph)
        return loss, acc, pred, acts, grads
