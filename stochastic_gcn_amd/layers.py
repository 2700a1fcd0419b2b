"""Layers of the training hot path, MI355X-native (mirror of gcn/layers.py).

Same class names and constructor meaning as the reference (``dot``, ``Dense``,
``AugmentedDropoutDense``, ``Dropout``, ``PlainAggregator``, ``VRAggregator``), but instead
of building a TensorFlow graph each layer runs eagerly on HBM-resident tensors and carries an
explicit ``backward``: the graph is tiny and static, so hand-written gradients (the same ones
TF autodiff derives) keep every sparse product on our HIP kernels in both directions:

  dot(x, y, sparse=True)   -> ops.spmm               (gcn/layers.py:31-37)
  VRAggregator forward     -> ops.vr_aggregate       (gcn/layers.py:298-319,350-362), fused
  aggregator backward      -> ops.spmm on the transposed CSR the sampler emitted (K6)
  history rows             -> read in place by the fused kernel / ops.scatter_rows (models.py)

Dense GEMMs / LayerNorm / ReLU / dropout are the "downstream dense" part:
one fused fp32 MFMA launch per dense layer forward (dropout on the operand load + GEMM + LayerNorm
+ ReLU, ops.dense_fwd), two GEMM launches backward (ops.gemm with the dropout mask recomputed /
applied in the epilogue), at every size: there is no library GEMM on the product path.
"""
import torch

from . import ops
from .flags import FLAGS

class Dropped(object):
    """A dense activation with a PENDING dropout (tf.nn.dropout, gcn/layers.py:396,425-433).

    Dropout masks are a counter-based hash of (seed, layer index, step, element index)
    (include/sgcn.h sgcn_dropout_t), so a mask never has to be stored or even produced: a Dense
    layer consumes a Dropped as is -- its forward GEMM masks the operand while loading it, its
    weight-gradient GEMM recomputes the mask, its input-gradient GEMM applies it in the epilogue.
    Any other consumer calls ``materialize()``."""

    def __init__(self, x, drop):
        self.x, self.drop, self._m = x, drop, None

    @property
    def shape(self):
        return self.x.shape

    def materialize(self):
        if self._m is None:
            x = self.x.materialize() if isinstance(self.x, ops.GatheredRows) else self.x
            self._m = ops.dropout(x, self.drop)
        return self._m


def dense_of(x):
    """A plain tensor from a pending dropout (Dropped) or a pending row gather (ops.GatheredRows)."""
    return x.materialize() if isinstance(x, (Dropped, ops.GatheredRows)) else x


def dot(x, y, sparse=False):
    """Wrapper for matmul (sparse vs dense), as gcn/layers.py:31-37."""
    if sparse:
        return ops.spmm(x, y)
    return ops.gemm(x, y)


class SparseInput(object):
    """A row-sliced sparse feature block on device: CSR + COO row ids (for its transpose)."""

    def __init__(self, csr, t=None):
        self.csr = csr
        self._t = t         # transpose index, shared by the views of one minibatch slice

    @property
    def shape(self):
        return self.csr.shape

    def with_values(self, val):
        return ops.DeviceCSR(self.csr.shape, self.csr.rowptr, self.csr.col, val)

    def transpose_of(self, val):
        """CSR of X^T for dW = X^T g.  The index is a device counting sort (ops.csr_transpose_index, rows
        ascending inside a column -> ordered float sums), built once per minibatch; the values -- they
        change with every dropout mask -- follow through one gather."""
        c = self.csr
        if self._t is None:
            self._t = ops.csr_transpose_index(c)
        t_rowptr, t_row, t_src = self._t
        return ops.DeviceCSR((c.shape[1], c.shape[0]), t_rowptr, t_row, ops.gather_f32(val, t_src))


class Layer(object):
    def __init__(self, name=None):
        self.name = name or self.__class__.__name__.lower()
        self.vars = {}      # name -> tensor view into the model's flat parameter buffer
        self.grads = {}     # name -> tensor view into the model's flat gradient buffer
        self.sparse_inputs = False
        self.need_dx = True  # models.py clears it on the first parametrised layer (no consumer)
        self.index = 0       # position in the model's layer list (part of the dropout key)
        self.key_fn = None   # () -> 32-bit dropout key of this layer at the current step

    def drop_site(self, keep):
        """The dropout site of this layer at the current step, or None when nothing is dropped."""
        if keep >= 1.0:
            return None
        return ops.Drop(keep, self.key_fn() if self.key_fn else ops.dropout_key(FLAGS.seed, self.index, 0))

    def param_shapes(self):
        return []

    def __call__(self, inputs):
        return self.forward(inputs)


class Dropout(Layer):
    """gcn/layers.py:415-433."""

    def __init__(self, keep_prob_fn, cvd, **kw):
        super(Dropout, self).__init__(**kw)
        self.keep_prob_fn, self.cvd = keep_prob_fn, cvd
        self.fuse_next = False      # models.py: the consumer is a Dense layer that applies the mask itself
        self.noise_key_fn = None    # models.py: () -> key of this layer's normal deviates at the current step

    def forward(self, inputs):
        keep = self.keep_prob_fn()
        self._drop, self._sparse, self._fused, self._gauss = self.drop_site(keep), False, False, None
        if self.cvd and isinstance(inputs, tuple):
            inputs = inputs[0]                          # keeps only dropout(h), :423-425
        elif isinstance(inputs, tuple):
            # det-dropout (mu, var): draw x ~ N(mu, var + 1e-10), then ordinary dropout (:425-428).  The normal
            # deviates are a counter hash like the masks (layer index + 4096 keeps the two streams apart)
            mu, var = inputs
            key = self.noise_key_fn() if self.noise_key_fn else ops.dropout_key(FLAGS.seed, self.index + 4096, 0)
            self._gauss = (var, key)
            inputs = ops.gauss_sample(mu, var, key)
        if isinstance(inputs, SparseInput):
            self._sparse = True
            if self._drop is None:
                return inputs
            if inputs._t is None:
                inputs._t = ops.csr_transpose_index(inputs.csr)
            out = SparseInput(inputs.csr, inputs._t)            # same structure, dropped values
            out.csr = inputs.with_values(ops.dropout(inputs.csr.val, self._drop))
            out.csr.coo_rows = inputs.csr.coo_rows
            return out
        if not (self.fuse_next and isinstance(inputs, ops.GatheredRows)):
            inputs = dense_of(inputs)      # (a pending row gather rides on into a fusing Dense layer)
        if self._drop is None:
            return inputs
        pending = Dropped(inputs, self._drop)
        if self.fuse_next:
            self._fused = True
            return pending
        return pending.materialize()

    def backward(self, g):
        if not (g is None or self._sparse or self._drop is None or self._fused):
            g = ops.dropout(g, self._drop)        # (fused: the consumer's dx GEMM already applied the mask)
        if self._gauss is not None and g is not None:
            var, key = self._gauss
            return g, ops.gauss_sample_bwd(var, g, key)
        return g


class Dense(Layer):
    """gcn/layers.py:100-138: x.W (dense or sparse x) -> MyLayerNorm -> act."""

    def __init__(self, input_dim, output_dim, placeholders=None, sparse_inputs=False, act=True,
                 norm=True, **kw):
        super(Dense, self).__init__(**kw)
        self.input_dim, self.output_dim = input_dim, output_dim
        self.sparse_inputs, self.act, self.norm = sparse_inputs, act, norm

    def param_shapes(self):
        s = [('weights', (self.input_dim, self.output_dim), 'glorot')]
        if self.norm:   # MyLayerNorm creates trainable offset/scale (gcn/layers.py:90-92)
            s += [('offset', (1, self.output_dim), 'zeros'), ('scale', (1, self.output_dim), 'ones')]
        return s

    def wd_vars(self):
        return ['weights']          # the LN variables are not in Dense.vars (gcn/models.py:73-75)

    def forward(self, x):
        W = self.vars['weights']
        self._drop = None
        if isinstance(x, Dropped):         # pending dropout: applied by our own GEMMs
            x, self._drop = x.x, x.drop
        if isinstance(x, ops.GatheredRows) and not (self.output_dim <= 128 or not (self.norm or self.act)):
            x = x.materialize()            # only the fused launch reads rows through an index
        self._x = x
        off = self.vars.get('offset') if self.norm else None
        sc = self.vars.get('scale') if self.norm else None
        self._ctx = None
        if self.sparse_inputs:
            y = ops.spmm(x.csr, W)
            if self.norm or self.act:      # fused LayerNorm + ReLU (one kernel)
                y, self._ctx = ops.ln_act_fwd(y, off, sc, self.act)
        elif self.output_dim <= 128 or not (self.norm or self.act):
            # dropout + GEMM + LN + ReLU, one launch
            y, self._ctx = ops.dense_fwd(x, W, off, sc, self.act, drop=self._drop)
        else:
            y, self._ctx = ops.ln_act_fwd(ops.gemm(x, W, drop_a=self._drop), off, sc, self.act)
        self._out = y
        return y

    def backward(self, g):
        if self.sparse_inputs:
            if self.norm or self.act:
                g = ops.ln_act_bwd(g, self._out, self._ctx, self.vars.get('scale') if self.norm else None,
                                   self.act, self.grads.get('offset'), self.grads.get('scale'))
            xt = self._x.transpose_of(self._x.csr.val)
            ops.spmm(xt, g, out=self.grads['weights'], beta=1.0)
            return None
        # LN/ReLU backward -> dW += dropout(x)^T g -> dx = (g W^T) * mask, one call
        return ops.dense_bwd(g, self._out, self._ctx, self.vars.get('scale') if self.norm else None, self.act,
                             self._x, self.vars['weights'], self.grads['weights'], self.grads.get('offset'),
                             self.grads.get('scale'), need_dx=self.need_dx, drop=self._drop)


class AugmentedDropoutDense(Layer):
    """gcn/layers.py:365-412: the CVD dense layer with a dropout stream x and a clean,
    stop-gradient stream mu sharing weights and LayerNorm parameters."""

    def __init__(self, keep_prob_fn, input_dim, output_dim, sparse_inputs=False, norm=True, **kw):
        super(AugmentedDropoutDense, self).__init__(**kw)
        self.keep_prob_fn = keep_prob_fn
        self.input_dim, self.output_dim = input_dim, output_dim
        self.sparse_inputs, self.norm = sparse_inputs, norm

    def param_shapes(self):
        s = [('weights', (self.input_dim, self.output_dim), 'glorot')]
        if self.norm:
            s += [('offset', (1, self.output_dim), 'zeros'), ('scale', (1, self.output_dim), 'ones')]
        return s

    def wd_vars(self):
        return [n for n, _, _ in self.param_shapes()]

    def forward(self, inputs):
        x, mu = inputs if isinstance(inputs, tuple) else (inputs, inputs)
        keep = self.keep_prob_fn()
        W = self.vars['weights']
        self._drop = drop = self.drop_site(keep)
        off = self.vars.get('offset') if self.norm else None
        sc = self.vars.get('scale') if self.norm else None
        if self.sparse_inputs:
            val = x.csr.val if drop is None else ops.dropout(x.csr.val, drop)
            self._xd = (x, val)
            xs = ops.spmm(x.with_values(val), W)
            mus = xs if (mu is x and drop is None) else ops.spmm(mu.csr, W)
            hx, self._ctx = ops.ln_act_fwd(xs, off, sc, True)       # fused LN + ReLU
            hmu = hx if mus is xs else ops.ln_act_fwd(mus, off, sc, True)[0]
            self._out = hx
            return hx, hmu
        fused = self.output_dim <= 128
        if not fused:
            x, mu = dense_of(x), dense_of(mu)
        else:       # a pending dropout is materialised; a pending row gather rides into the GEMMs
            same = mu is x
            x = x.materialize() if isinstance(x, Dropped) else x
            mu = x if same else (mu.materialize() if isinstance(mu, Dropped) else mu)
        self._x = x
        if mu is x and drop is None:
            # test models run with dropout 0 on a single stream: both streams coincide
            hx, self._ctx = ops.dense_fwd(x, W, off, sc, True) if fused else \
                ops.ln_act_fwd(ops.gemm(x, W), off, sc, True)
            self._out = hx
            return hx, hx
        # the dropout stream and the clean stream share W and the LayerNorm parameters: ONE stacked
        # GEMM + LN + ReLU launch over [dropout(x) ; mu] (2n x d) -- the operand is never
        # concatenated and the dropout mask never materialised (ops.dense_fwd x2 / drop)
        n = x.shape[0]
        if fused:
            h2, ctx2 = ops.dense_fwd(x, W, off, sc, True, x2=mu, drop=drop)
        else:
            # wide layers (> 128 columns): the two streams' products land in the two halves of ONE pre-activation
            # buffer (the dropout applied while A is loaded, ops.gemm drop_a), then one LN + ReLU pass over both
            pre = torch.empty((n + int(mu.shape[0]), self.output_dim), dtype=torch.float32, device=x.device)
            ops.gemm(x, W, out=pre[:n], drop_a=drop)
            ops.gemm(mu, W, out=pre[n:])
            h2, ctx2 = ops.ln_act_fwd(pre, off, sc, True)
        self._ctx = (ctx2[0][:n], ctx2[1][:n]) if ctx2 is not None else None
        self._out = h2[:n]
        return h2[:n], h2[n:]

    def backward(self, g):
        # only the x stream carries gradient: mu is stop_gradient (gcn/layers.py:412)
        if self.sparse_inputs:
            g = ops.ln_act_bwd(g, self._out, self._ctx, self.vars.get('scale') if self.norm else None, True,
                               self.grads.get('offset'), self.grads.get('scale'))
            x, val = self._xd
            ops.spmm(x.transpose_of(val), g, out=self.grads['weights'], beta=1.0)
            return None
        return ops.dense_bwd(g, self._out, self._ctx, self.vars.get('scale') if self.norm else None, True,
                             self._x, self.vars['weights'], self.grads['weights'], self.grads.get('offset'),
                             self.grads.get('scale'), need_dx=self.need_dx, drop=self._drop)


class DetDropoutFC(Layer):
    """gcn/layers.py:141-202: X -> Dropout -> Linear -> LayerNorm -> ReLU with the dropout noise integrated out: the
    layer maps (mean, variance) to (mean, variance).  The two linear maps are ops.gemm (mu W and var_in (1.2 W^2), the
    factor being the reference's own `* 1.2  # TODO hack`), the mean stream's LayerNorm is the ordinary fused kernel,
    everything else is sgcn_det.hip; the backward is the hand-derived one the oracle pins against autograd."""
    LN_EPS = 1e-10

    def __init__(self, keep_prob_fn, input_dim, output_dim, sparse_inputs=False, norm=True, **kw):
        super(DetDropoutFC, self).__init__(**kw)
        if sparse_inputs:
            raise NotImplementedError("det_dropout on sparse input features (the reference marks it TODO, "
                                      "gcn/layers.py:145)")
        self.keep_prob_fn, self.input_dim, self.output_dim, self.norm = keep_prob_fn, input_dim, output_dim, norm

    def param_shapes(self):
        s = [('weights', (self.input_dim, self.output_dim), 'glorot')]
        if self.norm:
            s += [('offset', (1, self.output_dim), 'zeros'), ('scale', (1, self.output_dim), 'ones')]
        return s

    def wd_vars(self):
        return [n for n, _, _ in self.param_shapes()]

    def forward(self, inputs):
        mu, var = inputs if isinstance(inputs, tuple) else (dense_of(inputs), None)
        keep = self.keep_prob_fn()
        W = self.vars['weights']
        var_in = ops.det_pre(mu, var, keep)
        W2 = ops.square(W, 1.2)
        mu1, var1 = ops.gemm(mu, W), ops.gemm(var_in, W2)
        ctx = None
        if self.norm:
            mu2, ctx = ops.ln_act_fwd(mu1, self.vars['offset'], self.vars['scale'], False, eps=self.LN_EPS)
            var2 = ops.det_lnvar_fwd(var1, ctx[1], self.vars['scale'], self.LN_EPS)
        else:
            mu2, var2 = mu1, var1
        self._saved = (mu, var is not None, var_in, W2, var1, mu2, var2, ctx, keep)
        return ops.det_relu_fwd(mu2, var2)

    def backward(self, g):
        mu, had_var, var_in, W2, var1, mu2, var2, ctx, keep = self._saved
        W = self.vars['weights']
        g_mu, g_var = ops.det_relu_bwd(mu2, var2, g[0].contiguous(), g[1].contiguous())
        if self.norm:
            sc = self.vars['scale']
            g_mu1 = ops.ln_act_bwd(g_mu, mu2, ctx, sc, False, self.grads['offset'], self.grads['scale'])
            g_var = ops.det_lnvar_bwd(g_var, var1, ctx[0], ctx[1], sc, self.LN_EPS, g_mu1, self.grads['scale'])
            g_mu = g_mu1
        # dW = mu^T d_mu1 + 2.4 W (.) (var_in^T d_var1)
        ops.gemm(mu, g_mu, out=self.grads['weights'], trans_a=True, accumulate=True)
        ops.addmul(self.grads['weights'], W, ops.gemm(var_in, g_var, trans_a=True), 2.4)
        if not self.need_dx:
            return None
        d_mu = ops.gemm(g_mu, W, trans_b=True)
        d_var = ops.det_pre_bwd(mu, ops.gemm(g_var, W2, trans_b=True), keep, d_mu, had_var)
        return (d_mu, d_var) if had_var else d_mu


def _squared(A):
    """tf.square(adj) of a DeviceCSR (and of its transpose): same structure and launch plan, squared values."""
    A2 = ops.DeviceCSR(A.shape, A.rowptr, A.col, ops.square(A.val), A.plan)
    if A.transpose is not None:
        T = A.transpose
        A2.transpose = ops.DeviceCSR(T.shape, T.rowptr, T.col, ops.square(T.val), T.plan)
    return A2


class PlainAggregator(Layer):
    """gcn/layers.py:214-257: Z = A.H, or concat(H[:n1], A.H); on (mu, var): A.mu and A^2.var."""

    def __init__(self, model, l, **kw):
        super(PlainAggregator, self).__init__(**kw)
        self.model, self.l = model, l

    def forward(self, x):
        A = self.model.cur.adj[self.l]
        concat = FLAGS.normalization != 'gcn'
        self._det = isinstance(x, tuple)
        if self._det:                                   # (mu, var): A mu and A^2 var, gcn/layers.py:236-248
            mu, var = x
            n1, d = A.shape[0], mu.shape[1]
            self._A, self._A2, self._concat, self._d = A, _squared(A), concat, d
            if not concat:
                return ops.spmm(A, mu), ops.spmm(self._A2, var)
            om = torch.empty((n1, 2 * d), dtype=torch.float32, device=mu.device)
            ov = torch.empty((n1, 2 * d), dtype=torch.float32, device=mu.device)
            om[:, :d], ov[:, :d] = mu[:n1], var[:n1]
            ops.spmm(A, mu, out=om[:, d:])
            ops.spmm(self._A2, var, out=ov[:, d:])
            return om, ov
        x = dense_of(x)
        n1, d = A.shape[0], x.shape[1]
        self._A, self._concat, self._d = A, concat, d
        if not concat:
            return ops.spmm(A, x)
        out = torch.empty((n1, 2 * d), dtype=torch.float32, device=x.device)
        out[:, :d] = x[:n1]
        ops.spmm(A, x, out=out[:, d:])
        return out

    def backward(self, g):
        A, d = self._A, self._d
        if self._det:
            A2, n1 = self._A2, A.shape[0]
            if not self._concat:
                return ops.spmm(A.transpose, g[0]), ops.spmm(A2.transpose, g[1])
            return (ops.spmm(A.transpose, g[0][:, d:], add=g[0][:, :d], add_rows=n1),
                    ops.spmm(A2.transpose, g[1][:, d:], add=g[1][:, :d], add_rows=n1))
        if not self._concat:
            return ops.spmm(A.transpose, g)
        # dX = A^T g_nbr + [g_self ; 0]: the self term rides in the SpMM epilogue (one launch)
        return ops.spmm(A.transpose, g[:, d:], add=g[:, :d], add_rows=A.shape[0])


class VRAggregator(Layer):
    """gcn/layers.py:282-362 (cvd and plain-CV branches) on the fused HIP kernel."""

    def __init__(self, model, l, cvd, **kw):
        super(VRAggregator, self).__init__(**kw)
        self.model, self.l, self.cvd = model, l, cvd
        self.new_history = None

    def forward(self, inputs):
        cur, l = self.model.cur, self.l
        A, P = cur.adj[l], cur.fadj[l]
        concat = FLAGS.normalization != 'gcn'
        hist = self.model.history[l][0]
        self._det = isinstance(inputs, tuple) and not self.cvd
        if self._det:
            return self._forward_det(inputs, cur, A, P, concat)
        if self.cvd:
            h, mu = (dense_of(t) for t in inputs)
            out_h, out_mu = ops.vr_aggregate(A, P, h, mu, hist, cur.fields[l], cur.ffields[l],
                                             cur.scales[l], True, concat)
            self.new_history = [mu]
            out = (out_h, out_mu)
        else:
            x = inputs = dense_of(inputs)
            out_h, _ = ops.vr_aggregate(A, P, x, None, hist, cur.fields[l], cur.ffields[l],
                                        None, False, concat)
            self.new_history = [x]
            out = out_h
        self._A, self._concat, self._s = A, concat, (cur.scales[l] if self.cvd else None)
        self._d = (inputs[0] if self.cvd else inputs).shape[1]
        return out

    def _forward_det(self, inputs, cur, A, P, concat):
        """gcn/layers.py:320-349: the control-variate estimator on (mu, var) with a mean and a variance history:
            mu_nbr  = A (mu - Hm[if]) + P Hm[ff]
            var_nbr = relu(A^2 ds^2 + P^2 Hv[ff] + 2 M (ds . sbar)) + 1e-10,  ds = sqrt(var) - sbar, sbar = sqrt(Hv[if])
        M = the adjacency pattern with the sampler's medg weights.  Seven SpMMs on the general kernel (the history rows are
        read through the index, never gathered) around two element-wise launches."""
        l = self.l
        mu, var = inputs
        Hm, Hv = self.model.history[l]
        ifield, ffield = cur.fields[l], cur.ffields[l]
        n1, d = A.shape[0], mu.shape[1]
        A2, P2, M = _squared(A), _squared(P), cur.madj[l]
        delta_mu, ds2, msig2, ds, sbar = ops.det_agg_prep(mu, var, Hm, Hv, ifield)
        w = 2 * d if concat else d
        om = torch.empty((n1, w), dtype=torch.float32, device=mu.device)
        ov = torch.empty((n1, w), dtype=torch.float32, device=mu.device)
        nb = om[:, d:] if concat else om
        ops.spmm(A, delta_mu, out=nb)
        ops.spmm(P, Hm, gidx=ffield, out=nb, beta=1.0)
        raw = ops.spmm(A2, ds2)
        ops.spmm(P2, Hv, gidx=ffield, out=raw, beta=1.0)
        ops.spmm(M, msig2, out=raw, beta=1.0)
        ops.relu_eps(raw, 1e-10, out=ov[:, d:] if concat else ov)
        if concat:
            om[:, :d], ov[:, :d] = mu[:n1], var[:n1]
        self.new_history = [mu, var]
        self._A, self._concat, self._d = A, concat, d
        self._saved = (A2, M, var, ds, sbar, raw)
        return om, ov

    def _backward_det(self, g):
        A, d, n1 = self._A, self._d, self._A.shape[0]
        A2, M, var, ds, sbar, raw = self._saved
        gm, gv = (g[0][:, d:], g[1][:, d:]) if self._concat else g
        gv = ops.gate(raw, gv)
        if self._concat:
            d_mu = ops.spmm(A.transpose, gm, add=g[0][:, :d], add_rows=n1)
        else:
            d_mu = ops.spmm(A.transpose, gm)
        g_ds2, g_msig2 = ops.spmm(A2.transpose, gv), ops.spmm(M.transpose, gv)
        d_var = ops.det_agg_prep_bwd(var, ds, sbar, g_ds2, g_msig2, add=g[1][:, :d] if self._concat else None,
                                     add_rows=n1 if self._concat else 0)
        return d_mu, d_var

    def backward(self, g):
        """d/dh of h_nbr = (A (h - mu)) * s + mu_nbr  ->  A^T (s (.) g_nbr); mu and the history
        carry no gradient (stop_gradient gcn/layers.py:412, non-trainable gcn/vrgcn.py:31-32)."""
        if self._det:
            return self._backward_det(g)
        A, d = self._A, self._d
        if not self._concat:
            return ops.spmm(A.transpose, g, cscale=self._s)
        return ops.spmm(A.transpose, g[:, d:], cscale=self._s, add=g[:, :d], add_rows=A.shape[0])
