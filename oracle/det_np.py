"""oracle/det_np.py -- TEST INFRASTRUCTURE ONLY (the parity checker; never the product).

NumPy fp32 restatement of the reference's deterministic-dropout ("det-dropout", --det_dropout) arithmetic, forward and
hand-written backward: every activation is a pair (mean, variance) under the dropout noise and is pushed through the
layers analytically (moment propagation) instead of being sampled.

  DetDropoutFC            gcn/layers.py:141-202   (dropout moments :168-176, linear :178-181 incl. the `* 1.2` the
                                                    reference marks "TODO hack", LayerNorm :184-188, ReLU by moment
                                                    matching of a Gaussian :190-202)
  PlainAggregator (tuple) gcn/layers.py:236-248
  VRAggregator (tuple)    gcn/layers.py:320-349   (two histories per layer, gcn/vrgcn.py:28)
  Dropout (tuple)         gcn/layers.py:425-428   (Gaussian re-sampling, then dropout)

PARITY UNPINNED: TensorFlow ops in the reference (tf.contrib Normal, tf.nn.moments ...), no reference test;
tests/test_model_oracle.py checks this file's backward against float64 torch.autograd of the same forward.
"""
import numpy as np
from scipy.special import erfc

from . import oracle_np as onp

f32 = np.float32
_INV_SQRT_2PI = f32(0.3989422804014327)
_INV_SQRT_2 = f32(0.7071067811865476)


def npdf(x):
    return (_INV_SQRT_2PI * np.exp(f32(-0.5) * x * x)).astype(f32)


def ncdf(x):
    return (f32(0.5) * erfc((-x * _INV_SQRT_2).astype(f32))).astype(f32)          # tf Normal.cdf


# ---- dropout moments ---------------------------------------------------------------------------------------------
def pre_fwd(mu, var, keep):
    """(var + mu^2) / p - mu^2 (tuple input) or (1 - p) / p * mu^2 (plain input): var / p + (1 / p - 1) mu^2."""
    ip = f32(1.0 / keep)
    out = (ip - f32(1)) * mu * mu
    if var is not None:
        out = out + var * ip
    return out.astype(f32)


def pre_bwd(mu, g, keep, had_var):
    ip = f32(1.0 / keep)
    return (g * f32(2) * (ip - f32(1)) * mu).astype(f32), ((g * ip).astype(f32) if had_var else None)


# ---- LayerNorm on (mu, var) --------------------------------------------------------------------------------------
def ln_fwd(mu1, var1, offset, scale, eps=1e-10):
    """mean stream: tf.nn.batch_normalization(mu, mean, variance, offset, scale, 1e-10); variance stream:
    var * scale^2 / variance (no epsilon, gcn/layers.py:186-187)."""
    mean = mu1.mean(axis=1, keepdims=True, dtype=f32)
    V = ((mu1 - mean) ** 2).mean(axis=1, keepdims=True, dtype=f32)
    rstd = (1.0 / np.sqrt(V + f32(eps))).astype(f32)
    xhat = ((mu1 - mean) * rstd).astype(f32)
    mu2 = (xhat * scale + offset).astype(f32)
    var2 = (var1 * (scale * scale) / V).astype(f32)
    return mu2, var2, (xhat, rstd, V, mu1 - mean, var1)


def ln_bwd(g_mu2, g_var2, ctx, scale):
    xhat, rstd, V, xc, var1 = ctx
    d = xhat.shape[1]
    dscale = (g_mu2 * xhat).sum(axis=0, keepdims=True, dtype=f32)
    doffset = g_mu2.sum(axis=0, keepdims=True, dtype=f32)
    dxhat = (g_mu2 * scale).astype(f32)
    d_mu1 = rstd * (dxhat - dxhat.mean(axis=1, keepdims=True, dtype=f32)
                    - xhat * (dxhat * xhat).mean(axis=1, keepdims=True, dtype=f32))
    # variance stream
    s2 = scale * scale
    d_var1 = (g_var2 * s2 / V).astype(f32)
    dscale = dscale + (g_var2 * var1 * f32(2) * scale / V).sum(axis=0, keepdims=True, dtype=f32)
    dV = -(g_var2 * var1 * s2).sum(axis=1, keepdims=True, dtype=f32) / (V * V)
    d_mu1 = d_mu1 + dV * f32(2) * xc / f32(d)
    return d_mu1.astype(f32), d_var1, doffset, dscale.astype(f32)


# ---- ReLU by moment matching ---------------------------------------------------------------------------------------
def relu_fwd(mu, v):
    sigma = np.sqrt(v).astype(f32)
    alpha = (-mu / sigma).astype(f32)
    phi = npdf(alpha)
    Phi = ncdf(alpha)
    Z = (ncdf(-alpha) + f32(1e-10)).astype(f32)
    r = (phi / Z).astype(f32)
    m = (mu + sigma * r).astype(f32)
    mo = (Z * m).astype(f32)
    q = (f32(1) + alpha * r - r * r).astype(f32)
    t = (v * q).astype(f32)
    vr = (np.maximum(t, 0) + f32(1e-10)).astype(f32)
    vo = (Z * vr + Z * Phi * mo * mo).astype(f32)
    return mo, vo, (sigma, alpha, phi, Phi, Z, r, m, mo, q, t, vr)


def relu_bwd(g_mo, g_vo, ctx):
    sigma, a, phi, Phi, Z, r, m, mo, q, t, vr = ctx
    s = sigma
    a_mu, a_s = -1.0 / s, -a / s
    dr = -a * r + r * r
    Z_mu, Z_s = phi / s, a * phi / s
    m_mu, m_s = 1.0 - dr, r - a * dr
    mo_mu, mo_s = Z_mu * m + Z * m_mu, Z_s * m + Z * m_s
    dq = r + a * dr - 2.0 * r * dr
    t_mu, t_s = -s * dq, 2.0 * s * q - s * a * dq
    gate = (t > 0).astype(f32)
    P_mu, P_s = phi * a_mu, phi * a_s
    mo2 = mo * mo
    vo_mu = Z_mu * vr + Z * gate * t_mu + (Z_mu * Phi + Z * P_mu) * mo2 + 2.0 * Z * Phi * mo * mo_mu
    vo_s = Z_s * vr + Z * gate * t_s + (Z_s * Phi + Z * P_s) * mo2 + 2.0 * Z * Phi * mo * mo_s
    d_mu = g_mo * mo_mu + g_vo * vo_mu
    d_var = (g_mo * mo_s + g_vo * vo_s) / (2.0 * s)
    return d_mu.astype(f32), d_var.astype(f32)


# ---- DetDropoutFC ----------------------------------------------------------------------------------------------------
def fc_fwd(inp, W, offset, scale, keep):
    """inp: array (first layer) or (mu, var).  Returns (mu_out, var_out), ctx."""
    mu, var = inp if isinstance(inp, tuple) else (inp, None)
    var_in = pre_fwd(mu, var, keep)
    mu1 = (mu @ W).astype(f32)
    W2 = (f32(1.2) * W * W).astype(f32)                                   # `* 1.2  # TODO hack` (gcn/layers.py:180)
    var1 = (var_in @ W2).astype(f32)
    lctx = None
    if offset is not None:
        mu2, var2, lctx = ln_fwd(mu1, var1, offset, scale)
    else:
        mu2, var2 = mu1, var1
    mo, vo, rctx = relu_fwd(mu2, var2)
    return (mo, vo), (mu, var is not None, var_in, W, W2, lctx, rctx, scale, keep)


def fc_bwd(g, ctx):
    """g = (g_mu_out, g_var_out).  Returns (d_mu, d_var or None), dW, doffset, dscale."""
    mu, had_var, var_in, W, W2, lctx, rctx, scale, keep = ctx
    g_mu2, g_var2 = relu_bwd(g[0], g[1], rctx)
    doff = dsc = None
    if lctx is not None:
        g_mu1, g_var1, doff, dsc = ln_bwd(g_mu2, g_var2, lctx, scale)
    else:
        g_mu1, g_var1 = g_mu2, g_var2
    dW = (mu.T @ g_mu1).astype(f32) + (f32(2.4) * W * (var_in.T @ g_var1)).astype(f32)
    d_mu = (g_mu1 @ W.T).astype(f32)
    d_var_in = (g_var1 @ W2.T).astype(f32)
    d_mu_pre, d_var = pre_bwd(mu, d_var_in, keep, had_var)
    return ((d_mu + d_mu_pre).astype(f32), d_var), dW.astype(f32), doff, dsc


# ---- Gaussian re-sampling in front of a Dropout that receives (mu, var) --------------------------------------------
def gauss_noise(key, shape):
    """N(0, 1) per element as the product generates it (csrc/sgcn_det.hip gauss_of): Box-Muller on two counter-based
    hashes of the row-major element index."""
    from .model_np import _fmix32
    n = int(np.prod(shape))
    M = np.uint64(0xFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    h1 = _fmix32((idx * np.uint64(0x9E3779B1) + np.uint64(key)) & M)
    h2 = _fmix32((((idx * np.uint64(0x85EBCA6B)) & M) + np.uint64(0x165667B1) & M) ^ np.uint64(key))
    u1 = ((h1 >> np.uint64(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
    u2 = ((h2 >> np.uint64(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
    z = np.sqrt(f32(-2.0) * np.log(u1)).astype(f32) * np.cos(f32(6.283185307179586) * u2).astype(f32)
    return z.astype(f32).reshape(shape)


def sample_fwd(mu, var, eps):
    return (mu + eps * np.sqrt(var + f32(1e-10))).astype(f32)


def sample_bwd(g, var, eps):
    return g, (g * eps * f32(0.5) / np.sqrt(var + f32(1e-10))).astype(f32)


# ---- aggregators on (mu, var) ---------------------------------------------------------------------------------------------
def _sq(a):
    b = a.copy()
    b.data = (b.data * b.data).astype(f32)
    return b


def _mm(a, x):
    return onp.spmm(a.indptr, a.indices, a.data, np.ascontiguousarray(x, dtype=f32))


def plain_agg_fwd(adj, mu, var, concat):
    """gcn/layers.py:236-248."""
    n1 = adj.shape[0]
    mu_n, var_n = _mm(adj, mu), _mm(_sq(adj), var)
    if not concat:
        return (mu_n, var_n), (adj, concat)
    return (np.concatenate([mu[:n1], mu_n], axis=1), np.concatenate([var[:n1], var_n], axis=1)), (adj, concat)


def plain_agg_bwd(g, ctx):
    adj, concat = ctx
    n1 = adj.shape[0]
    d = g[0].shape[1] // 2 if concat else g[0].shape[1]
    gm, gv = (g[0][:, d:], g[1][:, d:]) if concat else g
    at = adj.T.tocsr()
    d_mu, d_var = _mm(at, gm), _mm(_sq(at), gv)
    if concat:
        d_mu[:n1] += g[0][:, :d]
        d_var[:n1] += g[1][:, :d]
    return d_mu.astype(f32), d_var.astype(f32)


def vr_agg_fwd(adj, fadj, madj, mu, var, Hm, Hv, ifield, ffield, concat):
    """gcn/layers.py:320-349.  madj: the adjacency pattern of adj with the sampler's medg weights."""
    n1 = adj.shape[0]
    delta_mu = (mu - Hm[ifield]).astype(f32)
    sigma = np.sqrt(var).astype(f32)
    sbar = np.sqrt(Hv[ifield]).astype(f32)
    ds = (sigma - sbar).astype(f32)
    msig = (ds * sbar).astype(f32)
    mu_n = (_mm(adj, delta_mu) + _mm(fadj, Hm[ffield])).astype(f32)
    raw = (_mm(_sq(adj), ds * ds) + _mm(_sq(fadj), Hv[ffield]) + f32(2) * _mm(madj, msig)).astype(f32)
    var_n = (np.maximum(raw, 0) + f32(1e-10)).astype(f32)
    ctx = (adj, madj, sigma, sbar, ds, raw, concat)
    if not concat:
        return (mu_n, var_n), ctx
    return (np.concatenate([mu[:n1], mu_n], axis=1), np.concatenate([var[:n1], var_n], axis=1)), ctx


def vr_agg_bwd(g, ctx):
    adj, madj, sigma, sbar, ds, raw, concat = ctx
    n1 = adj.shape[0]
    d = g[0].shape[1] // 2 if concat else g[0].shape[1]
    gm, gv = (g[0][:, d:], g[1][:, d:]) if concat else g
    gv = (gv * (raw > 0)).astype(f32)
    at, mt = adj.T.tocsr(), madj.T.tocsr()
    d_mu = _mm(at, gm)
    g_ds2 = _mm(_sq(at), gv)
    g_msig = f32(2) * _mm(mt, gv)
    d_ds = f32(2) * ds * g_ds2 + sbar * g_msig
    d_var = (d_ds / (f32(2) * sigma)).astype(f32)
    if concat:
        d_mu[:n1] += g[0][:, :d]
        d_var[:n1] += g[1][:, :d]
    return d_mu.astype(f32), d_var.astype(f32)
