"""Pins the hand-written backward of the NumPy model oracle (oracle/model_np.py): the same
forward is rebuilt with float64 torch tensors on CPU and differentiated by torch.autograd;
loss and every parameter gradient must agree.  (The reference relies on TF autodiff and has
no numeric test for this part -- SURVEY.md §4 -- so autograd is the independent check.)"""
import numpy as np
import pytest
import torch

import model_cases as mc
from oracle import model_np as mnp
from oracle import oracle_np as onp


def _t(x):
    return torch.tensor(np.asarray(x, dtype=np.float64))


def _sp(m):
    m = m.tocoo()
    return torch.sparse_coo_tensor(np.vstack([m.row, m.col]), m.data.astype(np.float64), m.shape).coalesce()


def _ln(x, off, sc):
    mean = x.mean(dim=1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=1, keepdim=True)
    return (x - mean) / torch.sqrt(var + 1e-9) * sc + off


def autograd_forward(model, feed, ph, dropout, masks, params_t):
    """The reference forward (gcn/layers.py, gcn/models.py) in torch float64 with autograd."""
    fl = model.flags
    keep = 1.0 - dropout
    concat = fl['normalization'] != 'gcn'
    f0 = feed[ph['fields'][0]]
    import scipy.sparse as sp
    sparse_x = sp.issparse(model.features)
    act = model.features[f0].tocsr() if sparse_x else _t(model.features[f0])
    for s in model.specs:
        kind = s[0]
        if kind == 'add':
            _, name, fin, fout, sparse_in, norm = s
            W = params_t[name + '/weights']
            x, mu = act if isinstance(act, tuple) else (act, act)
            if sparse_in:
                m = masks('x', (x.nnz,)) if dropout > 0 else None
                xd = mnp.sparse_dropout(x, keep, m)
                xs, mus = torch.sparse.mm(_sp(xd), W), torch.sparse.mm(_sp(x), W)
            else:
                m = masks('x', tuple(x.shape)) if dropout > 0 else None
                xd = x * _t(m) / keep if m is not None else x
                xs, mus = xd @ W, mu @ W
            if norm:
                xs = _ln(xs, params_t[name + '/offset'], params_t[name + '/scale'])
                mus = _ln(mus, params_t[name + '/offset'], params_t[name + '/scale'])
            act = (torch.relu(xs), torch.relu(mus).detach())          # tf.stop_gradient(mu)
        elif kind == 'det':                                           # gcn/layers.py:141-202, verbatim formulas
            _, name, fin, fout, sparse_in, norm = s
            W = params_t[name + '/weights']
            if isinstance(act, tuple):
                mu, var = act
                var = (var + mu ** 2) / keep - mu ** 2
            else:
                mu, var = act, (1 - keep) / keep * act ** 2
            mu, var = mu @ W, (var @ W ** 2) * 1.2
            if norm:
                mean = mu.mean(dim=1, keepdim=True)
                variance = ((mu - mean) ** 2).mean(dim=1, keepdim=True)
                sc, off = params_t[name + '/scale'], params_t[name + '/offset']
                mu = (mu - mean) / torch.sqrt(variance + 1e-10) * sc + off
                var = var * (sc ** 2 / variance)
            ncdf = lambda x: 0.5 * torch.erfc(-x / np.sqrt(2.0))           # noqa: E731
            sigma = torch.sqrt(var)
            alpha = -mu / sigma
            phi = torch.exp(-0.5 * alpha ** 2) / np.sqrt(2 * np.pi)
            Phi, Z = ncdf(alpha), ncdf(-alpha) + 1e-10
            m_ = mu + sigma * phi / Z
            mu = Z * m_
            var = torch.relu(var * (1 + alpha * phi / Z - (phi / Z) ** 2)) + 1e-10
            act = (mu, Z * var + Z * Phi * mu ** 2)
        elif kind == 'dropout' and isinstance(act, tuple) and not model.cvd:    # gcn/layers.py:425-428
            mu, var = act
            x = mu + _t(masks.noise('x', tuple(var.shape))) * torch.sqrt(var + 1e-10)
            m = masks('x', tuple(x.shape)) if dropout > 0 else None
            act = x * _t(m) / keep if m is not None else x
        elif kind == 'agg' and isinstance(act, tuple) and not model.cvd:        # gcn/layers.py:236-248, 320-349
            l = s[1]
            adj = onp.coo_to_csr(feed[ph['adj'][l]])
            A, n1 = _sp(adj), adj.shape[0]
            A2 = _sp(adj.multiply(adj).tocsr())
            mu, var = act
            if model.cv:
                fadj, madj = onp.coo_to_csr(feed[ph['fadj'][l]]), onp.coo_to_csr(feed[ph['madj'][l]])
                P, P2, M = _sp(fadj), _sp(fadj.multiply(fadj).tocsr()), _sp(madj)
                Hm, Hv = _t(model.history[l]), _t(model.history_var[l])
                ifield = torch.tensor(feed[ph['fields'][l]].astype(np.int64))
                ffield = torch.tensor(feed[ph['ffields'][l]].astype(np.int64))
                sbar = torch.sqrt(Hv[ifield])
                ds = torch.sqrt(var) - sbar
                mu_n = torch.sparse.mm(A, mu - Hm[ifield]) + torch.sparse.mm(P, Hm[ffield])
                var_n = torch.sparse.mm(A2, ds ** 2) + torch.sparse.mm(P2, Hv[ffield]) + 2 * torch.sparse.mm(M, ds * sbar)
                var_n = torch.relu(var_n) + 1e-10
            else:
                mu_n, var_n = torch.sparse.mm(A, mu), torch.sparse.mm(A2, var)
            act = (torch.cat([mu[:n1], mu_n], 1), torch.cat([var[:n1], var_n], 1)) if concat else (mu_n, var_n)
        elif kind == 'dropout':
            if model.cvd and isinstance(act, tuple):
                act = act[0]
            if hasattr(act, 'tocsr'):
                m = masks('x', (act.nnz,)) if dropout > 0 else None
                act = mnp.sparse_dropout(act, keep, m)
            else:
                m = masks('x', tuple(act.shape)) if dropout > 0 else None
                act = act * _t(m) / keep if m is not None else act
        elif kind == 'dense':
            _, name, fin, fout, sparse_in, relu, norm = s
            W = params_t[name + '/weights']
            y = torch.sparse.mm(_sp(act), W) if sparse_in else act @ W
            if norm:
                y = _ln(y, params_t[name + '/offset'], params_t[name + '/scale'])
            act = torch.relu(y) if relu else y
        elif kind == 'agg':
            l = s[1]
            A = _sp(onp.coo_to_csr(feed[ph['adj'][l]]))
            n1 = A.shape[0]
            if model.cv:
                P = _sp(onp.coo_to_csr(feed[ph['fadj'][l]]))
                H = _t(model.history[l])
                ifield = torch.tensor(feed[ph['fields'][l]].astype(np.int64))
                ffield = torch.tensor(feed[ph['ffields'][l]].astype(np.int64))
                if model.cvd:
                    h, mu = act
                    mu_nbr = torch.sparse.mm(A, mu - H[ifield]) + torch.sparse.mm(P, H[ffield])
                    h_nbr = torch.sparse.mm(A, h - mu) * _t(feed[ph['scales'][l]])[:, None] + mu_nbr
                    act = (torch.cat([h[:n1], h_nbr], 1), torch.cat([mu[:n1], mu_nbr], 1)) if concat \
                        else (h_nbr, mu_nbr)
                else:
                    a_nbr = torch.sparse.mm(A, act) - torch.sparse.mm(A, H[ifield]) + torch.sparse.mm(P, H[ffield])
                    act = torch.cat([act[:n1], a_nbr], 1) if concat else a_nbr
            else:
                a_nbr = torch.sparse.mm(A, act)
                act = torch.cat([act[:n1], a_nbr], 1) if concat else a_nbr
    return act


@pytest.mark.parametrize("name", sorted(mc.CASES) + sorted(mc.DET_CASES))
def test_oracle_backward_matches_autograd(name):
    from stochastic_gcn_amd.scheduler import PyScheduler
    case = mc.build_case(name)
    fl, c, ph = case['flags'], case['cfg'], case['ph']
    om = mc.make_oracle_model(case, seed=5)
    # non-trivial LN parameters and history so that every term of the backward is exercised
    rng = np.random.RandomState(0)
    for k in om.params:
        if k.endswith('/offset'):
            om.params[k] = rng.standard_normal(om.params[k].shape).astype(np.float32) * 0.1
        if k.endswith('/scale'):
            om.params[k] = (1 + 0.1 * rng.standard_normal(om.params[k].shape)).astype(np.float32)
    for h in om.history:
        h[:] = rng.uniform(-1, 1, h.shape)
    for h in om.history_var:
        h[:] = rng.uniform(0.05, 1, h.shape)
    sch = mc.make_scheduler(case, 1)
    feed = sch.minibatch(c['batch'])
    masks = mc.MaskSource(7, 1.0 - fl['dropout'])
    logits, _ = om.forward(feed, ph, fl['dropout'], masks)
    loss, acc, pred, dlogits = om.loss_and_grad(logits, feed[ph['labels']])
    grads = om.backward(dlogits)

    params_t = {k: _t(v).requires_grad_(True) for k, v in om.params.items()}
    out = autograd_forward(om, feed, ph, fl['dropout'], masks.replay(), params_t)
    labels = _t(feed[ph['labels']])
    wd = sum(0.5 * fl['weight_decay'] * (params_t[k] ** 2).sum() for k in om._wd_names())
    tl = wd + (-(labels * torch.log_softmax(out, dim=1)).sum(dim=1)).mean()
    tl.backward()
    assert onp.rel_err(logits, out.detach().numpy()) < 2e-5
    assert abs(float(loss) - float(tl.detach())) < 2e-5 * max(1.0, abs(float(tl.detach())))
    for k, g in grads.items():
        ref = params_t[k].grad.numpy()
        assert onp.rel_err(g, ref) < 1e-4, (name, k, onp.rel_err(g, ref))


def test_adam_matches_torch_adam_up_to_epsilon_placement():
    """TF's Adam uses lr_t = lr*sqrt(1-b2^t)/(1-b1^t) and eps outside the bias correction; with
    eps -> 0 it coincides with the textbook form."""
    fl = mnp.make_flags(learning_rate=0.01)
    p = {'w/weights': np.ones((3, 2), np.float32)}
    m = mnp.Model.__new__(mnp.Model)
    m.flags, m.params, m.adam_t = fl, p, 0
    m.adam_m = {k: np.zeros_like(v) for k, v in p.items()}
    m.adam_v = {k: np.zeros_like(v) for k, v in p.items()}
    g = {'w/weights': np.full((3, 2), 0.5, np.float32)}
    m.adam_step(g)
    # first step: m_hat = g, v_hat = g^2 -> update = lr * g/|g| = lr
    np.testing.assert_allclose(m.params['w/weights'], 1 - 0.01, rtol=1e-5)
