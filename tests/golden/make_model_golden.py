#!/usr/bin/env python
"""Golden vectors for the model arithmetic ("parity unpinned" at the TensorFlow
boundary) from an INDEPENDENT fp32 implementation: the reference's layer definitions restated with
PyTorch-CPU's own ops -- torch.sparse.mm for tf.sparse_tensor_dense_matmul, index_select for
tf.gather, index_copy_ for tf.scatter_update, F.layer_norm for tf.nn.moments +
tf.nn.batch_normalization, torch.autograd for tf.gradients -- sharing NO arithmetic with
oracle/model_np.py / oracle_c.c (hand-written forward and backward) or with the HIP path.

    python tests/golden/make_model_golden.py            # writes tests/golden/model_steps.npz
    python tests/golden/make_model_golden.py --det      # writes tests/golden/model_steps_det.npz (the --det_dropout cases)

What is shared, and why that is legitimate: the INPUTS -- the synthetic cases (tests/model_cases.py),
the initial weights, the minibatches (the product's sampler, itself bit-exact against the real
reference C++, tests/test_sampler.py) and the dropout masks (a counter hash that is part of the C-ABI
contract, include/sgcn.h sgcn_dropout_t; TensorFlow's own mask stream cannot be reproduced by anything).

Reference definitions followed (file:line under /root/reference):
  layer stack            gcn/models.py:258-337          PP input hstack   gcn/models.py:231-241
  Dense / MyLayerNorm    gcn/layers.py:87-138           ADD layer         gcn/layers.py:365-412
  Dropout, sparse_dropout gcn/layers.py:23-28,415-433   aggregators       gcn/layers.py:249-257,298-319,350-362
  loss / accuracy        gcn/models.py:68-94            Adam              gcn/models.py:50-51 (TF: eps outside the sqrt,
                                                                            lr_t = lr sqrt(1-b2^t)/(1-b1^t))
  history alloc / update gcn/vrgcn.py:23-36, gcn/models.py:160-166,186-194
  --det_dropout          DetDropoutFC gcn/layers.py:141-202 (torch.distributions.Normal for tf Normal), aggregators on
                         (mu, var) :236-248, :320-349, Gaussian re-sampling :425-428, two histories gcn/vrgcn.py:28;
                         shared input besides the masks: the N(0, 1) deviates of the re-sampling (a counter hash like them)
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import model_cases as mc                      # noqa: E402
from oracle import model_np as mnp            # noqa: E402  (init_params + hash_mask only: inputs)
from oracle import det_np                     # noqa: E402  (gauss_noise only: the re-sampling's deviates, an input)

STEPS = 3
SAMPLER_SEED = 1
DROPOUT_SEED = 1


def coo_to_torch(triple):
    idx, w, shape = triple
    idx = np.asarray(idx).reshape(-1, 2)
    return torch.sparse_coo_tensor(torch.from_numpy(idx.T.astype(np.int64)), torch.from_numpy(np.asarray(w, np.float32)),
                                   size=tuple(int(x) for x in shape)).coalesce()


def csr_to_torch(a):
    a = a.tocoo()
    return torch.sparse_coo_tensor(torch.from_numpy(np.vstack([a.row, a.col]).astype(np.int64)),
                                   torch.from_numpy(a.data.astype(np.float32)), size=a.shape).coalesce()


class TorchRef(object):
    def __init__(self, case, params):
        fl, c = case['flags'], case['cfg']
        self.fl, self.case = fl, case
        self.cv, self.cvd = bool(fl['cv']), bool(fl['cvd'])
        self.det = bool(fl.get('det_dropout'))
        feats, nbr = case['feats'], case['nbr']
        self.sparse_input = sp.issparse(feats)
        input_dim = feats.shape[1]
        self_dim = 0 if fl['normalization'] == 'gcn' else input_dim
        pre = fl['preprocess']
        if pre and fl['pp_nbr']:
            self.features = sp.hstack((feats[:, :self_dim], nbr)).tocsr().astype(np.float32) if self.sparse_input \
                else np.hstack((feats[:, :self_dim], nbr)).astype(np.float32)
        else:
            self.features = feats
        self.sparse_mm = self.sparse_input
        if self.sparse_input and not pre:
            self.features = np.asarray(self.features.todense(), np.float32)
            self.sparse_mm = False
        L = fl['num_layers']
        self.L = L - 1 if pre else L
        H = fl['hidden1']
        agg0 = H if pre else input_dim
        self.out_dim = c['classes']
        # gcn/models.py:258-337, restated as a list of (kind, ...) tuples
        dim_s = 1 if fl['normalization'] == 'gcn' else 2
        nfc = fl['num_fc_layers']
        st, cnt = [], 0
        if pre:
            for l in range(nfc):
                ind = input_dim * dim_s if l == 0 else H
                last = self.L == 0 and l + 1 == nfc
                if self.det:                                   # gcn/models.py:275-282
                    st.append(('det', 'dense%d' % cnt, fl['layer_norm']))
                elif self.cvd:
                    st.append(('add', 'dense%d' % cnt, self.sparse_mm and l == 0, fl['layer_norm']))
                else:
                    st.append(('dropout',))
                    st.append(('dense', 'dense%d' % cnt, self.sparse_mm and l == 0, not last,
                               False if last else fl['layer_norm']))
                cnt += 1
        for l in range(self.L):
            st.append(('agg', l))
            for l2 in range(nfc):
                last = l2 + 1 == nfc and l + 1 == self.L
                norm = False if last else fl['layer_norm']
                if self.det and l + 1 != self.L:               # gcn/models.py:312-318
                    st.append(('det', 'dense%d' % cnt, norm))
                elif self.cvd and l + 1 != self.L:
                    st.append(('add', 'dense%d' % cnt, False, norm))
                else:
                    if not fl['reverse']:
                        st.append(('dropout',))
                    st.append(('dense', 'dense%d' % cnt, False, not last, norm))
                    if fl['reverse'] and not last:
                        st.append(('dropout',))
                cnt += 1
        self.stack = st
        self.p = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in params.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.t = 0
        self.history = [torch.zeros((c['n'], agg0 if i == 0 else H)) for i in range(self.L)] if self.cv else []
        self.history_var = [torch.zeros_like(h) for h in self.history] if self.det else []      # gcn/vrgcn.py:28
        self.normal = torch.distributions.Normal(torch.tensor(0.0), torch.tensor(1.0))

    def _mask(self, li, step, shape, keep):
        key = mnp.dropout_key(DROPOUT_SEED, li, step)
        return torch.from_numpy(mnp.hash_mask(key, tuple(shape), keep))

    def _ln(self, x, name):
        d = x.shape[1]
        return F.layer_norm(x, (d,), weight=self.p[name + '/scale'].reshape(-1), bias=self.p[name + '/offset'].reshape(-1),
                            eps=1e-9)

    def _sparse_dropout(self, x_csr, li, step, keep, on):
        if not on:
            return csr_to_torch(x_csr)
        m = self._mask(li, step, (x_csr.nnz,), keep).numpy().astype(bool)       # CSR storage order
        coo = x_csr.tocoo()       # tocoo() of a canonical CSR keeps row-major order
        return torch.sparse_coo_tensor(torch.from_numpy(np.vstack([coo.row[m], coo.col[m]]).astype(np.int64)),
                                       torch.from_numpy((coo.data[m] * np.float32(1.0 / keep)).astype(np.float32)),
                                       size=x_csr.shape).coalesce()

    def step(self, feed, ph, dropout, step):
        fl = self.fl
        keep = 1.0 - dropout
        on = dropout > 0
        concat = fl['normalization'] != 'gcn'
        f0 = torch.from_numpy(np.asarray(feed[ph['fields'][0]]).astype(np.int64))
        if sp.issparse(self.features):
            act = self.features[f0.numpy()].tocsr()
            act.sort_indices()
        else:
            act = torch.from_numpy(self.features).index_select(0, f0)
        new_hist, rec = {}, {}
        for li, s in enumerate(self.stack):
            kind = s[0]
            if kind == 'det':                                      # DetDropoutFC._call, gcn/layers.py:163-202
                _, name, norm = s
                W = self.p[name + '/weights']
                p_ = np.float32(keep)
                if isinstance(act, tuple):
                    mu, var = act
                    mu2 = mu * mu
                    var = (var + mu2) / p_ - mu2
                else:
                    mu = act
                    var = (1 - p_) / p_ * (mu * mu)
                mu = mu @ W
                var = (var @ (W * W)) * np.float32(1.2)
                if norm:
                    mean = mu.mean(1, keepdim=True)
                    variance = ((mu - mean) ** 2).mean(1, keepdim=True)
                    sc, off = self.p[name + '/scale'], self.p[name + '/offset']
                    mu = (mu - mean) * torch.rsqrt(variance + np.float32(1e-10)) * sc + off
                    var = var * (sc * sc / variance)
                sigma = torch.sqrt(var)
                alpha = -mu / sigma
                phi = torch.exp(self.normal.log_prob(alpha))
                Phi = self.normal.cdf(alpha)
                Z = self.normal.cdf(-alpha) + np.float32(1e-10)
                phiZ = phi / Z
                m = mu + sigma * phiZ
                mu = Z * m
                var = torch.relu(var * (1 + alpha * phiZ - phiZ * phiZ)) + np.float32(1e-10)
                var = Z * var + Z * Phi * (mu * mu)
                act = (mu, var)
            elif kind == 'dropout' and self.det and isinstance(act, tuple):          # gcn/layers.py:425-428
                mu, var = act
                eps = torch.from_numpy(det_np.gauss_noise(mnp.dropout_key(DROPOUT_SEED, li + 4096, step), tuple(mu.shape)))
                x = mu + eps * torch.sqrt(var + np.float32(1e-10))
                act = x * (self._mask(li, step, x.shape, keep) * np.float32(1.0 / keep)) if on else x
            elif kind == 'agg' and self.det and isinstance(act, tuple):              # gcn/layers.py:236-248, 320-349
                l = s[1]
                A = coo_to_torch(feed[ph['adj'][l]])
                A2 = torch.sparse_coo_tensor(A.indices(), A.values() ** 2, size=A.shape).coalesce()
                n1 = A.shape[0]
                mu, var = act
                if self.cv:
                    P = coo_to_torch(feed[ph['fadj'][l]])
                    P2 = torch.sparse_coo_tensor(P.indices(), P.values() ** 2, size=P.shape).coalesce()
                    Mj = coo_to_torch(feed[ph['madj'][l]])
                    ifield = torch.from_numpy(np.asarray(feed[ph['fields'][l]]).astype(np.int64))
                    ffield = torch.from_numpy(np.asarray(feed[ph['ffields'][l]]).astype(np.int64))
                    Hm, Hv = self.history[l], self.history_var[l]
                    delta_mu = mu - Hm.index_select(0, ifield)
                    mu_bar = Hm.index_select(0, ffield)
                    sigma = torch.sqrt(var)
                    sigma_bar = torch.sqrt(Hv.index_select(0, ifield))
                    delta_sigma = sigma - sigma_bar
                    var_bar = Hv.index_select(0, ffield)
                    msigma = delta_sigma * sigma_bar
                    mu_nbr = torch.sparse.mm(A, delta_mu) + torch.sparse.mm(P, mu_bar)
                    var_nbr = torch.sparse.mm(A2, delta_sigma * delta_sigma) + torch.sparse.mm(P2, var_bar) \
                        + 2 * torch.sparse.mm(Mj, msigma)
                    var_nbr = torch.relu(var_nbr) + np.float32(1e-10)
                    new_hist[l] = (ifield, mu.detach().clone(), var.detach().clone())
                else:
                    mu_nbr, var_nbr = torch.sparse.mm(A, mu), torch.sparse.mm(A2, var)
                act = (torch.cat((mu[:n1], mu_nbr), 1), torch.cat((var[:n1], var_nbr), 1)) if concat else (mu_nbr, var_nbr)
                rec['agg%d' % l] = act
                rec['aggvar%d' % l] = act[1]
            elif kind == 'add':
                _, name, sparse_in, norm = s
                W = self.p[name + '/weights']
                x, mu = act if isinstance(act, tuple) else (act, act)
                if sparse_in:
                    xs = torch.sparse.mm(self._sparse_dropout(x, li, step, keep, on), W)
                    mus = torch.sparse.mm(csr_to_torch(mu), W)
                else:
                    xd = x * (self._mask(li, step, x.shape, keep) * np.float32(1.0 / keep)) if on else x
                    xs, mus = xd @ W, mu @ W
                if norm:
                    xs, mus = self._ln(xs, name), self._ln(mus, name)
                act = (torch.relu(xs), torch.relu(mus).detach())
            elif kind == 'dropout':
                if self.cvd and isinstance(act, tuple):
                    h = act[0]
                    act = h * (self._mask(li, step, h.shape, keep) * np.float32(1.0 / keep)) if on else h
                elif sp.issparse(act):
                    act = ('sparse', self._sparse_dropout(act, li, step, keep, on))
                else:
                    act = act * (self._mask(li, step, act.shape, keep) * np.float32(1.0 / keep)) if on else act
            elif kind == 'dense':
                _, name, sparse_in, relu, norm = s
                W = self.p[name + '/weights']
                if sparse_in:
                    xin = act[1] if isinstance(act, tuple) and act[0] == 'sparse' else csr_to_torch(act)
                    y = torch.sparse.mm(xin, W)
                else:
                    y = act @ W
                if norm:
                    y = self._ln(y, name)
                act = torch.relu(y) if relu else y
            elif kind == 'agg':
                l = s[1]
                A = coo_to_torch(feed[ph['adj'][l]])
                n1 = A.shape[0]
                if self.cv:
                    P = coo_to_torch(feed[ph['fadj'][l]])
                    ifield = torch.from_numpy(np.asarray(feed[ph['fields'][l]]).astype(np.int64))
                    ffield = torch.from_numpy(np.asarray(feed[ph['ffields'][l]]).astype(np.int64))
                    hist = self.history[l]
                    if self.cvd:
                        h, mu = act
                        scale = torch.from_numpy(np.asarray(feed[ph['scales'][l]], np.float32))
                        mu_small = hist.index_select(0, ifield)
                        mu_large = hist.index_select(0, ffield)
                        mu_nbr = torch.sparse.mm(A, mu - mu_small) + torch.sparse.mm(P, mu_large)
                        h_nbr = torch.sparse.mm(A, h - mu) * scale.unsqueeze(1) + mu_nbr
                        new_hist[l] = (ifield, mu.detach().clone())
                        act = (torch.cat((h[:n1], h_nbr), 1), torch.cat((mu[:n1], mu_nbr), 1)) if concat \
                            else (h_nbr, mu_nbr)
                    else:
                        x = act
                        a_nbr = torch.sparse.mm(A, x) - torch.sparse.mm(A, hist.index_select(0, ifield)) \
                            + torch.sparse.mm(P, hist.index_select(0, ffield))
                        new_hist[l] = (ifield, x.detach().clone())
                        act = torch.cat((x[:n1], a_nbr), 1) if concat else a_nbr
                else:
                    a_nbr = torch.sparse.mm(A, act)
                    act = torch.cat((act[:n1], a_nbr), 1) if concat else a_nbr
                rec['agg%d' % l] = act
        logits = act
        labels = torch.from_numpy(np.asarray(feed[ph['labels']], np.float32))
        # gcn/models.py:68-83: weight decay on the vars of the first layer that has any
        first = next(s for s in self.stack if s[0] in ('add', 'dense', 'det'))
        wd_names = [k for k in (first[1] + '/weights', first[1] + '/offset', first[1] + '/scale') if k in self.p] \
            if first[0] in ('add', 'det') else [first[1] + '/weights']
        loss = sum(fl['weight_decay'] * 0.5 * (self.p[k] ** 2).sum() for k in wd_names) \
            + (-(labels * F.log_softmax(logits, dim=1)).sum(1)).mean()
        acc = (logits.argmax(1) == labels.argmax(1)).float().mean()
        for v in self.p.values():
            v.grad = None
        loss.backward()
        grads = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in self.p.items()}
        # tf.train.AdamOptimizer
        self.t += 1
        b1, b2 = fl['beta1'], fl['beta2']
        lr_t = fl['learning_rate'] * np.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
        with torch.no_grad():
            for k, v in self.p.items():
                g = grads[k]
                self.m[k] = b1 * self.m[k] + (1 - b1) * g
                self.v[k] = b2 * self.v[k] + (1 - b2) * g * g
                v -= np.float32(lr_t) * self.m[k] / (self.v[k].sqrt() + np.float32(1e-8))
            for l, nh in new_hist.items():                      # scatter AFTER the optimizer step
                self.history[l].index_copy_(0, nh[0], nh[1])
                if len(nh) == 3:
                    self.history_var[l].index_copy_(0, nh[0], nh[2])
        out = dict(loss=np.float32(loss.item()), acc=np.float32(acc.item()), logits=logits.detach().numpy())
        for k, a in rec.items():
            a = a[0] if isinstance(a, tuple) else a
            out[k] = a.detach().numpy()
        for k, g in grads.items():
            out['grad/' + k] = g.numpy()
        for k, v in self.p.items():
            out['param/' + k] = v.detach().numpy().copy()
        return out


def generate(det=False):
    torch.manual_seed(0)
    torch.set_num_threads(1)          # fixed reduction order -> reproducible file
    blob = {}
    for name in sorted(mc.DET_CASES if det else mc.CASES):
        case = mc.build_case(name)
        fl, c, ph = case['flags'], case['cfg'], case['ph']
        params = mnp.init_params(mc.make_oracle_model(case, seed=3).specs, 3)
        ref = TorchRef(case, params)
        sch = mc.make_scheduler(case, SAMPLER_SEED)
        for step in range(STEPS):
            feed = sch.minibatch(c['batch'])
            out = ref.step(feed, ph, fl['dropout'], step)
            blob['%s/s%d/field0' % (name, step)] = np.asarray(feed[ph['fields'][0]], np.int32)
            for k, v in out.items():
                blob['%s/s%d/%s' % (name, step, k)] = np.asarray(v, np.float32)
        for l, h in enumerate(ref.history):
            blob['%s/history%d' % (name, l)] = h.numpy()
        for l, h in enumerate(ref.history_var):
            blob['%s/history_var%d' % (name, l)] = h.numpy()
    return blob


if __name__ == "__main__":
    det = "--det" in sys.argv[1:]
    blob = generate(det)
    out = os.path.join(HERE, "model_steps_det.npz" if det else "model_steps.npz")
    np.savez_compressed(out, **blob)
    print("wrote %s: %d arrays, %.2f MB" % (out, len(blob), os.path.getsize(out) / 1e6))
