"""Data loading / metrics helpers the training driver needs (subset of gcn/utils.py).

``load_data(dataset)`` returns the reference's 10-tuple (gcn/utils.py:183,335).  It reads the
reference's ``.npz`` dataset cache -- same file names (``data/<name>.npz`` /
``data/<name>_deg<max_degree>.npz`` for GraphSAGE sets, gcn/utils.py:193-196;
``data/<name>_<normalization>.npz`` for the Planetoid sets, gcn/utils.py:34) and key schema
(gcn/utils.py:172-181, :325-333) -- so a cache produced by the reference drops in.  The
deterministic synthetic stand-ins of synthetic.py (no datasets and no network on the box) are
selected explicitly: ``--dataset s-reddit`` / ``s-cora`` / ``s-pubmed`` or ``--synthetic``.
Raw-format parsers (networkx-1.11 JSON, Planetoid pickles) are out of scope.
"""
import os

import numpy as np
import scipy.sparse as sp

from . import synthetic
from .flags import FLAGS


def _csr(z, prefix):
    shape = tuple(int(x) for x in z[prefix + '_shape'])
    return sp.csr_matrix((z[prefix + '_data'], z[prefix + '_indices'], z[prefix + '_indptr']),
                         shape=shape, dtype=np.float32)


def load_npz_cache(path):
    """The reference's dataset cache schema (gcn/utils.py:172-181, :325-333)."""
    z = np.load(path)
    num_data = int(z['num_data'])
    train_adj, full_adj = _csr(z, 'train_adj'), _csr(z, 'full_adj')
    if 'feats' in z.files:                                  # GraphSAGE sets: dense features
        feats = z['feats'].astype(np.float32)
        train_feats = z['train_feats'].astype(np.float32)
        test_feats = z['test_feats'].astype(np.float32)
    else:                                                   # Planetoid sets: sparse features
        feats, train_feats, test_feats = _csr(z, 'feats'), _csr(z, 'train_feats'), _csr(z, 'test_feats')
    return (num_data, train_adj, full_adj, feats, train_feats, test_feats,
            z['labels'].astype(np.float32), z['train_data'].astype(np.int32),
            z['val_data'].astype(np.int32), z['test_data'].astype(np.int32))


def save_npz_cache(path, tup):
    """Write a 10-tuple in the reference's cache schema (round-trips with load_npz_cache)."""
    num_data, train_adj, full_adj, feats, train_feats, test_feats, labels, tr, va, te = tup
    blob = dict(num_data=num_data, labels=labels, train_data=tr, val_data=va, test_data=te)
    for name, m in (('train_adj', train_adj), ('full_adj', full_adj)):
        blob.update({name + '_data': m.data, name + '_indices': m.indices,
                     name + '_indptr': m.indptr, name + '_shape': m.shape})
    if sp.issparse(feats):
        for name, m in (('feats', feats), ('train_feats', train_feats), ('test_feats', test_feats)):
            m = m.tocsr()
            blob.update({name + '_data': m.data, name + '_indices': m.indices,
                         name + '_indptr': m.indptr, name + '_shape': m.shape})
    else:
        blob.update(feats=feats, train_feats=train_feats, test_feats=test_feats)
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, 'wb') as f:
        np.savez(f, **blob)


GCN_DATASETS = ('cora', 'citeseer', 'pubmed', 'nell')     # gcn/utils.py:467


def cache_path(dataset):
    """The reference's cache file name for ``dataset`` under the current flags:
    ``data/{name}_{normalization}.npz`` for the Planetoid sets (gcn/utils.py:34),
    ``data/{name}.npz`` or ``data/{name}_deg{max_degree}.npz`` for GraphSAGE-format sets
    (gcn/utils.py:193-196)."""
    if dataset in GCN_DATASETS:
        return 'data/{}_{}.npz'.format(dataset, FLAGS.normalization)
    if FLAGS.max_degree == -1:
        return 'data/{}.npz'.format(dataset)
    return 'data/{}_deg{}.npz'.format(dataset, FLAGS.max_degree)


def load_data(dataset):
    """gcn/utils.py:466-473.  ``s-cora`` / ``s-pubmed`` / ``s-reddit`` (or any name with
    ``--synthetic``) select the deterministic synthetic stand-ins of synthetic.py; a real dataset
    name loads the reference's ``.npz`` cache and FAILS when it is absent -- a training run never
    silently reports numbers on stand-in data."""
    if dataset.startswith('s-') or FLAGS.synthetic:
        print('Synthetic stand-in for "{}" (scale={})'.format(dataset, FLAGS.scale))
        return synthetic.load_data(dataset, FLAGS.normalization, FLAGS.scale)
    cand = cache_path(dataset)
    if os.path.exists(cand):
        print('Found preprocessed dataset {}, loading...'.format(cand))
        return load_npz_cache(cand)
    raise FileNotFoundError(
        'no dataset cache {} for "{}" (the reference writes it on its first run; the raw-format '
        'parsers are out of scope here).  Use --dataset s-{} or --synthetic for the synthetic '
        'stand-in.'.format(cand, dataset, dataset))


class Averager(object):
    """gcn/utils.py:500-511."""

    def __init__(self, window_size):
        self.window_size = window_size
        self.window = []

    def add(self, n):
        self.window.append(n)
        if len(self.window) > self.window_size:
            self.window = self.window[1:]

    def mean(self):
        return np.mean(self.window)


def f1_from_classes(true, pred):
    """sklearn.metrics.f1_score(true, pred, average="micro" / "macro") for single-label class indices, from three
    bincounts: per label 2 tp / (true count + predicted count) over the labels that occur in either array, in float64 --
    sklearn's own expression (precision_recall_fscore_support) without its input validation and label encoding, which are
    most of the 5 ms a 24 k-row validation sweep spent on the host (tests/test_utils.py holds the two against each other)."""
    true, pred = np.asarray(true, dtype=np.int64), np.asarray(pred, dtype=np.int64)
    k = int(max(true.max(), pred.max())) + 1 if true.size else 1
    tp = np.bincount(true[true == pred], minlength=k)
    ts, ps = np.bincount(true, minlength=k), np.bincount(pred, minlength=k)
    present = (ts + ps) > 0
    if not present.any():
        return 0.0, 0.0
    micro = 2.0 * float(tp.sum()) / float(ts.sum() + ps.sum())
    f = 2.0 * tp[present].astype(np.float64) / (ts[present] + ps[present]).astype(np.float64)
    return float(micro), float(np.average(f))


def calc_f1(y_pred, y_true, multitask):
    """gcn/utils.py:521-529 (sklearn micro / macro F1)."""
    from sklearn.metrics import f1_score
    if multitask:
        y_pred = (y_pred > 0.5).astype(np.int64)
        y_true = (y_true > 0.5).astype(np.int64)
    else:
        y_true = np.argmax(y_true, axis=1)
        y_pred = np.argmax(y_pred, axis=1)
    return f1_score(y_true, y_pred, average="micro"), f1_score(y_true, y_pred, average="macro")
