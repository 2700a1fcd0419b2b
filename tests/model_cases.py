"""Small synthetic configurations that mirror the BASELINE.json configs, shared by the CPU
oracle tests and the GPU parity tests."""
import numpy as np
import scipy.sparse as sp

from oracle import model_np as mnp


class MaskSource(object):
    """Dropout randomness shared by two implementations: the first consumer draws and records,
    a `replay()` hands the same masks out again in call order."""

    def __init__(self, seed, keep):
        self.rng, self.keep, self.rec, self.pos, self.replaying = np.random.RandomState(seed), keep, [], 0, False

    def __call__(self, tag, shape):
        if self.replaying:
            m = self.rec[self.pos]
            self.pos += 1
            assert m.shape == tuple(shape), (tag, m.shape, shape)
            return m
        m = (self.rng.random_sample(shape) < self.keep).astype(np.float32)
        self.rec.append(m)
        return m

    def noise(self, tag, shape):
        """N(0, 1) draws for the Gaussian re-sampling of a det-dropout model, recorded / replayed like the masks."""
        if self.replaying:
            return self(tag, shape)
        z = self.rng.standard_normal(shape).astype(np.float32)
        self.rec.append(z)
        return z

    def replay(self):
        self.replaying, self.pos = True, 0
        return self


def placeholders(L, n_classes):
    class _Lab(object):
        shape = (None, n_classes)
    lab = _Lab()
    return {'adj': ['adj_%d' % i for i in range(L)], 'madj': ['madj_%d' % i for i in range(L)],
            'fadj': ['fadj_%d' % i for i in range(L)],
            'fields': ['field_%d' % i for i in range(L + 1)],
            'ffields': ['ffield_%d' % i for i in range(L)],
            'scales': ['scale_%d' % i for i in range(L)], 'labels': lab, 'dropout': 'dropout'}


def _graph(n, avg, seed, norm):
    from stochastic_gcn_amd import synthetic
    rng = np.random.RandomState(seed)
    a = synthetic.zipf_uniform_edges(n, n * avg // 2, 0.6, rng)
    return synthetic._gcn_normalize(a) if norm == 'gcn' else synthetic._row_normalize(a)


CASES = {
    # BASELINE config 3 in miniature: Reddit recipe (gcn/config/reddit.config:2) + --cv --cvd --degree=1
    'reddit_cvd_pp': dict(n=1500, avg=16, f=24, classes=5, sparse=False, model='vr', batch=64,
                          flags=dict(normalization='graphsage', weight_decay=0.0, dropout=0.2,
                                     layer_norm=True, hidden1=16, num_fc_layers=2, cv=True, cvd=True,
                                     degree=1, preprocess=True)),
    # BASELINE config 2: PubMed CVD+PP degree 1 (sparse first layer K9 + sparse dropout K12)
    'pubmed_cvd_pp': dict(n=1200, avg=6, f=60, classes=3, sparse=True, model='vr', batch=60,
                          flags=dict(normalization='gcn', hidden1=32, cv=True, cvd=True, degree=1,
                                     preprocess=True)),
    # BASELINE config 1: Cora exact (PlainGCN, degree 20, PP)
    'cora_exact': dict(n=800, avg=4, f=50, classes=7, sparse=True, model='plain', batch=140,
                       flags=dict(normalization='gcn', hidden1=32, degree=20, preprocess=True)),
    # plain CV (--cv without --cvd), CV+PP
    'reddit_cv_pp': dict(n=1500, avg=16, f=24, classes=5, sparse=False, model='vr', batch=64,
                         flags=dict(normalization='graphsage', weight_decay=1e-3, dropout=0.2,
                                    layer_norm=True, hidden1=16, num_fc_layers=2, cv=True, cvd=False,
                                    degree=2, preprocess=True)),
    # f-4: three layers with PP -> two aggregation layers, CVD (ADD layer between aggregators).
    # (CVD without PP is not runnable in the reference either: VRAggregator unpacks `h, mu =
    # inputs`, gcn/layers.py:299, which needs the ADD layer in front.)
    'cvd_pp_L3': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='vr', batch=40,
                      flags=dict(normalization='graphsage', dropout=0.3, layer_norm=True, hidden1=16,
                                 cv=True, cvd=True, degree=2, preprocess=True, num_layers=3)),
    # f-4: plain CV without PP, two aggregation layers reading two histories
    'cv_nopp_L2': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='vr', batch=40,
                       flags=dict(normalization='gcn', dropout=0.3, hidden1=16, cv=True, cvd=False,
                                  degree=2, preprocess=False)),
    # f-4: neighbour sampling without CV, two layers, gcn normalisation
    'ns_nopp_L2': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='plain', batch=40,
                       flags=dict(normalization='gcn', dropout=0.5, hidden1=16, degree=3,
                                  preprocess=False)),
    # --reverse ("original models", gcn/models.py:323-333): Dense -> Dropout -> aggregator, i.e. a
    # dropout whose consumer is NOT a Dense layer (the mask is materialised instead of fused)
    'ns_nopp_L2_reverse': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='plain', batch=40,
                               flags=dict(normalization='graphsage', dropout=0.4, hidden1=16, degree=3,
                                          preprocess=False, reverse=True, layer_norm=True)),
    # wide hidden layer (> 128): LayerNorm cannot ride in the GEMM epilogue (unfused LN path)
    'reddit_cvd_pp_wide': dict(n=700, avg=12, f=24, classes=5, sparse=False, model='vr', batch=48,
                               flags=dict(normalization='graphsage', dropout=0.2, layer_norm=True,
                                          hidden1=160, num_fc_layers=2, cv=True, cvd=True, degree=1,
                                          preprocess=True)),
    # f-4: importance-sampling baseline (IS+PP; gcn/scheduler.cpp:63-123, --importance gcn/train.py:61-62):
    # the sampler draws the layer's joint neighbourhood through the Fenwick multinomial and emits
    # importance-weighted edges; the model side is PlainGCN on those weights
    'is_pp': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='plain', batch=40,
                  flags=dict(normalization='gcn', dropout=0.3, hidden1=16, degree=4, preprocess=True,
                             importance=True)),
}


# --det_dropout (moment propagation, gcn/layers.py:141-202,236-248,320-349,425-428): its own golden file
# (tests/golden/model_steps_det.npz, make_model_golden.py --det; round 4), and the float64-autograd pin of the oracle's
# backward (tests/test_model_oracle.py) -- kept apart from CASES, whose golden file predates the variant
DET_CASES = {
    # CV + PP, three layers: DetDropoutFC -> VR aggregator on (mu, var) with two histories -> DetDropoutFC on a tuple
    # -> second aggregator -> Gaussian re-sampling + dropout -> Dense
    'det_cv_pp_L3': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='vr', batch=40,
                         flags=dict(normalization='graphsage', dropout=0.3, layer_norm=True, hidden1=16, cv=True,
                                    degree=2, preprocess=True, num_layers=3, det_dropout=True, weight_decay=1e-3)),
    # the Reddit recipe with two FC layers per block
    'det_cv_pp_fc2': dict(n=1200, avg=12, f=24, classes=5, sparse=False, model='vr', batch=48,
                          flags=dict(normalization='graphsage', dropout=0.2, layer_norm=True, hidden1=16,
                                     num_fc_layers=2, cv=True, degree=1, preprocess=True, det_dropout=True)),
    # neighbour sampling, no PP: the first aggregator sees plain features, the second one (mu, var); no LayerNorm
    'det_ns_nopp_L2': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='plain', batch=40,
                           flags=dict(normalization='gcn', dropout=0.5, hidden1=16, degree=3, preprocess=False,
                                      det_dropout=True)),
    # CV without PP, gcn normalisation
    'det_cv_nopp_L2': dict(n=900, avg=10, f=20, classes=4, sparse=False, model='vr', batch=40,
                           flags=dict(normalization='gcn', dropout=0.3, hidden1=16, cv=True, degree=2,
                                      preprocess=False, det_dropout=True, layer_norm=True)),
}


def make_scheduler(case, seed=1, data=None):
    """The product's sampler for a case (bit-exact against the reference C++, tests/test_sampler.py)."""
    from stochastic_gcn_amd.scheduler import PyScheduler
    fl = case['flags']
    return PyScheduler(case['adj'], case['labels'], case['L_sched'], [fl['degree']] * case['L_sched'], case['ph'],
                       seed, data=(case['train'] if data is None else data).copy(), cv=fl['cv'],
                       importance=bool(fl.get('importance', False)))


def planetoid_case(which):
    """BASELINE configs 1 / 2 at their SURVEY.md 8d sizes: S-Cora (N = 2,708, F = 1,433 sparse, 7 classes,
    exact PlainGCN, degree 20, one 140-id batch) and S-PubMed (N = 19,717, F = 500 sparse, 3 classes, CVD+PP
    degree 1, one 60-id batch) -- gcn/config/{cora,pubmed}.config flags with the README's CV switches."""
    from stochastic_gcn_amd import synthetic
    if which == 'cora':
        data = synthetic.cora_like('gcn', 123)
        flags = dict(normalization='gcn', hidden1=32, degree=20, preprocess=True, dropout=0.5, weight_decay=5e-4)
        cfg = dict(n=2708, classes=7, model='plain', batch=140, sparse=True)
    else:
        data = synthetic.pubmed_like('gcn', 123)
        flags = dict(normalization='gcn', hidden1=32, cv=True, cvd=True, degree=1, preprocess=True, dropout=0.5,
                     weight_decay=5e-4)
        cfg = dict(n=19717, classes=3, model='vr', batch=60, sparse=True)
    n, adj, _, feats, _, _, labels, tr, _, _ = data
    fl = mnp.make_flags(**flags)
    nbr = adj.dot(feats).tocsr().astype(np.float32)             # PP product (gcn/utils.py:169-170)
    nbr.sort_indices()
    return dict(cfg=cfg, flags=fl, adj=adj, feats=feats, nbr=nbr, labels=labels, train=tr.astype(np.int32),
                L_sched=1, ph=placeholders(1, cfg['classes']))


# BASELINE config 4's model at a size the NumPy oracle still finishes in seconds: the Reddit recipe (two hidden
# layers with LayerNorm, graphsage concat, CVD + PP, degree 1) on an S-Reddit-shaped graph of 12 k vertices; used by the
# two-rank tests only (it is not in CASES, whose entries all have golden vectors)
REDDIT_MID = dict(n=12000, avg=40, f=96, classes=41, sparse=False, model='vr', batch=256,
                  flags=dict(normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True, hidden1=64,
                             num_fc_layers=2, cv=True, cvd=True, degree=1, preprocess=True))


def build_case(name, seed=0):
    c = (CASES.get(name) or DET_CASES[name]) if isinstance(name, str) else name
    fl = mnp.make_flags(**c['flags'])
    rng = np.random.RandomState(seed)
    adj = _graph(c['n'], c['avg'], seed + 1, fl['normalization'])
    n = c['n']
    if c['sparse']:
        from stochastic_gcn_amd import synthetic
        feats = synthetic._sparse_features(n, c['f'], 6, rng)
        nbr = adj.dot(feats).tocsr().astype(np.float32)       # PP product (gcn/utils.py:169-170)
        nbr.sort_indices()
    else:
        feats = rng.standard_normal((n, c['f'])).astype(np.float32)
        nbr = adj.dot(feats).astype(np.float32)
    labels = np.zeros((n, c['classes']), np.float32)
    labels[np.arange(n), rng.randint(0, c['classes'], n)] = 1
    train = rng.permutation(n)[:c['batch'] * 3].astype(np.int32)
    L_sched = fl['num_layers'] - 1 if fl['preprocess'] else fl['num_layers']
    return dict(cfg=c, flags=fl, adj=adj, feats=feats, nbr=nbr, labels=labels, train=train,
                L_sched=L_sched, ph=placeholders(L_sched, c['classes']))


def make_oracle_model(case, params=None, is_training=True, seed=0):
    c, fl = case['cfg'], case['flags']
    probe = mnp.Model(fl, fl['num_layers'], fl['preprocess'], fl['cvd'], fl['cv'], case['feats'],
                      case['nbr'], c['n'], c['classes'], {}, is_training=is_training)
    if params is None:
        params = mnp.init_params(probe.specs, seed)
    return mnp.Model(fl, fl['num_layers'], fl['preprocess'], fl['cvd'], fl['cv'], case['feats'],
                     case['nbr'], c['n'], c['classes'], params, is_training=is_training)
