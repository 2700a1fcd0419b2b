"""BASELINE configs 1 and 2 (sparse input features) as step programs against the eager per-layer path: wall time per
training epoch (a handful of batches on these graphs; 25 epochs, the first 5 dropped), same seeds.
usage: python profiles/sparse_epoch_probe.py"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd.flags import FLAGS  # noqa: E402
from stochastic_gcn_amd.train import Trainer  # noqa: E402

CASES = (("s-pubmed CVD+PP degree 1", dict(dataset='s-pubmed', cv=True, cvd=True, preprocess=True, degree=1, test_degree=1, test_cv=True)),
         ("s-cora exact (PlainGCN, degree 20)", dict(dataset='s-cora', cv=False, cvd=False, preprocess=True, degree=20, test_degree=20)))
for name, kw in CASES:
    res = {}
    for native in (True, False):
        FLAGS.reset()
        FLAGS.update(normalization='gcn', weight_decay=5e-4, dropout=0.5, layer_norm=False, hidden1=32, num_fc_layers=1,
                     batch_size=1000, test_batch_size=1000, seed=1, native_step=native, **kw)
        trn = Trainer(verbose=False)
        times = []
        for ep in range(25):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trn.train_epoch()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        m = trn.train_model
        nb = -(-len(trn.train_d) // FLAGS.batch_size)
        res["program" if native else "eager"] = {"ms_per_epoch": round(1e3 * float(np.median(times[5:])), 3), "batches_per_epoch": nb,
                                                  "ms_per_step": round(1e3 * float(np.median(times[5:])) / nb, 3),
                                                  "programs": sum(p is not None for p in getattr(m, '_programs', {}).values()),
                                                  "note": getattr(m, '_program_note', None)}
    print(json.dumps({"config": name, **res}), flush=True)
