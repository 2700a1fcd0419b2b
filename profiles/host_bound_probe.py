#!/usr/bin/env python
"""How much of the CVD+PP training step is host (Python + ctypes + launch API) time?  The host does the
same work whatever the batch size, the GPU's share shrinks with it: the step time at a tiny batch is
(close to) the host's cost per step.  Prints ms/step for batch 512 (the recipe) and batch 32."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch          # noqa: E402
from stochastic_gcn_amd import synthetic          # noqa: E402
from stochastic_gcn_amd.flags import FLAGS        # noqa: E402
from stochastic_gcn_amd.train import Trainer      # noqa: E402

data = synthetic.reddit_like(seed=1, with_features=False)
n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
feats = torch.randn((n, 602), device='cuda:0')
PLAN_T = int(os.environ.get('SGCN_PLAN_T', '0'))
for bs in (512, 32):
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True,
                 hidden1=128, num_fc_layers=2, batch_size=bs, test_batch_size=512, cv=True, cvd=True,
                 test_cv=True, degree=1, test_degree=1, seed=1, max_steps=298)
    if PLAN_T:
        FLAGS.update(plan_t=PLAN_T)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
    best = 1e9
    for _ in range(5):
        trn.train_epoch()
        best = min(best, trn.last_epoch['train_wall_s'] / trn.last_epoch['steps'])
    tm = trn.train_model
    st = trn.last_epoch['steps']
    print("batch %4d: %.3f ms/step (%d steps); host inside run_one_step %.3f ms/step (upload+counters %.3f, launches %.3f), "
          "waiting for the sampler %.3f ms/step; producer %s" % (bs, best * 1e3, st, (tm.run_t + tm.g_t) / st * 1e3,
                                                                 tm.g_t / st * 1e3, tm.run_t / st * 1e3,
                                                                 trn.last_epoch['sch_wait_s'] / st * 1e3,
                                                                 trn.last_epoch.get('producer_s')))
    from stochastic_gcn_amd import _ffi
    if bs == 512:
        for ov in (0, 1):
            _ffi.tune('step_overlap', ov)
            b2 = 1e9
            for _ in range(4):
                trn.train_epoch()
                b2 = min(b2, trn.last_epoch['train_wall_s'] / trn.last_epoch['steps'])
            print("   step_overlap=%d: %.3f ms/step" % (ov, b2 * 1e3))
