#!/usr/bin/env python
"""cProfile of the main (launching) thread over one CVD+PP training epoch: where the host time of
the ~0.6 ms step goes once the GPU side is no longer the bound."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch          # noqa: E402
from stochastic_gcn_amd import synthetic          # noqa: E402
from stochastic_gcn_amd.flags import FLAGS        # noqa: E402
from stochastic_gcn_amd.train import Trainer      # noqa: E402

data = synthetic.reddit_like(seed=1, with_features=False)
FLAGS.reset()
FLAGS.update(dataset='reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True,
             hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512, cv=True, cvd=True,
             test_cv=True, degree=1, test_degree=1, seed=1)
n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
feats = torch.randn((n, 602), device='cuda:0')
trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
trn.train_epoch()
trn.train_epoch()
pr = cProfile.Profile()
pr.enable()
trn.train_epoch()
pr.disable()
print(trn.last_epoch)
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(32)
