"""Does the training epoch slow down when validation runs between epochs?  Prints train_wall_s per epoch."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stochastic_gcn_amd import synthetic
from stochastic_gcn_amd.flags import FLAGS
from stochastic_gcn_amd.train import Trainer
data = synthetic.reddit_like(seed=1, with_features=True, planted=True)
FLAGS.reset()
FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True,
             hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512, cv=True, cvd=True,
             test_cv=True, degree=1, test_degree=1, seed=1)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Trainer(data=data, verbose=False)
for phase, do_eval in (("train only", False), ("with validation", True), ("train only again", False)):
    ts = []
    for ep in range(5):
        tr.train_epoch()
        ts.append(round(tr.last_epoch['train_wall_s'], 4))
        if do_eval:
            with contextlib.redirect_stdout(io.StringIO()):
                tr.evaluate(tr.val_d)
    print(phase, ts, "sch_wait", round(tr.last_epoch['sch_wait_s'], 4), flush=True)
