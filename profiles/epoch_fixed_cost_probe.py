"""The CVD+PP minibatch epoch's FIXED cost: wall time of epochs of 298 / 149 / 75 / 38 steps (max_steps) on one GPU -- an 8-GPU
data-parallel epoch is 38 steps per rank, so whatever an epoch costs besides its steps (sampler thread start, shuffle, the end-of-
epoch synchronisation) decides whether 8 GPUs give 6 x.  With SGCN_FORCE_PG=1 the one-rank job takes every collective path (RCCL).

    [SGCN_FORCE_PG=1] python profiles/epoch_fixed_cost_probe.py  ->  JSON lines
"""
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import synthetic                     # noqa: E402
from stochastic_gcn_amd.flags import FLAGS                   # noqa: E402
from stochastic_gcn_amd.train import Trainer                 # noqa: E402


def main():
    dev = torch.device("cuda:0")
    data = synthetic.reddit_like(with_features=False)
    n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2,
                 layer_norm=True, hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512,
                 cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, seed=1)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    feats = torch.randn((n, 602), device=dev, generator=g)
    with contextlib.redirect_stdout(sys.stderr):
        trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
    pts = []
    for steps in (298, 149, 75, 38, 38, 75, 149, 298):
        FLAGS.update(max_steps=steps)
        walls, outer = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trn.train_epoch()
            outer.append(time.perf_counter() - t0)
            walls.append(trn.last_epoch['train_wall_s'])
        rec = dict(steps=trn.last_epoch['steps'], best_wall_ms=round(min(walls) * 1e3, 3), best_outer_ms=round(min(outer) * 1e3, 3),
                   walls_ms=[round(w * 1e3, 2) for w in walls], host_loop_ms=round(trn.last_epoch['host_loop_s'] * 1e3, 3),
                   forced_pg=os.environ.get("SGCN_FORCE_PG") == "1")
        pts.append((rec['steps'], rec['best_outer_ms']))
        print(json.dumps(rec), flush=True)
    x, y = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts], float)
    A = np.stack([x, np.ones_like(x)], 1)
    (per_step, fixed), *_ = np.linalg.lstsq(A, y, rcond=None)
    print(json.dumps(dict(fit_ms_per_step=round(float(per_step), 5), fit_fixed_ms_per_epoch=round(float(fixed), 3),
                          note="outer = shuffle + train_epoch incl. the final synchronisation")), flush=True)


if __name__ == "__main__":
    main()
