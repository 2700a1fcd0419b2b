import os, sys, time, contextlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from stochastic_gcn_amd import synthetic, scheduler
from stochastic_gcn_amd.flags import FLAGS
from stochastic_gcn_amd.train import Trainer
data = synthetic.reddit_like(seed=1, with_features=False)
n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
feats = torch.randn((n, 602), device='cuda:0')
def run(prefetch, lag, timing=False):
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True,
                 hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512, cv=True, cvd=True,
                 test_cv=True, degree=1, test_degree=1, seed=1, prefetch=prefetch)
    orig = scheduler.NativePrefetcher.__init__
    def init(self, sch, batches, plan_T=0, depth=2, pin=True, lag_=lag):
        orig(self, sch, batches, plan_T, depth, pin, lag_)
    scheduler.NativePrefetcher.__init__ = init
    waits = [0.0, 0.0]
    if timing:
        onext = scheduler.NativePrefetcher.next
        oswait = scheduler.StagingSlot.wait
        def swait(self):
            t = time.perf_counter(); r = oswait(self); waits[0] += time.perf_counter() - t; return r
        scheduler.StagingSlot.wait = swait
    with contextlib.redirect_stdout(sys.stderr):
        trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
    best = 1e9
    for _ in range(5):
        waits[0] = 0.0
        trn.train_epoch()
        best = min(best, trn.last_epoch['train_wall_s'] / trn.last_epoch['steps'])
    scheduler.NativePrefetcher.__init__ = orig
    if timing:
        scheduler.StagingSlot.wait = oswait
    print("prefetch=%d lag=%d: %.3f ms/step; sch_wait %.3f ms/step; of it slot-event waits %.3f ms/step; producer %s"
          % (prefetch, lag, best * 1e3, trn.last_epoch['sch_wait_s'] / 298 * 1e3, waits[0] / 298 * 1e3, trn.last_epoch['producer_s']))
run(2, 2, True)
run(4, 4, True)
run(8, 6, True)
