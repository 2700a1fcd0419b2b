"""How fast is the C++ sampler thread alone vs next to the launching thread?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stochastic_gcn_amd import synthetic
from stochastic_gcn_amd.flags import FLAGS
from stochastic_gcn_amd.train import Trainer, epoch_batches
from stochastic_gcn_amd.scheduler import NativePrefetcher
data = synthetic.reddit_like(seed=1, with_features=False)
FLAGS.reset()
FLAGS.update(dataset='reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True,
             hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512, cv=True, cvd=True,
             test_cv=True, degree=1, test_degree=1, seed=1)
n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
feats = torch.randn((n, 602), device='cuda:0')
trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
sch = trn.train_sch
batches = epoch_batches(sch.data, 512, 298)
for rep in range(2):
    pre = NativePrefetcher(sch, batches, 0, depth=2)
    t0 = time.perf_counter(); k = 0
    while True:
        pb = pre.next()
        if pb is None: break
        k += 1
    print("producer alone: %.3f ms per batch" % ((time.perf_counter() - t0) / k * 1e3))
# spin the main thread with pure-Python work while draining
pre = NativePrefetcher(sch, batches, 0, depth=2)
t0 = time.perf_counter(); k = 0; x = 0
while True:
    pb = pre.next()
    if pb is None: break
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 0.0004: x += 1       # 0.4 ms of interpreter work per batch
    k += 1
print("with a busy Python consumer (0.4 ms/batch): %.3f ms per batch" % ((time.perf_counter() - t0) / k * 1e3))
for _ in range(3):
    trn.train_epoch(); print(trn.last_epoch['train_wall_s'], trn.last_epoch['sch_wait_s'])
print(os.sched_getaffinity(0).__len__(), "cpus in affinity;", open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max')
