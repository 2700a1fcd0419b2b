import cProfile, pstats, sys, os, io, contextlib, time
sys.path.insert(0, '/root/repo')
import torch
from stochastic_gcn_amd import synthetic
from stochastic_gcn_amd.flags import FLAGS
from stochastic_gcn_amd.train import Trainer
data = synthetic.reddit_like(seed=1, with_features=False)
FLAGS.reset()
FLAGS.update(dataset='reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True,
             hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512, cv=True, cvd=True,
             test_cv=True, degree=1, test_degree=1, seed=1)
n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
feats = torch.randn((n, 602), device='cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
trn.train_epoch()
for native in (True, False):
    FLAGS.update(native_step=native)
    trn.evaluate(trn.val_d); trn.evaluate(trn.val_d)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0=time.perf_counter(); trn.evaluate(trn.val_d); torch.cuda.synchronize(); el=time.perf_counter()-t0
    pr.disable()
    print("native", native, "sweep", el)
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
