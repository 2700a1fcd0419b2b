"""The LDS plan's residual through the row-gather kernel (one launch, no lock-step) against the four-group column sweep:
p_in 0.95 (1.8 M residual nonzeros) 2.083 vs 2.102 ms per product, p_in 0.8 (5.8 M) 3.22 vs 2.44 -- the sweep stays.
usage: python profiles/lds_residual_rows_probe.py"""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic
dev = torch.device("cuda:0"); d = 602
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for p_in in (0.95, 0.8):
    n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=p_in)
    lab, _ = ops.reorder_labels(a)
    B = torch.zeros((n, 608), device=dev); B[:, :d] = torch.randn((n, d), device=dev)
    out = torch.empty((n, 608), device=dev)[:, :d]
    host = ops.LdsPlanHost(a, labels=lab, min_reuse=3)
    for G in (4, 0):
        A = ops.LdsSweepCSR(a, dev, host=host, residual_G=G)
        if G: A.autotune(B[:, :d])
        print(p_in, "residual_G", G, "res nnz", host.residual.nnz, "total %.3f" % t(lambda: ops.spmm_lds(A, B[:, :d], out=out)))
