"""Forward (one value per row) and transposed (one value per column -> unit plan + row scale of the operand) LDS products on\nS-Reddit-SBM: total, planned part, residual.  usage: python profiles/lds_colfold_probe.py"""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic
dev = torch.device("cuda:0"); d = 602
n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=0.8)
comm = labels.argmax(1).astype(np.int32)
at = a.T.tocsr().astype(np.float32); at.sort_indices()
B = torch.zeros((n, 608), device=dev); B[:, :d] = torch.randn((n, d), device=dev)
out = torch.empty((n, 608), device=dev)[:, :d]
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for name, m in (("A", a), ("At", at)):
    A = ops.LdsSweepCSR(m, dev, labels=comm, min_reuse=3)
    A.autotune(B[:, :d])
    print(name, "unit", A.unit, "col_fold", A.col_fold is not None, "stats", A.host_stats["local_nnz"], A.host_stats["nent"],
          "total %.3f local %.3f residual %.3f" % (t(lambda: ops.spmm_lds(A, B[:, :d], out=out)), t(lambda: ops.spmm_lds(A, B[:, :d], out=out, local_only=True)),
           t(lambda: ops.spmm_cs(A.residual, B[:, :d], out=out, beta=1.0))), "residual G", A.residual.G, "pace", A.residual.pace)
