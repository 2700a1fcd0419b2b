"""A few LDS-sweep products on S-Reddit-SBM for a profiler to look at (profiles/lds_pmc.sh).
usage: python profiles/lds_run.py [p_in] [min_reuse] [reps] [local|all]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic  # noqa: E402

p_in = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
mr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
local = (sys.argv[4] if len(sys.argv) > 4 else "local") == "local"
dev = torch.device("cuda:0")
d = 602
n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=p_in)
comm = labels.argmax(1).astype(np.int32)
B = torch.zeros((n, 608), device=dev)
B[:, :d] = torch.randn((n, d), device=dev)
out = torch.empty((n, 608), device=dev)[:, :d]
A = ops.LdsSweepCSR(a, dev, labels=comm, min_reuse=mr)
if A.residual is not None and not local:
    A.residual.pace[d] = 232 if A.residual.G == 4 else 273
for _ in range(reps):
    ops.spmm_lds(A, B[:, :d], out=out, local_only=local)
torch.cuda.synchronize()
print("nnz %d local %d staged %d" % (A.nnz, A.host_stats["local_nnz"], A.host_stats["staged"]))
