"""The LDS sweep's residual (S-Reddit-SBM, p_in 0.8: the 26 % of the nonzeros whose column a tile uses < 3 times) on the
column sweep: two lane groups (two rounds of resident tiles) against four (one round), by bin alignment; autotuned clock,
sustained time, pad share.  One JSON line per setting."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
d = 602
p_in = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=p_in)
comm = labels.argmax(1).astype(np.int32)
host = ops.LdsPlanHost(a, labels=comm, min_reuse=3)
res = host.residual
B = torch.zeros((n, 608), device=dev)
B[:, :d] = torch.randn((n, d), device=dev)
out = torch.zeros((n, 608), device=dev)[:, :d]


def sustained(A, reps=10):
    ops.spmm_cs(A, B[:, :d], out=out, beta=1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.spmm_cs(A, B[:, :d], out=out, beta=1.0)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


print(json.dumps({"residual_nnz": int(res.nnz), "nnz": int(a.nnz)}), flush=True)
CASES = [(2, 2048, 0), (2, 8192, 0), (4, 2048, 0), (4, 4096, 0), (4, 8192, 0), (4, 16384, 0), (4, 32768, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "T":          # the split threshold of long rows (0: the plan's default, 4 x the mean row)
    CASES = [(4, 8192, 0), (4, 8192, 128), (4, 8192, 256), (4, 8192, 512), (4, 8192, 2048)]
for G, align, Tsplit in CASES:
    A = ops.ColumnSweepCSR(res, dev, G=G, align=align, T=Tsplit)
    A._tuning = True
    t, pace = A.autotune(B[:, :d])
    A._tuning = True
    steps = int(A._tile_nnz.max())
    print(json.dumps({"G": G, "align": align, "T": Tsplit, "split_rows": int(A.nfix), "pad_fraction": round(A.pad_fraction, 4), "steps_heaviest_tile": steps,
                      "autotuned_pace": pace, "autotune_ms": round(t, 4), "sustained_ms": round(sustained(A), 4)}), flush=True)
