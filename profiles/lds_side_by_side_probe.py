"""LDS sweep + residual: one after the other on the whole chip vs side by side on split compute units
(hipExtStreamCreateWithCUMask streams: removed from the product after this measurement), for a few splits.  One JSON line per (p_in, mode)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic  # noqa: E402
from profiles.lds_probe import timed  # noqa: E402

dev = torch.device("cuda:0")
d = 602
for p_in in [float(x) for x in sys.argv[1:]] or [0.8]:
    n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=p_in)
    comm = labels.argmax(1).astype(np.int32)
    B = torch.zeros((n, 608), device=dev)
    B[:, :d] = torch.randn((n, d), device=dev)
    out = torch.empty((n, 608), device=dev)[:, :d]
    bytes_alg = a.nnz * 8 + (n + 1) * 4 + 2 * n * d * 4
    for mr in (2, 3):
        host = ops.LdsPlanHost(a, labels=comm, min_reuse=mr)
        for w in (0, 5, 6, 7):
            A = ops.LdsSweepCSR(a, dev, host=host, overlap_words=w)
            tuned = A.autotune(B[:, :d])
            t = timed(lambda: ops.spmm_lds(A, B[:, :d], out=out), reps=8)
            print(json.dumps({"p_in": p_in, "min_reuse": mr, "planned_cus_per_xcd": 4 * w if w else 32,
                              "residual_cus_per_xcd": 32 - 4 * w if w else 32, "mode": "side by side" if w else "one after the other",
                              "ms": round(t, 4), "frac": round(bytes_alg / (t * 1e-3) / 8e12, 4),
                              "local": round(A.host_stats["local_nnz"] / a.nnz, 3), "tuned": tuned}), flush=True)
