"""What is left on the S-Reddit headline product: the sweep clock around the autotuner's choice, the
clock of the 90-column fifth pass (cs_last_pct), the alignment of the two bins, and the share of the fix-up launch --
sustained forward products (20 in a row) per setting.  One JSON line per setting."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic, _ffi  # noqa: E402

dev = torch.device("cuda:0")
d = 602
n, _, a, *_ = synthetic.reddit_like(with_features=False)
B = torch.zeros((n, 608), device=dev)
B[:, :d] = torch.randn((n, d), device=dev)
out = torch.empty((n, 608), device=dev)[:, :d]


def sustained(A, reps=20):
    ops.spmm_cs(A, B[:, :d], out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.spmm_cs(A, B[:, :d], out=out)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


for align in (2048, 1024):
    A = ops.ColumnSweepCSR(a, dev, G=2, align=align)
    A._tuning = True                      # (no guard samples, no re-tunes while the knobs move)
    t, pace = A.autotune(B[:, :d])
    A._tuning = True
    print(json.dumps({"align": align, "autotuned_pace": pace, "ms": round(t, 4), "pad_fraction": round(A.pad_fraction, 4)}), flush=True)
    for p in (pace - 12, pace - 8, pace - 4, pace, pace + 4, pace + 8):
        A.pace[d] = p
        print(json.dumps({"align": align, "pace": p, "last_pct": 90, "ms": round(sustained(A), 4)}), flush=True)
    A.pace[d] = pace
    for pct in (70, 80, 85, 95, 100):
        _ffi.tune("cs_last_pct", pct)
        print(json.dumps({"align": align, "pace": pace, "last_pct": pct, "ms": round(sustained(A), 4)}), flush=True)
    _ffi.tune("cs_last_pct", 90)
# the fix-up launch: the same plan with T so large that no row is split (one hub row per tile then sets the launch time)
A = ops.ColumnSweepCSR(a, dev, G=2)
A._tuning = True
print(json.dumps({"split_rows": int(A.nfix), "note": "cs_fix_kernel: 37 us per product (profiles/r08_rocprof_summary.txt) = 1.1 %"}), flush=True)
