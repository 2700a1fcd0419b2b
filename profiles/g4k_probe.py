"""Four lane groups per wave (one round of resident tiles) on the S-Reddit product: time against the sweep clock, with
and without the arithmetic (lds_dbg bit 1: gathers + pacing only), by bin alignment.  One JSON line per setting.
Needs the kernel of commit af99d18 (cs_spmm16g4k_kernel, removed after this measurement: profiles/r29_g4k_probe.jsonl)."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic, _ffi  # noqa: E402

dev = torch.device("cuda:0")
d = 602
n, _, a, *_ = synthetic.reddit_like(with_features=False)
B = torch.zeros((n, 608), device=dev)
B[:, :d] = torch.randn((n, d), device=dev)
out = torch.empty((n, 608), device=dev)[:, :d]


def sustained(A, reps=10):
    ops.spmm_cs(A, B[:, :d], out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.spmm_cs(A, B[:, :d], out=out)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


for G, align in ((4, 2048), (4, 4096), (2, 2048)):
    A = ops.ColumnSweepCSR(a, dev, G=G, align=align)
    A._tuning = True
    for dbg in (0, 2):
        _ffi.tune("lds_dbg", dbg)
        row = {}
        for p in (-1, 120, 140, 160, 170, 180, 190, 200, 210, 220, 240, 260):
            A.pace[d] = p
            row[p] = round(sustained(A), 3)
        print(json.dumps({"G": G, "align": align, "pad_fraction": round(A.pad_fraction, 4), "no_arithmetic": bool(dbg), "ms_by_pace": row}), flush=True)
    _ffi.tune("lds_dbg", 0)
