"""Where the LDS sweep's waves spend their cycles: per-phase cycle counters of every wave (sgcn_lds_profile_buffer),
summed per phase and as a share of the wave's life, for one product on S-Reddit-SBM."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic, _ffi  # noqa: E402

dev = torch.device("cuda:0")
d = 602
p_in = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
mr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=p_in)
comm = labels.argmax(1).astype(np.int32)
B = torch.zeros((n, 608), device=dev)
B[:, :d] = torch.randn((n, d), device=dev)
out = torch.empty((n, 608), device=dev)[:, :d]
A = ops.LdsSweepCSR(a, dev, labels=comm, min_reuse=mr, residual_G=0)
blocks = 8 * max(A.xcd_tile_ptr[x + 1] - A.xcd_tile_ptr[x] for x in range(8)) * 5
buf = torch.zeros(blocks * 8 * 8, dtype=torch.int64, device=dev)
nblocks = blocks
for _ in range(2):
    ops.spmm_lds(A, B[:, :d], out=out, local_only=True)
_ffi.check(_ffi.lib.sgcn_lds_profile_buffer(buf.data_ptr()))
ops.spmm_lds(A, B[:, :d], out=out, local_only=True)
torch.cuda.synchronize()
_ffi.check(_ffi.lib.sgcn_lds_profile_buffer(None))
p = buf.cpu().numpy().reshape(blocks, 8, 8).astype(np.float64)
xcd_busy = [float(p[x::8, 0, 6].sum()) for x in range(8)]          # workgroup b runs on XCD b % 8: cycles of its items, summed
p = p[p[:, 0, 6] > 0]                                   # workgroups that ran an item
names = ["prologue", "fill_issue", "chunk_statements", "fill_wait", "barrier", "epilogue", "all"]
tot = p[:, :, 6].sum()
rec = {"items": int(p.shape[0]), "chunks_per_item": float(p[:, 0, 7].mean()),
       "cycles_per_item_mean": float(p[:, :, 6].mean()), "cycles_per_item_max": float(p[:, :, 6].max()),
       "share": {nm: round(float(p[:, :, i].sum() / tot), 4) for i, nm in enumerate(names[:6])},
       "cycles_per_chunk": {nm: round(float(p[:, :, i].sum() / p[:, :, 7].sum()), 1) for i, nm in enumerate(names[1:5], start=1)}}
rec["xcd_busy_Mcycles"] = [round(x / 1e6, 2) for x in xcd_busy]
rec["by_wave"] = {nm: [round(float(p[:, w, i].mean()), 0) for w in range(8)] for i, nm in ((2, "chunk_statements"), (4, "barrier"), (1, "fill_issue"), (5, "epilogue"))}
print(json.dumps(rec))
