"""The column sweep's clock in work coordinates (sgcn_csplan_t.dev_warp) on one GPU's block of S-RMAT 10 M / 200 M (config 5):
forward product time per sweep clock, for plans with / without the warp table and different bin alignments.

    python profiles/warp_probe.py [rank/world] [d] [G]  ->  JSON lines (gpurun_out/warp_probe.jsonl)
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic                      # noqa: E402
from stochastic_gcn_amd.parallel import partition_rows_by_nnz      # noqa: E402


def main():
    rw = sys.argv[1] if len(sys.argv) > 1 else "3/8"
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rank, world = (int(x) for x in rw.split("/"))
    dev = torch.device("cuda:0")
    a = synthetic.cached_graph("rmat_10m_200m_seed1", lambda: synthetic.rmat_like(10_000_000, 200_000_000, seed=1))
    at = a.T.tocsr()
    b = partition_rows_by_nnz(a.indptr.astype(np.int64) + at.indptr.astype(np.int64), world)
    blk = a[int(b[rank]):int(b[rank + 1])]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    X = torch.rand((a.shape[1], d), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
    out_path = os.path.join("gpurun_out", "warp_probe.jsonl")
    os.makedirs("gpurun_out", exist_ok=True)
    variants = [dict(align=2048, warp=False), dict(align='auto', warp=True), dict(align=20000, warp=True),
                dict(align=300000, warp=True), dict(align='auto', warp=False)]
    if G == 4:
        variants = [dict(align=2048, warp=False), dict(align='auto', warp=True), dict(align=300000, warp=True)]
    paces = (-1, 200, 240, 280, 320, 360, 400, 450, 500, 560) if G == 2 else (-1, 240, 320, 400, 480, 560, 640, 720, 800, 900)
    with open(out_path, "a") as f:
        for kw in variants:
            A = ops.ColumnSweepCSR(blk, dev, G=G, **kw)
            C = torch.empty((blk.shape[0], d), dtype=torch.float32, device=dev)
            rec = dict(block=rw, d=d, G=G, nnz=int(blk.nnz), rows=int(blk.shape[0]), align=A.align, warp=A.warp is not None,
                       pad_fraction=round(A.pad_fraction, 4), ms={})
            for p in paces:
                A.pace[d] = p
                for _ in range(2):
                    ops.spmm_cs(A, X, out=C, d=d)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    ops.spmm_cs(A, X, out=C, d=d)
                e1.record()
                e1.synchronize()
                rec["ms"][str(p)] = round(e0.elapsed_time(e1) / 6, 4)
            f.write(json.dumps(rec) + "\n")
            f.flush()
            print(json.dumps(rec), flush=True)
            del A


if __name__ == "__main__":
    main()
