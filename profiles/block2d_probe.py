"""Would a small row block (an eighth of S-Reddit: 29 k rows, config 4's strong scaling at 8 GPUs) be better off with its
COLUMNS dealt to the XCDs?  A 1-D sweep makes every XCD read (most of) B: 8 x 0.79 x 561 MB = 3.5 GB for 70 MB of output, 0.62 ms.
With XCD x taking the nonzeros of column range x, B crosses the fabric once (561 MB) and every output row gets 8 partial sums.
Emulated with the shipped kernels: A' = the 8 column-restricted copies of the block stacked (8 M rows), a grouped plan (tiles inside
groups, consecutive tiles of a launch on the same XCD), C' = A' . B; the reduction of the 8 partials is NOT timed here.

    python profiles/block2d_probe.py [world]  ->  JSON lines (gpurun_out/block2d_probe.jsonl)
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic                      # noqa: E402
from stochastic_gcn_amd.parallel import partition_rows_by_nnz      # noqa: E402


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    d = 602
    dev = torch.device("cuda:0")
    n, _, a, *_ = synthetic.reddit_like(with_features=False)
    at = a.T.tocsr()
    b = partition_rows_by_nnz(a.indptr.astype(np.int64) + at.indptr.astype(np.int64), world)
    rank = min(3, world - 1)
    blk = a[int(b[rank]):int(b[rank + 1])].tocsr()
    M, K = blk.shape
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Xp = torch.zeros((K, 608), device=dev)
    Xp[:, :d] = torch.randn((K, d), device=dev, generator=g)
    X = Xp[:, :d]
    os.makedirs("gpurun_out", exist_ok=True)
    out = open(os.path.join("gpurun_out", "block2d_probe.jsonl"), "a")
    rec = dict(world=world, rows=M, nnz=int(blk.nnz), d=d)
    # the 1-D plan the product ships with
    A1 = ops.ColumnSweepCSR(blk, dev, G=ops.ColumnSweepCSR.choose_g(d, blk.nnz / M, M))
    A1.autotune(X)
    C = torch.empty((M, 608), device=dev)[:, :d]
    rec["ms_1d"] = round(timed(lambda: ops.spmm_cs(A1, X, out=C)), 4)
    rec["plan_1d"] = dict(G=A1.G, T=A1.T, pace=A1.pace.get(d))
    ref = C.clone()
    # 2-D: column range x -> XCD x
    for nx in (8, 16):
        coo = blk.tocoo()
        rng_id = (coo.col.astype(np.int64) * nx // K).astype(np.int64)
        stacked = sp.coo_matrix((coo.data, (coo.row + rng_id * M, coo.col)), shape=(nx * M, K)).tocsr()
        stacked.sort_indices()
        labels = np.repeat(np.arange(nx, dtype=np.int32), M)
        for T in (0, 16):
            A2 = ops.ColumnSweepCSR(stacked, dev, row_labels=labels, T=T)
            C2f = torch.empty((nx * M, 608), device=dev)
            C2 = C2f[:, :d]
            ms = timed(lambda: ops.spmm_cs(A2, X, out=C2))
            red = C2f.view(nx, M, 608)[:, :, :d].sum(dim=0)
            err = float((red - ref).abs().max() / ref.abs().max())
            rec["ms_2d_nx%d_T%d" % (nx, T)] = round(ms, 4)
            rec["tiles_nx%d_T%d" % (nx, T)] = int(A2.ntiles)
            rec["err_nx%d_T%d" % (nx, T)] = err
            del A2, C2, C2f
    out.write(json.dumps(rec) + "\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
