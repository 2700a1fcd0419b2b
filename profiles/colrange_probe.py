"""A small row block (an eighth / a quarter of S-Reddit: config 4's strong scaling) with its rows split BY COLUMN RANGE instead of
by stride, range j on XCDs [8 j / NR, 8 (j + 1) / NR), and the sweep still clock-paced.

Why: the 1-D sweep of an eighth sits on its fabric floor -- every XCD holds a random eighth of the block's rows, whose 362 k nonzeros
touch 79 % of B's 233 k rows: 8 x 0.79 x 561 MB = 3.5 GB at 7.2 TB/s = 0.49 ms of its 0.57 - 0.62.  The round-5 probe dealt the
columns to EIGHT ranges (B crosses once) and lost: 12 nonzeros per virtual row, 3.6 rounds of tiles, unpaced.  The block's rows are
split anyway (auto_t: T = 61 to fill the round), so split them in TWO column ranges instead: the same 58 k virtual rows, one round,
and an XCD's 362 k nonzeros fall on half of B: 8 x 0.48 x 561 MB = 2.15 GB.  Two-rate model: 0.46 ms against 0.60.

Emulated with the shipped kernels and NO library change: A' = the NR column-restricted copies of the block stacked, row labels = the
range (a grouped plan: tiles inside ranges, consecutive tiles of a launch on one XCD), a warp table that maps a column to its
position INSIDE its range scaled to [0, K) (all ranges sweep at once on one clock), and the plan struct patched to carry a pace next
to xcd_map.  The sum of the NR partial outputs is not timed (the 1-D plan's time includes its fix-up launch).

    python profiles/colrange_probe.py [world] [rank]  ->  JSON lines (gpurun_out/colrange_probe.jsonl)
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import _ffi, ops, synthetic                # noqa: E402
from stochastic_gcn_amd.parallel import ShardedSpMM, partition_rows_by_nnz      # noqa: E402


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


def paced_struct(A, d, pace, xcd_map):
    p = A.struct(d)
    p.pace_ns_per_nnz = int(pace)
    p.xcd_map = int(xcd_map)
    return p


def run_plan(A, X, out, pace, xcd_map=1):
    """spmm_cs with the plan struct patched (a grouped plan normally runs unpaced)"""
    import ctypes as C
    M, K = A.shape
    d = int(X.shape[1])
    plan = paced_struct(A, d, pace, xcd_map)
    ops.check(ops.lib.sgcn_spmm_cs_f32(C.byref(plan), M, K, d, X.data_ptr(), X.stride(0), None, None, None,
                                       out.data_ptr(), out.stride(0), 0.0, ops._stream()))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    d = 602
    dev = torch.device("cuda:0")
    n, _, a, *_ = synthetic.reddit_like(with_features=False)
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else min(3, world - 1)
    at = a.T.tocsr()
    load = a.indptr.astype(np.int64) + at.indptr.astype(np.int64) + np.arange(n + 1, dtype=np.int64) * (2 * ShardedSpMM.ROW_WEIGHT)
    bounds = partition_rows_by_nnz(load, world)
    blk = a[int(bounds[rank]):int(bounds[rank + 1])].tocsr()
    blk.sort_indices()
    M, K = blk.shape
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Xp = torch.zeros((K, 608), device=dev)
    Xp[:, :d] = torch.randn((K, d), device=dev, generator=g)
    X = Xp[:, :d]
    os.makedirs("gpurun_out", exist_ok=True)
    out = open(os.path.join("gpurun_out", "colrange_probe.jsonl"), "a")
    rec = dict(world=world, rank=rank, rows=M, nnz=int(blk.nnz), d=d)
    # the 1-D plan the product ships with
    A1 = ops.ColumnSweepCSR(blk, dev, G=ops.ColumnSweepCSR.choose_g(d, blk.nnz / M, M))
    A1.autotune(X)
    C1 = torch.empty((M, 608), device=dev)[:, :d]
    rec["ms_1d"] = round(timed(lambda: ops.spmm_cs(A1, X, out=C1)), 4)
    rec["plan_1d"] = dict(G=A1.G, T=A1.T, pace=A1.pace.get(d), tiles=int(A1.ntiles), nfix=int(A1.nfix), nslots=int(A1.nslots))
    ref = C1.clone()
    print(json.dumps(rec), flush=True)
    shift = 0
    while (K >> shift) > 16384:
        shift += 1
    bucket = 1 << shift
    colhist = np.bincount(blk.indices, minlength=K).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(colhist)])
    nrs = [int(x) for x in os.environ.get('NRS', '2,4').split(',')]
    ts = [int(x) for x in os.environ.get('TS', '0,32').split(',')]
    for nr in nrs:
        # range boundaries: equal nonzeros, on bucket boundaries
        cuts = [0]
        for j in range(1, nr):
            c = int(np.searchsorted(cum, cum[-1] * j / nr))
            cuts.append(min(K, (c + bucket // 2) // bucket * bucket))
        cuts.append(K)
        cuts = np.asarray(cuts, dtype=np.int64)
        coo = blk.tocoo()
        rng_id = (np.searchsorted(cuts, coo.col, side="right") - 1).astype(np.int64)
        stacked = sp.coo_matrix((coo.data, (coo.row + rng_id * M, coo.col)), shape=(nr * M, K)).tocsr()
        stacked.sort_indices()
        labels = np.repeat(np.arange(nr, dtype=np.int32), M)
        # the clock's coordinates: position inside the column's range, scaled to [0, K)
        nb = -(-K // bucket)
        first = np.arange(nb, dtype=np.int64) * bucket
        rj = np.searchsorted(cuts, first, side="right") - 1
        lo, hi = cuts[rj], cuts[rj + 1]
        table = ((first - lo) * K // np.maximum(hi - lo, 1)).astype(np.uint32)
        for T in ts:
            A2 = ops.ColumnSweepCSR(stacked, dev, row_labels=labels, T=T)
            A2.warp = torch.from_numpy(table.view(np.int32)).to(dev)
            A2.warp_shift = shift
            C2f = torch.empty((nr * M, 608), device=dev)
            C2 = C2f[:, :d]
            key = "nr%d_T%d" % (nr, T)
            res = {}
            for xm in (1, 0):
                for pace in (-1, 220, 240, 250, 260, 270, 280, 290, 300, 320, 360):
                    ms = timed(lambda: run_plan(A2, X, C2, pace, xm), reps=6)
                    res["xcd%d_p%d" % (xm, pace)] = round(ms, 4)
            best = min((v, k) for k, v in res.items() if k.startswith("xcd1"))
            best0 = min((v, k) for k, v in res.items() if k.startswith("xcd0"))
            run_plan(A2, X, C2, int(best[1].split("_p")[1]), 1)
            red = C2f.view(nr, M, 608)[:, :, :d].sum(dim=0)
            err = float((red - ref).abs().max() / ref.abs().max())
            rec2 = dict(rec, form=key, tiles=int(A2.ntiles), nfix=int(A2.nfix), cuts=[int(x) for x in cuts], best_xcdmap=best,
                        best_roundrobin=best0, err=err, table=res)
            out.write(json.dumps(rec2) + "\n")
            out.flush()
            print(json.dumps({k: rec2[k] for k in ("form", "tiles", "nfix", "best_xcdmap", "best_roundrobin", "err")}), flush=True)
            del A2, C2, C2f


if __name__ == "__main__":
    main()
