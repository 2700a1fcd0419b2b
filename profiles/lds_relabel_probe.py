"""Does the placement of B's rows matter to the ring fills?  The same S-Reddit-SBM product with the vertices (a) as the
generator leaves them -- a community's rows scattered over the whole 561 MB operand -- and (b) RELABELLED community by
community (rows and columns of the adjacency permuted alike, i.e. the dataset renumbered once): the LDS sweep's planned
part, full / fills only / neither (lds_dbg knob), and the two-lane-group column sweep on the same matrix."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic, _ffi  # noqa: E402
from profiles.lds_probe import timed  # noqa: E402

dev = torch.device("cuda:0")
d = 602
p_in = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
n, _, a0, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=p_in)
comm0 = labels.argmax(1).astype(np.int32)
B = torch.zeros((n, 608), device=dev)
B[:, :d] = torch.randn((n, d), device=dev)
out = torch.empty((n, 608), device=dev)[:, :d]
for name in ("scattered", "relabelled"):
    if name == "scattered":
        a, comm = a0, comm0
    else:
        perm = np.argsort(comm0, kind="stable")
        a = a0[perm][:, perm].tocsr()
        a.sort_indices()
        comm = comm0[perm]
    for mr in (2, 3):
        A = ops.LdsSweepCSR(a, dev, labels=comm, min_reuse=mr)
        if A.residual is not None:
            A.autotune(B[:, :d])
        rec = {"p_in": p_in, "vertices": name, "min_reuse": mr, "reuse": round(A.host_stats["reuse"], 2),
               "local": round(A.host_stats["local_nnz"] / a.nnz, 3)}
        for dbg, key in ((0, "planned_ms"), (2, "fills_only_ms"), (3, "neither_ms")):
            _ffi.tune("lds_dbg", dbg)
            rec[key] = round(timed(lambda: ops.spmm_lds(A, B[:, :d], out=out, local_only=True)), 4)
        _ffi.tune("lds_dbg", 0)
        rec["all_ms"] = round(timed(lambda: ops.spmm_lds(A, B[:, :d], out=out)), 4)
        rec["residual_ms"] = round(rec["all_ms"] - rec["planned_ms"], 4)
        print(json.dumps(rec), flush=True)
    cs = ops.ColumnSweepCSR(a, dev, G=2)
    t, pace = cs.autotune(B[:, :d])
    print(json.dumps({"p_in": p_in, "vertices": name, "column_sweep_G2_ms": round(timed(lambda: ops.spmm_cs(cs, B[:, :d], out=out)), 4), "pace": pace}), flush=True)
