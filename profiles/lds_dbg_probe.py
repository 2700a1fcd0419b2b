import json, sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic, _ffi
from profiles.lds_probe import timed
dev = torch.device("cuda:0"); d = 602
n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=0.8)
comm = labels.argmax(1).astype(np.int32)
B = torch.zeros((n, 608), device=dev); B[:, :d] = torch.randn((n, d), device=dev); Bd = B[:, :d]
out = torch.empty((n, 608), device=dev)[:, :d]
for mr, slots in ((2, 80), (2, 128), (3, 80), (3, 128)):
    A = ops.LdsSweepCSR(a, dev, labels=comm, min_reuse=mr, residual_G=0, ring_slots=slots)
    for dbg in (0, 2):
        _ffi.tune("lds_dbg", dbg)
        t = timed(lambda: ops.spmm_lds(A, Bd, out=out, local_only=True))
        print(json.dumps({"min_reuse": mr, "ring_slots": slots, "chunks": A.nchunks, "dbg": dbg, "ms": round(t, 4)}), flush=True)
    _ffi.tune("lds_dbg", 0)
    for dd in ():
        t = timed(lambda: ops.spmm_lds(A, B[:, :dd], out=out[:, :dd], local_only=True))
        print(json.dumps({"min_reuse": mr, "d": dd, "ms": round(t, 4)}), flush=True)
