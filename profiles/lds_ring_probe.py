"""Every column staged (min_reuse 1): two ring parts of 128 slots (requests from inside the chunk statement, one chunk time to
land) against three parts of 80 (two chunk times, request burst in front of the statement) -- the single-use pieces of an
all-staged plan come across the fabric, and their latency shows as fill wait with two parts.
usage: python profiles/lds_ring_probe.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic
dev = torch.device("cuda:0"); d = 602
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for p_in in (0.95, 0.9, 0.8):
    n, _, a, *_ = synthetic.reddit_sbm(p_in=p_in)
    lab, _ = ops.reorder_labels(a)
    B = torch.zeros((n, 608), device=dev); B[:, :d] = torch.randn((n, d), device=dev)
    out = torch.empty((n, 608), device=dev)[:, :d]
    for ring in (128, 80):
        A = ops.LdsSweepCSR(a, dev, labels=lab, min_reuse=1, ring_slots=ring)
        print(p_in, "ring", ring, "chunks", A.nchunks, "ms %.3f" % t(lambda: ops.spmm_lds(A, B[:, :d], out=out)), flush=True)
