"""Kernel durations of the dense-layer pair (2,042 x 1,204 -> 128 -> 128) under rocprofv3: fused (sgcn_dense2_fwd_f32), separate
calls (split-K + reduce), one unsplit GEMM.   rocprofv3 --kernel-trace --stats -d <dir> -- python profiles/dense2_probe.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from stochastic_gcn_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
src = torch.randn(232965, 1204, device=dev)
idx = torch.randperm(232965, device=dev)[:1021].to(torch.int32)
idx2 = torch.randperm(232965, device=dev)[:1021].to(torch.int32)
W1 = torch.randn(1204, 128, device=dev) * 0.03; W2 = torch.randn(128, 128, device=dev) * 0.1
o = torch.zeros(1, 128, device=dev); s = torch.ones(1, 128, device=dev)
d1 = ops.Drop(0.8, 1); d2 = ops.Drop(0.8, 2)
x = ops.GatheredRows(src, idx); x2 = ops.GatheredRows(src, idx2)
xd = src[idx.long()].contiguous(); x2d = src[idx2.long()].contiguous()
for rep in range(20):
    # A: fused pair, gathered + dropout (the step's configuration)
    ops.dense2_fwd(x, W1, o, s, True, W2, o, s, True, x2=x2, drop1=d1, drop2=d2)
    # B: fused pair, dense rows, no dropout
    ops.dense2_fwd(xd, W1, o, s, True, W2, o, s, True, x2=x2d)
    # C: separate calls (split-K + splitk_ln_act)
    y, _ = ops.dense_fwd(x, W1, o, s, True, x2=x2, drop=d1)
    ops.dense_fwd(y[:1021], W2, o, s, True, x2=y[1021:], drop=d2)
    # D: plain GEMM without split-K scratch (S = 1)
    big = torch.cat([xd, x2d])
    out = torch.empty(2042, 128, device=dev)
    from stochastic_gcn_amd._ffi import lib, check
    check(lib.sgcn_gemm_f32(0, 0, 2042, 128, 1204, big.data_ptr(), 1204, W1.data_ptr(), 128, out.data_ptr(), 128, 0, None, None, None, None))
torch.cuda.synchronize()
