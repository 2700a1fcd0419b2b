import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import ops
dev = torch.device('cuda:0')
shapes = [  # (name, ta, tb, M, N, K)
    ("fwd0 stacked", 0, 0, 2042, 128, 1204), ("fwd1 stacked", 0, 0, 2042, 128, 128), ("fwd2", 0, 0, 512, 128, 256), ("fwd3", 0, 0, 512, 41, 128),
    ("dW0", 1, 0, 1204, 128, 1021), ("dW1", 1, 0, 128, 128, 1021), ("dW2", 1, 0, 256, 128, 512), ("dW3", 1, 0, 128, 41, 512),
    ("dx1", 0, 1, 1021, 128, 128), ("dx2", 0, 1, 512, 256, 128), ("dx3", 0, 1, 512, 128, 41)]
def timeit(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, ta, tb, M, N, K in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.zeros((M, N), device=dev)
    own = timeit(lambda: ops.gemm(A, B, out=out, trans_a=bool(ta), trans_b=bool(tb), accumulate=bool(ta)))
    a_, b_ = (A.t() if ta else A), (B.t() if tb else B)          # the library GEMM (rocBLAS via torch): comparison only
    lib = timeit(lambda: out.addmm_(a_, b_) if ta else torch.mm(a_, b_, out=out))
    print("%-14s M=%5d N=%4d K=%5d  own %7.1f us   rocBLAS(torch) %7.1f us" % (name, M, N, K, own, lib))

print("---- fused dense forward (stacked streams, dropout on the operand load)")
for name, n, K, N in [("fwd0", 1021, 1204, 128), ("fwd1", 1021, 128, 128), ("fwd2 (single)", 512, 256, 128)]:
    x = torch.randn((n, K), device=dev); mu = torch.randn((n, K), device=dev)
    W = torch.randn((K, N), device=dev) * 0.05
    off = torch.zeros((1, N), device=dev); sc = torch.ones((1, N), device=dev)
    drop = ops.Drop(0.8, 12345)
    def ev(f, reps=50):
        for _ in range(5): f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]
    a = ev(lambda: ops.dense_fwd(x, W, off, sc, True))
    b = ev(lambda: ops.dense_fwd(x, W, off, sc, True, x2=mu))
    c = ev(lambda: ops.dense_fwd(x, W, off, sc, True, x2=mu, drop=drop))
    d_ = ev(lambda: torch.mm(torch.cat((x, mu)), W))
    print("%-14s n=%5d K=%5d N=%4d  single %6.1f us | stacked %6.1f | stacked+dropout %6.1f | cat+rocBLAS mm %6.1f" % (name, n, K, N, a, b, c, d_))
