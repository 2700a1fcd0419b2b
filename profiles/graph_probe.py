import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(1024, 256, device=dev); W = torch.randn(256, 128, device=dev) * 0.05
off = torch.zeros(1, 128, device=dev); sc = torch.ones(1, 128, device=dev)
out = torch.zeros(1024, 128, device=dev); dW = torch.zeros(256, 128, device=dev)
def step():
    xd, m = torch.native_dropout(x, 0.2, True)
    y, ctx = ops.dense_fwd(xd, W, off, sc, True)
    g = ops.ln_act_bwd(y, y, ctx, sc, True, torch.zeros_like(off), torch.zeros_like(sc))
    ops.gemm(xd, g, out=dW, trans_a=True, accumulate=False)
    big = torch.mm(x, W)
    out.copy_(y + big)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
a = []
for i in range(3):
    g.replay(); torch.cuda.synchronize(); a.append((float(out.sum()), float(dW.abs().sum())))
print("replays (dropout must differ each time):", a)
x.mul_(2.0); g.replay(); torch.cuda.synchronize(); print("after changing x:", float(out.sum()))
t0 = time.perf_counter()
for _ in range(1000): g.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("graph replay: %.1f us per step" % ((t1 - t0) * 1e3))
with torch.cuda.stream(s):
    t0 = time.perf_counter()
    for _ in range(1000): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("eager: %.1f us per step" % ((t1 - t0) * 1e3))

# ---- per-node cost of a replayed graph: 50 launches of a ~2 us kernel
theta = torch.zeros(4096, device=dev); gr = torch.ones(4096, device=dev); m1 = torch.zeros(4096, device=dev); v1 = torch.zeros(4096, device=dev)
def tiny():
    for _ in range(50):
        ops.adam_step(theta, gr, m1, v1, 0.01, 0.9, 0.999)
with torch.cuda.stream(s):
    tiny(); torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    tiny()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): g2.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("50 tiny kernels: graph replay %.1f us (%.2f us per node)" % ((t1 - t0) / 200 * 1e6, (t1 - t0) / 200 / 50 * 1e6))
with torch.cuda.stream(s):
    t0 = time.perf_counter()
    for _ in range(200): tiny()
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("50 tiny kernels: eager        %.1f us (%.2f us per launch)" % ((t1 - t0) / 200 * 1e6, (t1 - t0) / 200 / 50 * 1e6))
