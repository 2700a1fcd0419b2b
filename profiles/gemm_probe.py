"""The step's dense-layer GEMMs one by one (round 3): duration of each shape of the Reddit CVD+PP step as a dependent
chain of identical launches (each launch reads what the previous one wrote, so nothing overlaps), for a few
settings of the split-K rule (`gemm_min_steps`: K-steps a split-K slice keeps at least).

    python profiles/gemm_probe.py            -> JSON lines {shape, knob, us_per_call}
"""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import _ffi, ops      # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)


def rnd(*s):
    return torch.randn(*s, device=dev, generator=g)


PLUG = torch.randn(8192, 8192, device=dev)


def timed(fn, reps=100):
    """device-elapsed per call with the host out of the picture: the calls are queued behind a ~10 ms plug"""
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            torch.mm(PLUG, PLUG)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


SHAPES = [   # name, rows, rows of the second (stacked) operand, K, N, LayerNorm
    ("dense0 [dropout(x); x] 2 x 1018 x 1204 -> 128 (feature rows through an index)", 1018, 1018, 1204, 128, True),
    ("dense0-half x 1018 x 1204 -> 128 (one stream only: what a dual-accumulator form would load)", 1018, 0, 1204, 128, True),
    ("dense1 [h; mu] 2 x 1018 x 128 -> 128", 1018, 1018, 128, 128, True),
    ("dense2 512 x 256 -> 128", 512, 0, 256, 128, True),
    ("dense3 512 x 128 -> 41 (no LayerNorm)", 512, 0, 128, 41, False),
]

only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
feat = rnd(232965, 1204)
for name, n, n2, K, N, norm in SHAPES:
    if only and not name.startswith(only):
        continue
    W = rnd(K, N) * 0.05
    off, sc = (rnd(1, N) * 0.1, 1 + rnd(1, N) * 0.1) if norm else (None, None)
    if K == 1204:
        idx = torch.randint(0, 232965, (n,), device=dev, dtype=torch.int32)
        x = ops.GatheredRows(feat, idx)
        x2 = ops.GatheredRows(feat, idx) if n2 else None
    else:
        x = rnd(n, K)
        x2 = rnd(n2, K) if n2 else None
    for knob in ((3,) if only else (3, 5, 100)):
        _ffi.tune("gemm_min_steps", knob)
        us = timed(lambda: ops.dense_fwd(x, W, off, sc, norm, x2=x2))
        print(json.dumps({"shape": name, "gemm_min_steps": knob, "us_per_call": round(us, 2),
                          "split_ws_floats": int(_ffi.lib.sgcn_gemm_ws_floats(n + n2, N, K))}))
_ffi.tune("gemm_min_steps", 3)
