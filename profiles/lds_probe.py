"""LDS-staged sweep on S-Reddit-SBM: plan statistics, forward SpMM time (planned part alone, residual alone, both)
against the two-lane-group column sweep on the same matrix, sampled rows against SciPy.  One JSON line per variant.
usage: python profiles/lds_probe.py [p_in ...]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stochastic_gcn_amd import ops, synthetic  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    d = 602
    for p_in in [float(x) for x in sys.argv[1:]] or [0.8]:
        n, _, a, _, _, _, labels, *_ = synthetic.reddit_sbm(p_in=p_in)
        comm = labels.argmax(1).astype(np.int32)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        B = torch.zeros((n, 608), device=dev)
        B[:, :d] = torch.randn((n, d), device=dev, generator=g)
        Bd = B[:, :d]
        out = torch.empty((n, 608), device=dev)[:, :d]
        bytes_alg = a.nnz * 8 + (n + 1) * 4 + 2 * n * d * 4
        rows = np.random.RandomState(0).choice(n, 64, replace=False)
        ref = a[rows].astype(np.float64).dot(Bd.cpu().numpy().astype(np.float64))

        def err(c):
            return float(np.abs(c[torch.from_numpy(rows).to(dev).long()].cpu().numpy() - ref).max() / np.abs(ref).max())

        cs = ops.ColumnSweepCSR(a, dev, G=2)
        t_tune, pace = cs.autotune(Bd)
        t_cs = timed(lambda: ops.spmm_cs(cs, Bd, out=out))
        print(json.dumps({"p_in": p_in, "variant": "column sweep G=2 (ungrouped, paced %d)" % pace, "ms": round(t_cs, 4),
                          "frac": round(bytes_alg / (t_cs * 1e-3) / 8e12, 4), "err": err(out)}), flush=True)
        for lab_name, lab in (("labels", comm), ("lp", None)):
            if lab is None:
                t0 = time.time()
                lab, ncomm = ops.reorder_labels(a)
                lp_s = time.time() - t0
            for mr, general in ((1, False), (2, False), (2, True), (3, False)):
                t0 = time.time()
                A = ops.LdsSweepCSR(a, dev, labels=lab, min_reuse=mr, general=general)
                build_s = time.time() - t0
                if A.residual is not None:
                    A.autotune(Bd)
                t_loc = timed(lambda: ops.spmm_lds(A, Bd, out=out, local_only=True))
                t_all = timed(lambda: ops.spmm_lds(A, Bd, out=out))
                ops.spmm_lds(A, Bd, out=out)
                e = err(out)
                rec = {"p_in": p_in, "variant": "lds sweep (%s, min_reuse %d)" % (lab_name, mr), "ms": round(t_all, 4),
                       "ms_planned_part": round(t_loc, 4), "frac": round(bytes_alg / (t_all * 1e-3) / 8e12, 4), "err": e,
                       "build_s": round(build_s, 2), **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in A.host_stats.items()}}
                if A.residual is not None:
                    rec["residual_pace"] = A.residual.pace.get(d)
                    rec["residual_nnz"] = A.residual.nnz
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
