#!/usr/bin/env python
"""The CVD+PP minibatch epoch of bench.py's `train_epoch` leg on its own (for rocprofv3):

    rocprofv3 --kernel-trace --stats -d <dir> -- python profiles/epoch_profile.py [epochs]
    python profiles/epoch_profile.py --summarize <dir> <steps>  > profiles/rNN_train_epoch_kernels.txt
"""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(epochs):
    import torch
    import bench
    from stochastic_gcn_amd import synthetic
    data = synthetic.reddit_like(seed=1, with_features=False)
    te = bench.train_epoch_leg(data, torch.device("cuda:0"), epochs=epochs)
    print(te, file=sys.stderr)


def summarize(src, steps):
    f = glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
    t = sqlite3.connect(f[0])
    rows = list(t.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print("== rocprofv3 --kernel-trace --stats of profiles/epoch_profile.py: %d training steps "
          "(batch 512, CVD+PP, S-Reddit), all kernels incl. set-up" % steps)
    print("GPU-busy total %.1f ms = %.3f ms per step (upper bound: includes the one-off PP SpMM and set-up kernels)"
          % (tot / 1e3, tot / 1e3 / steps))
    print("%-92s %8s %10s %12s %10s %6s" % ("kernel", "calls", "calls/step", "total_ms", "avg_us", "pct"))
    for r in rows[:40]:
        print("%-92s %8d %10.2f %12.1f %10.2f %6.2f" % (r[0][:92], r[1], r[1] / steps, r[2] / 1e3, r[3], r[4]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2], int(sys.argv[3]))
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
