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
    from stochastic_gcn_amd import _ffi
    for kv in os.environ.get("SGCN_TUNE", "").split():          # library knobs for this run: SGCN_TUNE="key=value ..."
        k, v = kv.split("=")
        _ffi.tune(k, int(v))
    from stochastic_gcn_amd import synthetic
    from stochastic_gcn_amd.flags import FLAGS
    over = {}
    for kv in os.environ.get("SGCN_FLAGS", "").split():         # trainer flags for this run: SGCN_FLAGS="fuse_bwd=0 ..."
        k, v = kv.split("=")
        over[k] = type(getattr(FLAGS, k))(int(v)) if isinstance(getattr(FLAGS, k), (bool, int)) else float(v)
    if over:
        orig = FLAGS.update
        FLAGS.update = lambda **kw: (orig(**kw), orig(**over))[0]
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


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("--gaps", "--timeline", "--stepgaps")):
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2], int(sys.argv[3]))
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 3)


def gaps(src):
    """Idle time between consecutive kernels of the training steps (rocpd kernel dispatch timestamps):
    ~1.5 us back to back means the GPU is the bound, several us means it waits for the launching host."""
    f = glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
    t = sqlite3.connect(f[0])
    tabs = [r[0] for r in t.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tabs else None
    if view is None:
        print("tables:", tabs)
        return
    cols = [r[1] for r in t.execute("pragma table_info(%s)" % view)]
    rows = list(t.execute("select name, start, end from %s order by start" % view))
    import numpy as np
    st = np.array([r[1] for r in rows], dtype=np.int64)
    en = np.array([r[2] for r in rows], dtype=np.int64)
    names = [r[0] for r in rows]
    # steady state: the last third of the trace, only sgcn kernels' neighbours
    lo = len(rows) * 2 // 3
    gap = (st[lo + 1:] - en[lo:-1]) / 1e3
    dur = (en[lo:] - st[lo:]) / 1e3
    span = (en[-1] - st[lo]) / 1e3
    print("steady-state window: %d kernels over %.1f ms: busy %.1f ms (%.1f %%), idle between kernels %.1f ms"
          % (len(dur), span / 1e3, dur.sum() / 1e3, 100 * dur.sum() / span, gap.sum() / 1e3))
    print("gap between consecutive kernels: median %.2f us, mean %.2f us, p90 %.2f us, share of gaps > 4 us: %.1f %%"
          % (np.median(gap), gap.mean(), np.percentile(gap, 90), 100 * (gap > 4).mean()))
    adam = [i for i in range(lo, len(rows)) if "adam_" in names[i]]
    if len(adam) > 2:
        per = np.diff(st[adam]) / 1e3
        print("step period (adam to adam): median %.1f us; kernels per step %.1f"
              % (np.median(per), (adam[-1] - adam[0]) / (len(adam) - 1)))
    # the step's dependency chain, independent of how fast the (traced, slowed) host launches: per queue, kernels
    # and summed kernel time per step -- the compute queue's sum is the floor of the step on the GPU
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    if qcol and len(adam) > 2:
        qs = [r[0] for r in t.execute("select %s from %s order by start" % (qcol, view))]
        a, b = adam[0], adam[-1]
        nsteps = len(adam) - 1
        per_q = {}
        for i in range(a, b):
            k = per_q.setdefault(qs[i], [0, 0.0])
            k[0] += 1
            k[1] += (en[i] - st[i]) / 1e3
        for q, (cnt, us) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
            print("queue %s: %.1f kernels and %.1f us of kernel time per step" % (q, cnt / nsteps, us / nsteps))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--gaps":
    gaps(sys.argv[2])


def timeline(src, n=70):
    """One steady-state step as a kernel timeline (start offset, duration, queue): shows what overlaps."""
    f = glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
    t = sqlite3.connect(f[0])
    cols = [r[1] for r in t.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(t.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
    adam = [i for i, r in enumerate(rows) if "adam_" in r[0]]
    a, b = adam[len(adam) // 2], adam[len(adam) // 2 + 1]
    t0 = rows[a][2]
    print("columns:", cols)
    for r in rows[a:b + 1][:n]:
        print("%9.2f us  +%7.2f us  q=%s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", r[0][:70]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--timeline":
    timeline(sys.argv[2])


def stepgaps(src):
    """Where in the step the queue idles: over the steady-state training steps with the usual kernel count, the median idle
    time in FRONT of each kernel of the step and the kernel's median duration."""
    import numpy as np
    f = glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
    t = sqlite3.connect(f[0])
    rows = list(t.execute("select name, start, end from kernels order by start"))
    adam = [i for i, r in enumerate(rows) if "adam_" in r[0]]
    adam = adam[len(adam) // 3:]
    lens = np.diff(adam)
    usual = int(np.bincount(lens).argmax())
    steps = [a for a, n in zip(adam[:-1], lens) if n == usual]
    gap = np.array([[rows[a + j + 1][1] - rows[a + j][2] for j in range(usual)] for a in steps]) / 1e3
    dur = np.array([[rows[a + j + 1][2] - rows[a + j + 1][1] for j in range(usual)] for a in steps]) / 1e3
    print("%d steady-state steps of %d kernels: idle in front of each kernel / its duration (median us; p90 of the idle)"
          % (len(steps), usual))
    for j in range(usual):
        print("  idle %6.2f (p90 %6.2f)  run %6.2f  %s" % (np.median(gap[:, j]), np.percentile(gap[:, j], 90),
                                                            np.median(dur[:, j]), rows[steps[0] + j + 1][0][:60]))
    print("  per step: idle %.1f us + kernels %.1f us" % (np.median(gap.sum(1)), np.median(dur.sum(1))))
    # ... and over ALL steady-state steps (whatever their kernel count): the epoch pays the mean, not the median
    st = np.array([rows[a][1] for a in adam], dtype=np.int64)
    per = np.diff(st) / 1e3
    per = per[per < 2000]                       # (an epoch boundary is not a step)
    busy = np.array([sum(rows[i][2] - rows[i][1] for i in range(a + 1, b + 1)) for a, b in zip(adam[:-1], adam[1:])]) / 1e3
    busy = busy[:len(np.diff(st))][np.diff(st) / 1e3 < 2000]
    slow = [(int(i), round(float(p), 1)) for i, p in enumerate(np.diff(st) / 1e3) if 160 < p < 2000]
    print("  periods over 160 us (step index in the window, us):", slow[:40])
    print("  all %d steps: period mean %.1f / median %.1f / p90 %.1f / max %.1f us; kernel time mean %.1f / p90 %.1f us"
          % (len(per), per.mean(), np.median(per), np.percentile(per, 90), per.max(), busy.mean(), np.percentile(busy, 90)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--stepgaps":
    stepgaps(sys.argv[2])
