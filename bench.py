#!/usr/bin/env python
"""bench.py -- the headline measurement (BASELINE.json: "training edges/s (SpMM) + epoch time,
Reddit CVD+PP, 1/2/4/8 MI355X").

A *step* is one pass of the SpMM hot path over the Reddit-shape synthetic CSR (S-Reddit,
SURVEY.md §8d: N=232,965, nnz~23.17 M, 602-dim fp32 features, row pitch 608):
    forward   C  = A  . X        (K1/K11: tf.sparse_tensor_dense_matmul, gcn/layers.py:31-37)
    backward  dX = A^T . dC      (K6: its autodiff, run on the transposed CSR)
and, when N > 1, the RCCL all-reduce of the model's weight-gradient buffer (the path's one
exchange step).  `value` = edges (nonzeros) pushed through the SpMM kernels per
second, whole job, inputs resident in HBM.  Weak scaling: every rank owns one Reddit-shape
vertex-range shard.

One JSON line on stdout (rank 0).  Extra objects:
  roofline      dominant kernel (forward SpMM): algorithmic bytes per launch
                nnz*8 + (M+1)*4 + 2*N*d*4 / mean launch time from HIP events
                recorded on the launch stream inside the timed region, vs the 8 TB/s HBM peak.
  cpu_baseline  the oracle's OpenMP C restatement of the same product (oracle/oracle_c.c) on a
                bounded row sample of the same matrix, all host cores.
  train_epoch   (when available) the CVD+PP minibatch epoch on the same graph.
"""
import argparse
import json
import re
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md:35)
L2_PEAK = 34.5e12          # aggregate L2 bandwidth, MI355X_MICROARCH.md "L2 (per XCD)"
HBM_COPY = 6.29e12         # measured float4-copy ceiling (ibid.)


def _finish_args(args):
    if args.kernel is None:
        args.kernel = "lds" if args.workload == "reddit-sbm" else "cs"
    return args


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="reddit", choices=["reddit", "reddit-small", "reddit-114m", "reddit-sbm", "rmat", "rmat-10m"])
    p.add_argument("--reorder", default=None, choices=["none", "lp", "labels"],
                   help="locality-preserving column-sweep plan: communities from label propagation on the graph "
                        "(lp) or from the dataset's labels; default: lp for reddit-sbm, none otherwise")
    p.add_argument("--p-in", type=float, default=0.8, help="reddit-sbm: fraction of a vertex's edges inside its community")
    p.add_argument("--d", type=int, default=0, help="width of the dense operand (0: 602 for the Reddit shapes, 256 for S-RMAT -- BASELINE configs 3 / 5)")
    p.add_argument("--pitch", type=int, default=0, help="row pitch of X/C in floats (0: d rounded up to 32)")
    p.add_argument("--plan-t", type=int, default=0)
    p.add_argument("--tune", action="append", default=[], help="key=value passed to sgcn_tune")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-setup-report", action="store_true", help="skip the plan-cost / row-gather comparison (no row-gather launches in the run)")
    p.add_argument("--pmc", default="auto", choices=["auto", "live", "off"],
                   help="roofline.traffic: live = three separate rocprofv3 --pmc passes of this same command, run as child "
                        "processes (FETCH_SIZE / WRITE_SIZE / TCC_HIT+MISS: MI355X_MICROARCH.md's recipe), counted in THIS run; "
                        "auto = live for the default single-GPU column-sweep line, the committed profiles/*_traffic.json record "
                        "otherwise or when a pass fails; off = the committed record")
    p.add_argument("--no-epoch", action="store_true")
    p.add_argument("--no-strong", action="store_true", help="--gpus N > 1: skip the strong-scaling leg (ONE graph row-block sharded)")
    p.add_argument("--epoch-timeout", type=int, default=240, help="watchdog of the train-epoch leg, s")
    p.add_argument("--no-backward", action="store_true")
    p.add_argument("--kernel", default=None, choices=["rows", "cs", "lds"],
                   help="rows = row-gather SpMM (sgcn_spmm.hip); cs = column-sweep (sgcn_spmm_cs.hip); lds = LDS-staged "
                        "sweep for graphs with communities (sgcn_spmm_lds.hip) + the column sweep on its residual; "
                        "default: lds for reddit-sbm, cs otherwise")
    p.add_argument("--lds-residual-g", type=int, default=4, choices=[2, 4],
                   help="--kernel lds: lane groups per wave of the column sweep that multiplies the residual (4: every row resident in one round)")
    p.add_argument("--lds-min-reuse", type=int, default=0,
                   help="--kernel lds: a column is staged for a tile only if the tile references it this often (0: 3, or 1 -- everything through the ring -- when that leaves under 16 % of the nonzeros to the residual)")
    p.add_argument("--cs-t", type=int, default=0)
    p.add_argument("--colmod", type=int, default=0,
                   help="[experiment] fold column ids modulo this (makes B L2-resident: all-hit ceiling)")
    p.add_argument("--cs-align", type=int, default=-1, help="--cs-g 2 / 4: sweep positions one bin of a wave may run ahead of the slowest (-1: a third of the L2 window, ops.ColumnSweepCSR.auto_align)")
    p.add_argument("--shard-row-weight", type=int, default=-1,
                   help="--shard: nonzero-equivalents a row adds to its block's load (-1: parallel.ShardedSpMM.ROW_WEIGHT)")
    p.add_argument("--cs-ranges", default="auto", choices=["auto", "off"],
                   help="a sharded block's rows split by column range when the block is small enough "
                        "(ops.ColumnSweepCSR.choose_ranges); off: the 1-D plan")
    p.add_argument("--cs-warp", default="auto", choices=["auto", "on", "off"],
                   help="the sweep clock in work coordinates (sgcn_csplan_t.dev_warp): auto = when the nonzeros are not spread evenly over the column ids")
    p.add_argument("--cs-g", type=int, default=0, choices=[0, 1, 2, 4],
                   help="lane groups per wavefront of the column sweep (2: two 16-row bins on 128-column passes; "
                        "0: what ops.ColumnSweepCSR.choose_g picks for d -- also what the training path uses)")
    p.add_argument("--cpu-sample-rows", type=int, default=40000)
    p.add_argument("--grad-floats", type=int, default=0, help="size of the all-reduced gradient buffer")
    p.add_argument("--emulate-shard", default=None, metavar="R/W",
                   help="with --shard resident on ONE GPU: build and time rank R's block of a W-way sharding")
    p.add_argument("--shard", choices=["resident", "allgather"], default=None,
                   help="STRONG scaling: ONE graph row-block sharded over the ranks (load-balanced: nonzeros + a weight per row), dense "
                        "operand resident on every GPU or all-gathered per product (SURVEY.md 8e)")
    p.add_argument("--dry-run", action="store_true",
                   help="[test hook] no GPU work: only the launch / rendezvous / collective skeleton of "
                        "the run (what the CPU test of `--gpus N` exercises under SGCN_DIST_BACKEND=gloo); "
                        "the JSON line carries \"dry_run\": true and no throughput")
    return _finish_args(p.parse_args(argv))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-exec under
    torch.distributed.run, one rank per GPU (the driver's own invocation goes through
    torch.distributed.run already and never reaches this).  Fails loudly when the box has fewer
    than N GPUs -- a line with n_gpus != --gpus is never printed."""
    import subprocess
    backend = os.environ.get("SGCN_DIST_BACKEND", "nccl")
    if backend == "nccl" and not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible; RCCL needs one "
                             "device per rank (SGCN_DIST_BACKEND=gloo shares a device for smoke tests)\n"
                             % (args.gpus, have))
            return 2, None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    rc = proc.wait()
    return rc, (json.loads(line) if line else None)


def make_graph(args, rank):
    """Returns (n, full_adj, name, data10) -- data10 is the reference-style 10-tuple when the
    workload is a trainable dataset (used by the train_epoch leg), else None."""
    from stochastic_gcn_amd import synthetic
    if args.workload == "reddit":
        data = synthetic.reddit_like(seed=1 + rank, with_features=False)
        name = "S-Reddit full-graph CSR x dense (N=232965, Zipf(0.6) x uniform)"
    elif args.workload == "reddit-small":
        data = synthetic.reddit_like(n=23296, m=1160000, splits=(15241, 2369, 5533),
                                     seed=1 + rank, with_features=False)
        name = "S-Reddit/10 (N=23296)"
    elif args.workload == "reddit-sbm":     # S-Reddit degree law + 41 planted communities (locality-bearing)
        data = synthetic.reddit_sbm(seed=1 + rank, p_in=args.p_in)
        name = "S-Reddit-SBM full-graph CSR x dense (N=232965, Zipf(0.6) sources, 41 communities, p_in=%.2f)" % args.p_in
    elif args.workload == "reddit-114m":    # the denser Reddit distribution (114.6 M nnz, avg deg 492)
        data = synthetic.reddit_like(m=57_400_000, seed=1 + rank, with_features=False)
        return data[0], data[2], "S-Reddit-114M full-graph CSR x dense (N=232965, avg degree ~490)", None
    elif args.workload == "rmat-10m":       # BASELINE config 5 (SURVEY.md 8d S-RMAT); minutes of host time
        n = 10_000_000
        return n, synthetic.cached_graph("rmat_10m_200m_seed1", lambda: synthetic.rmat_like(n, 200_000_000, seed=1)), \
            "S-RMAT 10 M vertices, 200 M edges", None
    else:
        n = 1 << 20
        return n, synthetic.rmat_like(n, 20 * n, seed=1 + rank), "S-RMAT 2^20 vertices, 20 M edges", None
    return data[0], data[2], name, data


def train_epoch_leg(data, dev, epochs=6):
    """The CVD+PP minibatch epoch of BASELINE config 3 on the same synthetic graph: reddit.config
    flags (gcn/config/reddit.config:2) + --cv --cvd --degree=1 (README.md:46-55), 298 steps of
    batch 512, the real training path (sampler thread -> packed H2D -> fused step -> Adam ->
    history scatter), validation excluded (the reference reports time - ttime)."""
    import torch
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2,
                 layer_norm=True, hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512,
                 cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, seed=1,
                 native_prefetch=os.environ.get("SGCN_NATIVE_PREFETCH", "1") == "1",
                 plan_t=int(os.environ.get("SGCN_PLAN_T", FLAGS.plan_t)))
    n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    feats = torch.randn((n, 602), device=dev, generator=g)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # keep stdout to the single JSON line
        trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
    walls = []
    for _ in range(epochs):
        trn.train_epoch()
        walls.append(trn.last_epoch['train_wall_s'])
    le = trn.last_epoch
    best = min(walls[1:]) if len(walls) > 1 else walls[0]
    edges = le['sampled_edges'] + le['full_edges']
    chain = step_chain_probe(trn)
    # validation (gcn/train.py:133-160; the reference's epoch line reports it as ttime): one sweep over the 23,699
    # validation vertices through the compiled evaluation program, and the same sweep on the eager per-layer path
    val = {}
    try:
        with contextlib.redirect_stdout(sys.stderr):
            for name, native in (("program", True), ("eager", False), ("program", True), ("eager", False)):
                FLAGS.update(native_step=native)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                trn.evaluate(trn.val_d)
                torch.cuda.synchronize()
                val[name] = min(val.get(name, 1e9), time.perf_counter() - t0)
        FLAGS.update(native_step=True)
        val = {"val_sweep_s": val["program"], "val_sweep_eager_s": val["eager"],
               "batches": -(-len(trn.val_d) // FLAGS.test_batch_size),
               "epoch_plus_validation_s": best + val["program"], "epoch_plus_validation_eager_eval_s": best + val["eager"]}
    except Exception as e:
        val = {"error": repr(e)}
    # each aggregation edge is used by the forward aggregate; sampled edges again by the backward
    return {"epoch_time_s": best, "epoch_times_s": walls, "steps": le['steps'], "batch_size": 512,
            "ms_per_step": best / le['steps'] * 1e3, "sch_wait_s": le["sch_wait_s"], "host_loop_s": le.get("host_loop_s"),
            "producer_busy_s": le.get('producer_s'),
            "gpu_chain_us": chain.get("gpu_chain_us"), "host_launch_us": chain.get("host_launch_us"), "chain_probe": chain,
            "validation": val,
            "agg_edges_per_epoch": edges, "agg_edges_per_s": edges / best,
            "recipe": "reddit.config + --cv --cvd --degree=1 (CVD+PP), validation excluded"}


def step_chain_probe(trn, reps=20):
    """Host noise separated from the GPU's own step: the compiled step program of the LAST minibatch (its staging
    buffer is still on the device) replayed `reps` times behind a ~15 ms PLUG (two large library GEMMs on the same
    stream), so that the host has queued every launch of every replay before the GPU starts on the first -- no
    sampler, no H2D copy, no Python, and no launching host in the measured interval.  `gpu_chain_us` = device-elapsed
    time per replay between the plug's end and the last replay's end (HIP events on the step's stream): the step's
    chain of dependent kernels incl. the GPU's own dispatch gaps, i.e. what the GPU needs per step when it never waits
    for the host; `host_launch_us` = host time per sgcn_step_run call while queueing (launch calls only, nothing
    waits).  The epoch's ms_per_step is max(these two, the sampler's ms per batch) plus what the epoch loop adds."""
    import time
    import torch
    m = trn.train_model
    progs = [p for p in getattr(m, '_programs', {}).values() if p is not None]
    if not progs or (((m.grad_hook is not None) or (m.history_hook is not None)) and not getattr(progs[-1], 'native_world', 0)):
        return {"note": "no single-GPU step program to replay"}
    prog = progs[-1]
    dev = m.device
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.randn((8192, 8192), device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        prog.run('all', stream)
    torch.mm(a, a, out=b)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(a, a, out=b)
        torch.mm(b, a, out=b)
        e0.record()
        t0 = time.perf_counter()
        for _ in range(reps):
            prog.run('all', stream)
        host = time.perf_counter() - t0
        e1.record()
        e1.synchronize()
        dev_s = e0.elapsed_time(e1) * 1e-3
        if best is None or dev_s < best[0]:
            best = (dev_s, host)
    return {"gpu_chain_us": best[0] / reps * 1e6, "host_launch_us": best[1] / reps * 1e6, "replays": reps, "ops_per_step": prog.n_all,
            "what": "last minibatch's step program replayed behind a GEMM plug: device-elapsed per replay (GPU never waits "
                    "for the host) and host time per sgcn_step_run while queueing"}


def profiled_traffic(kernel_prefix, nnz, d):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, written by profiles/summarize.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command, gfx950 x2 FETCH correction applied).  None when no
    profile of this kernel/shape is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("kernel", "").startswith(kernel_prefix) and j.get("nnz") == nnz and j.get("d") == d:
            best = (j, os.path.basename(f))
    return best


def live_traffic(argv, pace_fwd, pace_bwd, kernel_prefix, launches, timeout_s=60):
    """HBM-side bytes per launch of the timed kernel, COUNTED IN THIS RUN: three separate ``rocprofv3 --pmc`` passes
    (FETCH_SIZE; WRITE_SIZE; TCC_HIT_sum + TCC_MISS_sum -- one counter group per run with --kernel-trace only, as
    /opt/skills/guides/MI355X_MICROARCH.md's HBM section prescribes) of this same bench command as child processes, with
    the clock the parent's autotune chose, no epoch / CPU / setup legs and a few steps.  FETCH_SIZE is KiB and reports
    half of a wide coalesced read on gfx950 (x 2); WRITE_SIZE is KiB.  None when rocprofv3 is missing or a pass fails
    (every pass is bounded by a timeout: the caller falls back to the committed record)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    base = tempfile.mkdtemp(prefix="sgcn_pmc_", dir=os.environ.get("TMPDIR") or "/tmp")
    keep, skip = [], False
    for a in argv:                       # (the parent's own --pmc choice does not travel)
        if skip:
            skip = False
        elif a == "--pmc":
            skip = True
        elif not a.startswith("--pmc="):
            keep.append(a)
    child = [sys.executable, os.path.abspath(__file__)] + keep + \
        ["--pmc", "off", "--no-epoch", "--no-cpu-baseline", "--no-setup-report", "--steps", "4", "--warmup", "1",
         "--tune", "cs_pace=%d" % int(pace_fwd)]
    env = dict(os.environ, TMPDIR="/tmp", SGCN_BENCH_CHILD="1")
    got, t0 = {}, time.time()
    try:
        for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
            out = os.path.join(base, name)
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "-d", out, "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None
            dbs = glob.glob(os.path.join(out, "**", "*results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            db = sqlite3.connect(dbs[0])
            for c in counters:
                row = db.execute("select name, avg(counter_value), avg(duration), count(*) from pmc_events where counter_name=? "
                                 "and name like ? group by name order by sum(duration) desc limit 1", (c, kernel_prefix + "%")).fetchone()
                if row is None:
                    return None
                got[c] = row
            db.close()
    except Exception:
        return None
    finally:
        shutil.rmtree(base, ignore_errors=True)
    f, w = got["FETCH_SIZE"][1] * 1024 * 2, got["WRITE_SIZE"][1] * 1024
    hit, mis = got["TCC_HIT_sum"][1], got["TCC_MISS_sum"][1]
    return {"kernel": got["FETCH_SIZE"][0], "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes_per_launch": f + w,
            "kernel_launches_per_spmm": launches, "hbm_bytes_per_spmm": (f + w) * launches,
            "l2_hit_rate": hit / max(hit + mis, 1.0), "ns_per_launch_profiled": got["FETCH_SIZE"][2],
            "dispatches_per_pass": int(got["FETCH_SIZE"][3]), "pace_ns": int(pace_fwd), "wall_s": round(time.time() - t0, 1)}


def gather_ceiling():
    """{'hit': TB/s, 'miss': TB/s} of the pure-gather microbenchmark (profiles/gather_ceiling.json), or None."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "gather_ceiling.json")))
        b = j["best_1216B"]
        return {"hit": b["hit"]["TBps"], "miss": b["miss"]["TBps"]}
    except Exception:
        return None


def setup_report(setup, full_adj, X, C, dev, cs_ms, ops):
    """What the headline kernel's plan costs before its first product, next to the kernel that needs no such plan.

    The reference runs this product once per matrix (gcn/utils.py:169-170, 321-322: train_adj . feats and full_adj . feats,
    cached in the dataset's .npz); a full-batch model runs it every layer of every step.  The column sweep pays a host
    plan (all cores), its upload and a clock autotune; the row-gather kernel (sgcn_spmm_csr_f32) pays the CSR's upload
    and a row-pointer pass.  `products_to_break_even_vs_rows_kernel` = the number of products from which the column
    sweep is the faster choice end to end (train.pp_products decides by the same arithmetic)."""
    def _round(x):
        return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in x.items()}
    rep = {k: (_round(v) if isinstance(v, dict) else round(v, 4)) for k, v in setup.items()}
    import torch
    torch.cuda.synchronize()
    t_s = time.perf_counter()
    R = ops.DeviceCSR.from_scipy(full_adj, dev, with_transpose=False)
    torch.cuda.synchronize()
    rows_setup = time.perf_counter() - t_s
    ops.spmm(R, X, out=C)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.spmm(R, X, out=C)
    e1.record()
    e1.synchronize()
    rows_ms = e0.elapsed_time(e1) / 3
    cs_setup = sum(setup["fwd"].get(k, 0.0) for k in ("plan_total_s", "autotune_s"))
    rep["fwd"]["setup_total_s"] = round(cs_setup, 4)
    if "bwd" in setup:
        rep["bwd"]["setup_total_s"] = round(setup.get("transpose_s", 0.0) + sum(setup["bwd"].get(k, 0.0) for k in ("plan_total_s", "autotune_s")), 4)
    rep["rows_kernel"] = {"setup_s": round(rows_setup, 4), "ms_per_product": round(rows_ms, 4),
                          "what": "sgcn_spmm_csr_f32 (spmm_seg_kernel): CSR upload + row plan, no autotune"}
    gain = (rows_ms - cs_ms) * 1e-3
    rep["products_to_break_even_vs_rows_kernel"] = (int(np.ceil(max(cs_setup - rows_setup, 0.0) / gain)) if gain > 0 else None)
    rep["first_product_s"] = {"column_sweep": round(cs_setup + cs_ms * 1e-3, 4), "rows_kernel": round(rows_setup + rows_ms * 1e-3, 4)}
    return rep


def reddit_grad_floats(d_in=602, hidden=128, classes=41):
    """Weight count of the Reddit CVD+PP stack:
    [2*602 -> 128] -> [128 -> 128] -> agg -> [256 -> 128] -> [128 -> 41] + LN offset/scale."""
    return 2 * d_in * hidden + hidden * hidden + 2 * hidden * hidden + hidden * classes + 6 * hidden


def usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(full_adj, d, rows, seed=0, budget_s=2.5):
    """The oracle's OpenMP C restatement of the same product on a bounded row sample, as an honest CPU
    number: one subprocess per thread count (oracle/cpu_baseline.py) with the threads PINNED and spread
    over the sockets, the dense operand page-interleaved over the NUMA nodes, at 1 / 8 / 16 / 32 / 64 /
    all usable cores -- `value` is the BEST of them and the whole scaling curve is printed (on the GPU
    box the container's CPU quota makes counts beyond it slower, not faster); plus scipy.sparse
    single-threaded, which is literally what the reference runs for this product
    (gcn/utils.py:321-322)."""
    import subprocess
    import tempfile
    n = full_adj.shape[0]
    rows = min(rows, n)
    sub = full_adj[:rows].tocsr()
    ncores = os.cpu_count() or 1
    usable = usable_cores()
    counts = sorted({c for c in (1, 8, 16, 32, 64, usable) if c <= ncores})
    script = os.path.join(ROOT, "oracle", "cpu_baseline.py")
    scaling, note = [], None
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "sample.npz")
        np.savez(path, indptr=sub.indptr, indices=sub.indices, data=sub.data, K=n)
        small = os.path.join(tmp, "small.npz")                   # one core gets a smaller sample (same matrix)
        one = full_adj[:min(4000, n)].tocsr()
        np.savez(small, indptr=one.indptr, indices=one.indices, data=one.data, K=n)
        for c in counts:
            env = dict(os.environ, OMP_NUM_THREADS=str(c), OMP_PROC_BIND="spread", OMP_PLACES="cores",
                       OMP_DYNAMIC="false")
            try:
                res = subprocess.run([sys.executable, script, small if c == 1 else path, str(d), str(budget_s), str(c)],
                                     env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
                scaling.append(json.loads(res.stdout.strip().splitlines()[-1]))
            except Exception as e:      # the checker is optional: keep the bench line
                note = "thread count %d failed: %r" % (c, e)
    best = max(scaling, key=lambda r: r["edges_per_s"]) if scaling else None
    out = {"value": best["edges_per_s"] if best else None, "unit": "edges/s", "cores": best["threads"] if best else 0,
           "kind": "port",
           "sample": "oracle_c.c OpenMP CSR SpMM (loop order of gcn/history.cpp:10-48), first %d rows (%d edges) of the "
                     "same matrix, d=%d, ~%.0f s per thread count, threads pinned (OMP_PROC_BIND=spread, OMP_PLACES=cores), "
                     "B page-interleaved over the NUMA nodes; one core: first %d rows" % (rows, sub.nnz, d, budget_s, one.shape[0]),
           "host_cores": ncores, "usable_cores": usable,
           "scaling": [{"threads": r["threads"], "edges_per_s": r["edges_per_s"], "reps": r["reps"]} for r in scaling]}
    if note:
        out["note"] = note
    for r in scaling:
        if r["threads"] == 1:
            out["single_thread_edges_per_s"] = r["edges_per_s"]
    # B2: scipy.sparse csr @ dense, single thread -- literally what the reference
    # uses for the PP product (gcn/utils.py:321-322)
    B = np.random.RandomState(seed).standard_normal((n, d)).astype(np.float32)
    small = full_adj[:min(8000, n)].tocsr()
    t0 = time.time()
    small.dot(B)
    out["scipy_single_thread_edges_per_s"] = small.nnz / (time.time() - t0)
    return out


def sampler_baseline(data10):
    """B3: host sampler ms per 512-vertex minibatch (cv, degree 1, L=1) -- this
    build's sampler (incl. CSR/plan packing) and, when oracle/_ref travelled, the reference C++."""
    from stochastic_gcn_amd.scheduler import PyScheduler
    n, train_adj, _, _, _, _, labels, tr, _, _ = data10
    ph = {'adj': ['a'], 'madj': ['m'], 'fadj': ['f'], 'fields': ['f0', 'f1'], 'ffields': ['ff'],
          'scales': ['s'], 'labels': 'l'}
    res = {}
    sch = PyScheduler(train_adj, labels, 1, [1], ph, 1, data=tr.copy(), cv=True)
    t0, k = time.time(), 0
    while k < 150 and sch.minibatch_packed(512) is not None:
        k += 1
    res["sgcn_packed_ms_per_batch"] = (time.time() - t0) / max(k, 1) * 1e3
    # the same batches from the prefetcher (core thread + 3 packer threads, bit-identical): producer throughput
    from stochastic_gcn_amd.scheduler import NativePrefetcher
    ids = np.random.RandomState(0).permutation(tr).astype(np.int32)
    batches = [ids[i:i + 512] for i in range(0, len(ids) - 511, 512)][:150]
    for _ in range(2):
        pre = NativePrefetcher(sch, batches, 0, depth=2, pin=False)
        t0, k = time.time(), 0
        while pre.next() is not None:
            k += 1
        res["sgcn_prefetch_ms_per_batch"] = (time.time() - t0) / max(k, 1) * 1e3
    res["sgcn_prefetch_threads"] = "1 core + %d packers" % pre.packers
    try:
        from oracle import ref_binding as rb
        if rb.available():
            ref = rb.RefPyScheduler(train_adj, labels, 1, [1], ph, 1, data=tr.copy(), cv=True)
            t0, k = time.time(), 0
            while k < 150 and ref.minibatch(512) is not None:
                k += 1
            res["reference_cpp_ms_per_batch"] = (time.time() - t0) / max(k, 1) * 1e3
    except Exception as e:      # the checker is optional here
        res["reference_cpp_ms_per_batch"] = None
    return res


def strong_leg(args, dev, world, rank, full_adj0, d, pitch, steps):
    """STRONG scaling beside the weak step: ONE S-Reddit graph (rank 0's) row-block sharded over
    the ranks by load (nonzeros + a weight per row: parallel.ShardedSpMM), forward A.X + backward A^T.dC per step, with the dense operand
    (a) resident on every GPU -- no collective on the data path -- and (b) sharded like the output and all-gathered
    before each product (7/8 of it crosses xGMI per GPU).  Barrier + synchronize on both sides, max over ranks."""
    import torch
    import torch.distributed as dist
    from stochastic_gcn_amd.parallel import DataParallel, ShardedSpMM
    par = DataParallel(device=dev, init=False)
    n = full_adj0.shape[0]
    sh = ShardedSpMM(par, full_adj0, dev, kernel="cs" if args.kernel == "lds" else args.kernel, d=d)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321)                              # the same operand on every rank
    Xp = torch.zeros((n, pitch), device=dev)
    Xp[:, :d] = torch.randn((n, d), device=dev, generator=gen)
    dCp = torch.zeros((n, pitch), device=dev)
    dCp[:, :d] = torch.randn((n, d), device=dev, generator=gen)
    X, dC = Xp[:, :d], dCp[:, :d]
    if args.kernel in ("cs", "lds"):
        sh.autotune(X, dC)
    C = torch.zeros((n, pitch), device=dev)[sh.lo:sh.hi, :d]
    dX = torch.zeros((n, pitch), device=dev)[sh.lo:sh.hi, :d]
    Xl, dCl = X[sh.lo:sh.hi].contiguous(), dC[sh.lo:sh.hi].contiguous()
    res = {}
    for name, gather in (("resident", False), ("allgather", True)):
        def one():
            sh.forward(sh.allgather_rows(Xl) if gather else X, out=C)
            sh.backward(sh.allgather_rows(dCl) if gather else dC, out=dX)
        for _ in range(2):
            one()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        res[name + "_ms"] = el / steps * 1e3
    nnz = int(full_adj0.nnz)
    res["edges_per_s"] = {"resident": 2 * nnz / (res["resident_ms"] * 1e-3), "allgather": 2 * nnz / (res["allgather_ms"] * 1e-3)}
    res["what"] = ("ONE S-Reddit graph (%d nnz), load-balanced row blocks (nonzeros + ShardedSpMM.ROW_WEIGHT per row) over %d rank(s), fwd A.X + bwd A^T.dC per step, d=%d; "
                   "resident: dense operand on every GPU, no data-path collective; allgather: operand sharded like the "
                   "output, all-gathered (RCCL) before each product" % (nnz, world, d))
    res["steps"] = steps
    res["rows_rank0"] = [int(sh.lo), int(sh.hi)]
    if sh.A is not None and hasattr(sh.A, "ntiles"):      # rank 0's plan: lane groups, column ranges (a block that does not fill one round of tiles), clock
        res["plan_rank0"] = {"G": int(getattr(sh.A, "G", 1)), "col_ranges": int(getattr(sh.A, "ranged", 0) or 0),
                             "tiles": int(sh.A.ntiles), "pace_ns": {"fwd": sh.A.pace.get(d), "bwd": sh.AT.pace.get(d) if sh.AT is not None else None}}
    return res


def dry_run(args, world, rank):
    """The distributed skeleton of a run without any GPU work (CPU test of `--gpus N`): rendezvous,
    an all-gather of the ranks, the timed loop's barrier / all-reduce pattern, one JSON line."""
    import torch
    import torch.distributed as dist
    ranks, total = [0], 1.0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SGCN_DIST_BACKEND", "gloo"))
        got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([rank], dtype=torch.int64))
        ranks = [int(x.item()) for x in got]
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
        total = float(t.item())
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    strong = None
    if world > 1:
        # the strong-scaling leg's partition + all-gather on CPU tensors (no kernels): every rank's row block of ONE
        # graph travels through ShardedSpMM.allgather_rows and must come back as the whole operand
        from stochastic_gcn_amd import synthetic
        from stochastic_gcn_amd.parallel import DataParallel, ShardedSpMM
        a = synthetic.rmat_like(1 << 10, 12 << 10, seed=1)
        par = DataParallel(device=torch.device("cpu"), init=False)
        sh = ShardedSpMM(par, a, torch.device("cpu"), kernel=None)
        full = torch.arange(a.shape[0] * 6, dtype=torch.float32).view(a.shape[0], 6)
        got = sh.allgather_rows(full[sh.lo:sh.hi].contiguous())
        strong = {"resident_ms": None, "allgather_ms": None, "edges_per_s": None, "dry_run": True,
                  "allgather_ok": bool(torch.equal(got, full)), "rows": [int(sh.lo), int(sh.hi)],
                  "local_nnz": int(sh.local_nnz), "nnz": int(a.nnz)}
    if world > 1:
        dist.destroy_process_group()
    out = {"metric": "training edges/s (SpMM)", "value": None, "unit": "edges/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / max(args.steps, 1) * 1e3,
           "dry_run": True, "ranks": ranks, "allreduce_of_rank_plus_1": total, "strong": strong}
    if rank == 0:
        print(json.dumps(out), flush=True)
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        rc, line = launch(args, argv)
        if rc != 0:
            raise SystemExit(rc)
        return line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to "
                         "print a line whose n_gpus differs from --gpus" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if os.environ.get("SGCN_DIST_BACKEND", "nccl") == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible; RCCL needs one device per rank"
                         % (world, torch.cuda.device_count()))
    # one process per GPU.  (SGCN_DIST_BACKEND=gloo lets a 1-GPU box smoke-test the N>1 code
    # path with several ranks sharing cuda:0; RCCL itself refuses duplicate devices.)
    backend = os.environ.get("SGCN_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # SGCN_FORCE_PG=1: a one-rank job takes the collective paths too (a real RCCL process group of one rank), so that
    # `bench.py --gpus 1` shows RCCL loading and prints grad_allreduce_ms
    pg = world > 1 or os.environ.get("SGCN_FORCE_PG", "0") not in ("", "0")
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if local_rank == 0:
        g.build(quiet=True)          # no-op when the in-tree .so files travelled with the snapshot
    if pg:
        dist.barrier()
    from stochastic_gcn_amd import ops, _ffi
    for kv in args.tune:
        k, v = kv.split("=")
        _ffi.tune(k, int(v))

    n, full_adj, wname, data10 = make_graph(args, 0 if args.shard else rank)
    if args.colmod:
        import scipy.sparse as sp
        coo = full_adj.tocoo()
        full_adj = sp.csr_matrix((coo.data, (coo.row, coo.col % args.colmod)), shape=full_adj.shape)
        full_adj.sort_indices()
        wname += " [columns folded mod %d]" % args.colmod
    d = args.d or (256 if args.workload.startswith("rmat") else 602)
    pitch = args.pitch or (d + 31) // 32 * 32
    nnz = int(full_adj.nnz)
    sh = None
    setup = None
    reorder = args.reorder or ("lp" if args.workload == "reddit-sbm" else "none")
    reorder_info = None
    if args.shard:
        from stochastic_gcn_amd.parallel import DataParallel, ShardedSpMM
        par = DataParallel(device=dev, init=False)
        if args.emulate_shard:
            assert world == 1 and args.shard == "resident", "--emulate-shard: one process, resident operand"
            import types
            r_, w_ = (int(x) for x in args.emulate_shard.split("/"))
            par = types.SimpleNamespace(rank=r_, world=w_, active=False)    # no peers: no collectives
        sh = ShardedSpMM(par, full_adj, dev, kernel="cs" if args.kernel == "lds" else args.kernel, with_transpose=not args.no_backward,
                         d=d if args.cs_g == 0 else (None if args.cs_g == 1 else d), G=args.cs_g if args.cs_g in (2, 4) else None,
                         plan_kw=dict(align=('auto' if args.cs_align < 0 else args.cs_align), T=args.cs_t,
                                      warp={'auto': 'auto', 'on': True, 'off': False}[args.cs_warp],
                                      col_ranges='auto' if args.cs_ranges == 'auto' else 0),
                         row_weight=None if args.shard_row_weight < 0 else args.shard_row_weight)
        A = sh.A
    elif args.kernel in ("cs", "lds"):
        comm = None
        if args.kernel == "lds" and reorder == "none":
            raise SystemExit("bench.py: --kernel lds needs communities (--reorder lp | labels)")
        if reorder == "lp":
            t_lp = time.time()
            comm, ncomm = ops.reorder_labels(full_adj)
            reorder_info = {"method": "label propagation on the graph (sgcn_reorder_lp)", "communities": ncomm,
                            "host_s": round(time.time() - t_lp, 2)}
        elif reorder == "labels":
            comm = np.ascontiguousarray(data10[6].argmax(1), dtype=np.int32)
            reorder_info = {"method": "dataset labels", "communities": int(comm.max()) + 1}
        if args.kernel == "lds":
            # planned nonzeros through the LDS ring, the rest (columns a tile references < min_reuse times) through the
            # two-lane-group column sweep into the same output
            def lds_plan(m):
                if args.lds_min_reuse > 0:
                    return ops.LdsSweepCSR(m, dev, labels=comm, min_reuse=args.lds_min_reuse, residual_G=args.lds_residual_g)
                # (0: what train.pp_products gets from LdsSweepCSR.for_graph -- min_reuse 3, or everything through the ring
                # when that leaves a residual under 16 % of the nonzeros)
                return ops.LdsSweepCSR(m, dev, host=ops.LdsSweepCSR.auto_host(m, comm), residual_G=args.lds_residual_g)
            A = lds_plan(full_adj)
            A.transpose = None if args.no_backward else lds_plan(full_adj.T.tocsr())
            mm = ops.spmm_lds
            reorder_info["lds_plan"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in A.host_stats.items()}
        else:
            cs_g = args.cs_g or (ops.ColumnSweepCSR.choose_g(d, nnz / max(full_adj.shape[0], 1), full_adj.shape[0]) if comm is None else 1)
            gk = dict(G=cs_g, align=('auto' if args.cs_align < 0 else args.cs_align), warp={'auto': 'auto', 'on': True, 'off': False}[args.cs_warp]) if (cs_g != 1 and comm is None) else dict(col_labels=comm, row_labels=comm)
            # the plans' setup is timed: the reference runs this product ONCE per matrix (gcn/utils.py:321-322), so what a
            # plan costs to build is user-visible time there; reported as "setup" beside the line
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            A = ops.ColumnSweepCSR(full_adj, dev, T=args.cs_t, **gk)
            torch.cuda.synchronize()
            setup = {"fwd": dict(A.setup_s, plan_total_s=time.perf_counter() - t_s)}
            A.transpose = None
            if not args.no_backward:
                t_s = time.perf_counter()
                full_adj_t = ops.transpose_host(full_adj)          # (until round 5: SciPy's single-threaded csr -> csc pass)
                setup["transpose_s"] = time.perf_counter() - t_s
                t_s = time.perf_counter()
                A.transpose = ops.ColumnSweepCSR(full_adj_t, dev, T=args.cs_t, **gk)
                torch.cuda.synchronize()
                setup["bwd"] = dict(A.transpose.setup_s, plan_total_s=time.perf_counter() - t_s)
                del full_adj_t
            mm = ops.spmm_cs
    else:
        A = ops.DeviceCSR.from_scipy(full_adj, dev, plan_T=args.plan_t, with_transpose=not args.no_backward)
        mm = ops.spmm
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    Xp = torch.zeros((n, pitch), device=dev)
    Xp[:, :d] = torch.randn((n, d), device=dev, generator=gen)
    dCp = torch.zeros((n, pitch), device=dev)
    dCp[:, :d] = torch.randn((n, d), device=dev, generator=gen)
    X, dC = Xp[:, :d], dCp[:, :d]
    C = torch.zeros((n, pitch), device=dev)[:, :d]
    dX = torch.zeros((n, pitch), device=dev)[:, :d]
    tuned = None
    if sh is not None:
        gen.manual_seed(1234)                      # one B / dC for the whole job
        Xp[:, :d] = torch.randn((n, d), device=dev, generator=gen)
        dCp[:, :d] = torch.randn((n, d), device=dev, generator=gen)
        if args.kernel in ("cs", "lds") and not any(kv.startswith("cs_pace") for kv in args.tune):
            sh.autotune(X, None if args.no_backward else dC)
            tuned = {"fwd_pace": sh.A.pace.get(d) if sh.A is not None else None,
                     "bwd_pace": sh.AT.pace.get(d) if sh.AT is not None else None}
        C, dX = C[sh.lo:sh.hi], dX[sh.lo:sh.hi]
        Xl, dCl = X[sh.lo:sh.hi].contiguous(), dC[sh.lo:sh.hi].contiguous()
    elif args.kernel in ("cs", "lds") and not any(kv.startswith("cs_pace") for kv in args.tune):
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        tuned = {"fwd": A.autotune(X)}                  # untimed setup, once per plan and width (lds: its residual's clock)
        torch.cuda.synchronize()
        if setup is not None:
            setup["fwd"]["autotune_s"] = time.perf_counter() - t_s
        if not args.no_backward:
            t_s = time.perf_counter()
            tuned["bwd"] = A.transpose.autotune(dC)
            torch.cuda.synchronize()
            if setup is not None:
                setup["bwd"]["autotune_s"] = time.perf_counter() - t_s
    gfl = args.grad_floats or reddit_grad_floats()
    grad = torch.randn(gfl, device=dev, generator=gen)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    ev_ar = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(args.steps)] if (pg and not args.shard) else []

    def step_sharded(i=None):
        gather = args.shard == "allgather"
        Bf = sh.allgather_rows(Xl) if gather else X
        if i is not None:
            ev[i][0].record()
        sh.forward(Bf, out=C)
        if i is not None:
            ev[i][1].record()
        if not args.no_backward:
            sh.backward(sh.allgather_rows(dCl) if gather else dC, out=dX)

    def step(i=None):
        if sh is not None:
            return step_sharded(i)
        if i is not None:
            ev[i][0].record()
        mm(A, X, out=C)
        if i is not None:
            ev[i][1].record()
        if not args.no_backward:
            mm(A.transpose, dC, out=dX)
        if pg:
            if i is not None:
                ev_ar[i][0].record()
            dist.all_reduce(grad)
            if i is not None:
                ev_ar[i][1].record()

    for _ in range(args.warmup):
        step()
    if pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if pg:
        dist.barrier()
    el = time.perf_counter() - t0
    if pg:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    ar_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_ar])) if ev_ar else None
    edges_per_step = nnz * (1 if args.no_backward else 2)
    if args.emulate_shard:
        edges_per_step = sh.local_nnz * (1 if args.no_backward else 2)       # this block only
    value = edges_per_step * (1 if sh is not None else world) * args.steps / el
    bytes_alg = nnz * 8 + (n + 1) * 4 + 2 * n * d * 4
    if sh is not None:      # this rank's row block: its nonzeros, its C rows, every referenced B row
        nnz_l, m_l = sh.local_nnz, sh.hi - sh.lo
        bytes_alg = nnz_l * 8 + (m_l + 1) * 4 + sh.distinct_cols * d * 4 + m_l * d * 4
    achieved = bytes_alg / (fwd_ms * 1e-3)
    lds_parts = None
    if args.kernel == "lds" and sh is None:     # the forward product's two kernels on their own (untimed region)
        def _t(fn, reps=5):
            fn()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(reps):
                fn()
            b_.record()
            b_.synchronize()
            return a_.elapsed_time(b_) / reps
        lds_parts = {"planned_part": round(_t(lambda: ops.spmm_lds(A, X, out=C, local_only=True)), 4),
                     "residual_part": round(_t(lambda: ops.spmm_cs(A.residual, X, out=C, beta=1.0)), 4)
                     if A.residual is not None else 0.0}
    out = {
        "metric": "training edges/s (SpMM)", "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if sh is not None else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wname + ", fwd A.X%s, d=%d (pitch %d)%s" % (
            "" if args.no_backward else " + bwd A^T.dC", d, pitch,
            (", + all-gather of the dense operand per product" if args.shard == "allgather" else "") if sh is not None
            else (", + RCCL all-reduce of %d grad floats" % gfl if pg else "")),
            "N": n, "nnz": nnz, "d": d,
            "per_gpu": ("load-balanced row block (nonzeros + %d per row) of ONE graph, dense operand %s; rank 0 rows [%d, %d), %d nnz"
                        % (sh.row_weight, args.shard, sh.lo, sh.hi, sh.local_nnz)) if sh is not None
            else "one S-Reddit vertex-range shard",
            "kernel": args.kernel, "tune": args.tune, "grad_allreduce_ms": ar_ms,
            # column sweep: passes over the feature dimension x rounds of resident tiles, as the library reports it
            "kernel_launches_per_spmm": int(re.search(r" x (\d+) launches", A.variant(d)).group(1))
            if args.kernel == "cs" else 1,
            "lds_parts_ms": lds_parts,
            "cs_plan": ({"G": int(getattr(A, "G", 1)), "col_ranges": int(getattr(A, "ranged", 0) or 0), "tiles": int(A.ntiles),
                         "fix_rows": int(A.nfix), "align": getattr(A, "align", None),
                         "pad_fraction": round(float(getattr(A, "pad_fraction", 0.0)), 4), "T": getattr(A, "T", None),
                         "warp_table": None if getattr(A, "warp", None) is None else [int(A.warp.numel()), int(A.warp_shift)]}
                        if args.kernel == "cs" else None),
            "cs_autotune_ms_pace": tuned},
        "roofline": {"bound": "hbm", "kernel": ((A.variant(d) + " + cs_fix_kernel: one SpMM") if args.kernel == "cs"
                                else A.variant(d) if args.kernel == "lds"
                                else "sgcn::spmm_seg_kernel (forward A.X, incl. split-row fix-up)"),
                     "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "frac_of_copy_ceiling": achieved / HBM_COPY,
                     # "bound" is the contract's roof (north_star: HBM).  The roof the counters say this kernel actually
                     # sits on (L2 busy 95 %, VALU 8 % for the gathers; the LDS sweep's chunk
                     # statement is VALU-bound) is carried next to it: frac_of_l2 = per-edge L2 -> VGPR bytes / time
                     # over the guide's aggregate L2 rate (= l2_gather.frac below)
                     "bound_measured": "valu" if args.kernel == "lds" else "l2",
                     "frac_of_l2": (sh.local_nnz if sh is not None else nnz) * d * 4 / (fwd_ms * 1e-3) / L2_PEAK,
                     "traffic": None, "traffic_source": None, "bytes_alg_per_launch": bytes_alg,
                     "ms_per_launch": fwd_ms,
                     "edges_per_s_fwd": (sh.local_nnz if sh is not None else nnz) / (fwd_ms * 1e-3),
                     "gather_model_GBps": (nnz * (d * 4 + 8) + n * d * 4) / (fwd_ms * 1e-3) / 1e9
                     if sh is None else None,
                     # secondary view: every edge moves one d-float row of B from L2 to the VGPRs
                     # whatever the cache hit rate, so the L2 (MI355X_MICROARCH.md: ~34.5 TB/s
                     # aggregate) is the bound a gather-based SpMM on this graph actually meets
                     "l2_gather": {"bytes_per_launch": (sh.local_nnz if sh is not None else nnz) * d * 4,
                                   "achieved_GBps": (sh.local_nnz if sh is not None else nnz) * d * 4 / (fwd_ms * 1e-3) / 1e9,
                                   "peak_GBps": L2_PEAK / 1e9,
                                   "frac": (sh.local_nnz if sh is not None else nnz) * d * 4 / (fwd_ms * 1e-3) / L2_PEAK}},
    }
    if reorder_info:
        out["config"]["reorder"] = reorder_info
    if setup is not None and not args.no_setup_report:
        out["setup"] = setup_report(setup, full_adj, X, C, dev, fwd_ms, ops)
    gc = gather_ceiling()
    if gc is not None and args.kernel == "cs":
        # every edge moves one d-float row of B from an L2 into VGPRs whatever else happens; the pure
        # gather microbenchmark (profiles/gather_ceiling.hip) gives the rate of that alone
        nn = sh.local_nnz if sh is not None else nnz
        t_floor = nn * d * 4 / (gc["hit"] * 1e12)
        out["roofline"]["gather_ceiling"] = {
            "source": "profiles/gather_ceiling.json (pure row gather, no FMA; 1,216 B pieces)",
            "l2_hit_TBps": gc["hit"], "fabric_miss_TBps": gc["miss"],
            "ms_all_hit_floor": t_floor * 1e3,
            "note": "hits and misses do not overlap in the vector memory path (mixed launch = sum of the two): "
                    "T_model = miss_bytes/miss_rate + hit_bytes/hit_rate"}
        out["roofline"]["frac_of_gather_ceiling"] = t_floor / (fwd_ms * 1e-3)
    # the committed PMC record of exactly the kernel variant that was dispatched
    if args.kernel == "lds":         # (the whole two-kernel product: profiles/r32_lds_traffic.json is assembled from its PMC table)
        tr = profiled_traffic("void sgcn::lds_spmm_kernel", nnz, d) if not (args.tune or sh is not None) else None
    else:
        # (a block of a sharded graph: the committed passes of exactly that block -- its nonzero count identifies it)
        tr = profiled_traffic(("void " + A.variant(d).split(" x ")[0].replace("false>", "").replace("true>", "").rstrip(", "))
                              if args.kernel == "cs" else "void sgcn::spmm", sh.local_nnz if sh is not None else nnz, d) \
            if not (args.tune or reorder != "none") else None
    # ... or, better, counted in THIS run (a committed file is the builder's number): the default single-GPU
    # column-sweep line spends ~30 s on three child passes under rocprofv3 --pmc
    live = None
    want_live = args.pmc == "live" or (args.pmc == "auto" and sh is None and not args.tune and reorder == "none")
    if want_live and args.kernel == "cs" and rank == 0 and world == 1 and not os.environ.get("SGCN_BENCH_CHILD") and tuned:
        pf = tuned["fwd"][1] if "fwd" in tuned else tuned.get("fwd_pace")
        pb_ = (tuned["bwd"][1] if "bwd" in tuned else tuned.get("bwd_pace")) if not args.no_backward else pf
        if pf and pf > 0:
            live = live_traffic(argv, pf, pb_, "void sgcn::cs_spmm16", out["config"]["kernel_launches_per_spmm"])
    if live is not None:
        committed = tr
        tr = (dict(live, kernel=live["kernel"]), None)
        out["roofline"]["traffic_live"] = {k: live[k] for k in ("fetch_bytes_corrected", "write_bytes", "l2_hit_rate", "ns_per_launch_profiled",
                                                                 "dispatches_per_pass", "pace_ns", "wall_s")}
        if committed is not None:
            out["roofline"]["traffic_committed"] = {"hbm_bytes_per_spmm": committed[0]["hbm_bytes_per_spmm"], "source": "profiles/" + committed[1]}
    if tr is not None:
        out["roofline"]["traffic"] = tr[0]["hbm_bytes_per_spmm"]
        # what the memory side actually moves (profiled bytes / measured time), next to the compulsory model
        out["roofline"]["traffic_GBps"] = tr[0]["hbm_bytes_per_spmm"] / (fwd_ms * 1e-3) / 1e9
        out["roofline"]["traffic_frac_of_peak"] = tr[0]["hbm_bytes_per_spmm"] / (fwd_ms * 1e-3) / HBM_PEAK
        out["roofline"]["traffic_source"] = ("profiles/%s (separate rocprofv3 --pmc passes; kernel %s, L2 hit %.3f)" % (
            tr[1], tr[0]["kernel"], tr[0].get("l2_hit_rate", float("nan")))) if tr[1] is not None else (
            "live: three separate rocprofv3 --pmc passes (FETCH_SIZE x 2 [gfx950 wide-read correction]; WRITE_SIZE; TCC_HIT / TCC_MISS) of "
            "this command as child processes inside this run, clock fixed at the autotuned %d ns; kernel %s, L2 hit %.3f, "
            "mean over the forward and backward launches" % (tr[0]["pace_ns"], tr[0]["kernel"], tr[0]["l2_hit_rate"]))
        if gc is not None:      # the two-rate model of this kernel's own traffic
            miss_b = tr[0]["fetch_bytes_corrected"] * tr[0]["kernel_launches_per_spmm"]
            hit_b = max((sh.local_nnz if sh is not None else nnz) * (d * 4 + 8) - miss_b, 0)
            t_model = miss_b / (gc["miss"] * 1e12) + hit_b / (gc["hit"] * 1e12)
            out["roofline"].setdefault("gather_ceiling", {})["ms_model_for_profiled_traffic"] = t_model * 1e3
            out["roofline"]["frac_of_traffic_model"] = t_model / (fwd_ms * 1e-3)
    def emit():
        if rank == 0:
            print(json.dumps(out), flush=True)

    if world > 1 and sh is None and not args.no_strong:
        # the weak line above is ~N x by construction (every rank its own graph + a 0.84 MB all-reduce); the
        # informative numbers of a multi-GPU run are these.  A watchdog bounds a wedged collective (RCCL has not run on
        # hardware yet): the weak line is still printed.
        import threading

        def bail_strong():
            out["strong"] = {"error": "timed out after %d s" % args.epoch_timeout}
            emit()
            os._exit(0)
        sdog = threading.Timer(args.epoch_timeout, bail_strong)
        sdog.daemon = True
        sdog.start()
        try:
            adj0 = full_adj if rank == 0 else make_graph(args, 0)[1]
            del A
            torch.cuda.empty_cache()
            out["strong"] = strong_leg(args, dev, world, rank, adj0, d, pitch, max(3, min(args.steps, 10)))
            out["strong"]["grad_allreduce_ms"] = ar_ms
        except Exception as e:
            out["strong"] = {"error": repr(e)}
        sdog.cancel()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(full_adj, d, args.cpu_sample_rows)
        if data10 is not None:
            out["cpu_baseline"]["sampler"] = sampler_baseline(data10)

    if not args.no_epoch and data10 is not None and sh is None:
        # the minibatch training epoch, on every rank (vertex-range shards, RCCL gradient
        # all-reduce + history exchange when N > 1).  A watchdog bounds a wedged collective: the
        # SpMM line above is still printed.
        import threading

        def bail():
            out["train_epoch"] = {"error": "timed out after %d s" % args.epoch_timeout}
            emit()
            os._exit(0)
        dog = threading.Timer(args.epoch_timeout, bail)
        dog.daemon = True
        dog.start()
        A = None
        del A, Xp, dCp, X, dC, C, dX
        torch.cuda.empty_cache()
        try:
            if rank != 0:       # data parallelism needs ONE graph, replicated: rank 0's (seed 1)
                data10 = make_graph(args, 0)[3]
            te = train_epoch_leg(data10, dev)
            if world > 1:
                t = torch.tensor([te["epoch_time_s"]], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                te["epoch_time_s"] = float(t.item())
                te["ms_per_step"] = te["epoch_time_s"] / te["steps"] * 1e3
                te["agg_edges_per_s"] = None
                te["sharding"] = "train ids by vertex range over %d ranks; global batch %d" % (world, 512 * world)
            out["train_epoch"] = te
        except Exception as e:      # the headline SpMM measurement must survive this leg
            out["train_epoch"] = {"error": repr(e)}
        dog.cancel()
    emit()
    if pg:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
