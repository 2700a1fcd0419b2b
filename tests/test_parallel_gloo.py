"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel path in
stochastic_gcn_amd/parallel.py: vertex-range sharding, the single flat gradient all-reduce
(mean), replica-consistent history synchronisation, and a 2-rank training step whose compute
leg is the NumPy oracle (the HIP kernels need a GPU; the distributed logic does not)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from stochastic_gcn_amd.parallel import DataParallel
    par = DataParallel(backend="gloo", device=torch.device("cpu"))
    try:
        res = globals()[fn](par)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), **res)
    finally:
        par.shutdown()


def _run(fn, tmp_path, world=2):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, fn, str(tmp_path)), nprocs=world, join=True)
    return [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]


# ---- workers ---------------------------------------------------------------------------------
def w_shard_and_allreduce(par):
    n = 1001
    ids = np.random.RandomState(0).permutation(n)[:700]
    mine = par.shard_ids(ids, n)
    flat = torch.arange(10, dtype=torch.float32) * (par.rank + 1)
    par.allreduce_mean_(flat)
    theta = torch.full((5,), float(par.rank))
    par.broadcast_(theta)
    return dict(mine=mine, flat=flat.numpy(), theta=theta.numpy(), lo_hi=np.array(par.vertex_range(n)),
                mx=np.array([par.max_scalar(3 + par.rank)]))


def w_history(par):
    from oracle import oracle_np as onp
    N, d = 50, 6
    H = torch.zeros((N, d))
    rng = np.random.RandomState(10 + par.rank)
    n = 7 + 3 * par.rank                                   # ragged sizes across ranks
    idx = rng.choice(N, n, replace=False).astype(np.int32)
    idx[0] = 5                                             # a vertex both ranks update
    rows = rng.standard_normal((n, d)).astype(np.float32)

    def scatter(h, i, r):
        onp.scatter_rows(h.numpy(), i.numpy(), r.numpy())
    par.sync_history(H, torch.from_numpy(idx), torch.from_numpy(rows), scatter)    # size-exchange path
    assert not H.any()                                     # asynchronous: nothing lands before the join
    par.join_history()
    H2 = torch.zeros((N, d))
    par.history_cap = 12                                   # fixed-capacity path (-1 padded ids)
    par.sync_history(H2, torch.from_numpy(idx), torch.from_numpy(rows), scatter)
    par.sync_history(H2, torch.from_numpy(idx), torch.from_numpy(rows), scatter)   # two exchanges in flight: own buffers each
    par.join_history()
    par.sync_history(H2, torch.from_numpy(idx), torch.from_numpy(rows), scatter)   # reuses the first pair
    par.join_history()
    par.join_history()                                     # nothing pending: a no-op
    par.history_cap = None
    return dict(H=H.numpy(), H2=H2.numpy(), idx=idx, rows=rows)


def w_train_step(par):
    import model_cases as mc
    from oracle import oracle_np as onp
    from stochastic_gcn_amd.scheduler import PyScheduler
    case = mc.build_case('reddit_cvd_pp')
    fl, c, ph = case['flags'], case['cfg'], case['ph']
    om = mc.make_oracle_model(case, seed=3)
    train = par.shard_ids(case['train'], c['n']).astype(np.int32)
    sch = PyScheduler(case['adj'], case['labels'], 1, [1], ph, par.sampler_seed(1), data=train, cv=True)
    names = sorted(om.params)
    out = {}
    for step in range(2):
        feed = sch.minibatch(16)
        logits, _ = om.forward(feed, ph, 0.0, lambda *a: None)
        loss, acc, pred, dlogits = om.loss_and_grad(logits, feed[ph['labels']])
        grads = om.backward(dlogits)
        flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in names]))
        out["local_grad%d" % step] = flat.numpy().copy()
        par.allreduce_mean_(flat)                          # the ONE collective of the step
        off = 0
        for k in names:
            sz = grads[k].size
            grads[k] = flat.numpy()[off:off + sz].reshape(grads[k].shape).copy()
            off += sz
        om.adam_step(grads)
        hist = torch.from_numpy(om.history[0])
        par.sync_history(hist, torch.from_numpy(feed[ph['fields'][0]]), torch.from_numpy(om._new_hist[0]),
                         lambda h, i, r: onp.scatter_rows(h.numpy(), i.numpy(), r.numpy()))
        par.join_history()                                 # (the product joins in front of the next step's aggregator)
        out["avg_grad%d" % step] = flat.numpy().copy()
    out["theta"] = np.concatenate([om.params[k].ravel() for k in names])
    out["hist"] = om.history[0]
    out["n_train"] = np.array([len(train)])
    return out


def w_stop_verdict(par):
    """Ranks whose validation histories disagree (rank 1's plateaued, rank 0's did not) must still
    leave the epoch loop together (a rank-local `break` deadlocks the peers)."""
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import stop_verdict
    FLAGS.reset()
    FLAGS.update(early_stopping=2, epochs=100, data=0)
    falling = [1.0, 0.9, 0.8, 0.7, 0.6]
    rising = [1.0, 0.9, 0.8, 0.9, 1.0]
    mine = rising if par.rank == 1 else falling
    v1 = stop_verdict(par, 4, mine, 10)                     # rank 0 says go on -> everybody goes on
    mine = rising if par.rank == 0 else falling
    v2 = stop_verdict(par, 4, mine, 10)                     # rank 0 says stop -> everybody stops
    FLAGS.update(epochs=3, data=25)
    v3 = stop_verdict(par, 4, falling, 10)                  # job-wide data 20 < 25: go on
    v4 = stop_verdict(par, 4, falling, 13)                  # job-wide data 26 >= 25 and epoch > epochs
    return dict(v=np.array([v1, v2, v3, v4]))


def w_eight(par):
    """World size 8 (the node the driver's scaling run uses): sharding, the fixed-capacity history exchange with cap x 8
    receive blocks, duplicates that several ranks write in one step, the collective stop decision, the sampler's packer
    count under a shared host."""
    from oracle import oracle_np as onp
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.scheduler import default_packers
    from stochastic_gcn_amd.train import stop_verdict
    W, r = par.world, par.rank
    n = 10007
    ids = np.random.RandomState(0).permutation(n)[:6000]
    mine = par.shard_ids(ids, n)
    flat = torch.arange(12, dtype=torch.float32) * (r + 1)
    par.allreduce_mean_(flat)
    # history: 40 rows per rank; vertex 5 is written by EVERY rank, vertex 6 by ranks 2 and 6, vertex 7 by ranks 0 and 7
    N, d = 400, 5
    rng = np.random.RandomState(100 + r)
    idx = (8 + r * 40 + np.arange(40)).astype(np.int32)          # disjoint ranges ...
    idx[0] = 5
    if r in (2, 6):
        idx[1] = 6
    if r in (0, 7):
        idx[2] = 7
    nrow = 40 - (r % 3)                                          # ... of ragged sizes
    idx, rows = idx[:nrow], rng.standard_normal((nrow, d)).astype(np.float32)

    def scatter(h, i, rr):
        onp.scatter_rows(h.numpy(), i.numpy(), rr.numpy())
    assert par.set_history_cap(48, d) is False                   # (gloo: never the library's communicator)
    H = torch.zeros((N, d))
    par.sync_history(H, torch.from_numpy(idx), torch.from_numpy(rows), scatter)
    par.sync_history(H, torch.from_numpy(idx), torch.from_numpy(rows * 2), scatter)     # two exchanges in flight
    assert not H.any()
    par.join_history()
    recv_words = max(b[1].numel() for b in par._hist_bufs.values())
    # the unbounded form (size exchange + padded gather) gives the same history
    par.history_cap = None
    H2 = torch.zeros((N, d))
    par.sync_history(H2, torch.from_numpy(idx), torch.from_numpy(rows * 2), scatter)
    par.join_history()
    FLAGS.reset()
    FLAGS.update(early_stopping=2, epochs=100, data=0)
    falling, rising = [1.0, 0.9, 0.8, 0.7, 0.6], [1.0, 0.9, 0.8, 0.9, 1.0]
    v1 = stop_verdict(par, 4, falling if r == 0 else rising, 10)            # rank 0 decides: go on
    v2 = stop_verdict(par, 4, rising if r == 0 else falling, 10)            # ... stop
    FLAGS.update(epochs=3, data=100)
    v3 = stop_verdict(par, 4, falling, 12)                                  # job-wide data 96 < 100
    v4 = stop_verdict(par, 4, falling, 13)                                  # 104 >= 100
    os.environ["LOCAL_WORLD_SIZE"] = str(W)
    return dict(mine=mine, flat=flat.numpy(), lo_hi=np.array(par.vertex_range(n)), H=H.numpy(), H2=H2.numpy(), idx=idx,
                rows=rows, recv_words=np.array([recv_words]), v=np.array([v1, v2, v3, v4]),
                packers=np.array([default_packers()]), seed=np.array([par.sampler_seed(123)]))


# ---- tests -----------------------------------------------------------------------------------
def test_eight_ranks_sharding_history_exchange_and_collective_decisions(tmp_path):
    """The paths the driver's 8-GPU run takes, at world size 8 on CPU (gloo) -- so that its first contact with eight ranks
    is RCCL itself and nothing of ours."""
    W = 8
    r = _run("w_eight", tmp_path, world=W)
    ids = np.random.RandomState(0).permutation(10007)[:6000]
    assert sorted(np.concatenate([x["mine"] for x in r]).tolist()) == sorted(ids.tolist())
    for k, x in enumerate(r):
        lo, hi = x["lo_hi"]
        assert lo == (10007 * k) // W and hi == (10007 * (k + 1)) // W
        assert len(x["mine"]) == 0 or (x["mine"].min() >= lo and x["mine"].max() < hi)
        np.testing.assert_allclose(x["flat"], np.arange(12, dtype=np.float32) * 4.5)        # mean of 1 .. 8
        assert x["seed"][0] == 123 + k
        assert x["v"].tolist() == [0, 1, 0, 2]
        assert x["recv_words"][0] == W * 48 * (5 + 1)                   # cap x world blocks of [ids | rows]
        assert 0 <= x["packers"][0] <= 3
    assert len({int(x["packers"][0]) for x in r}) == 1                  # every rank takes the same share of the host
    want = np.zeros((400, 5), np.float32)
    for x in r:                                                         # rank order; the second exchange wrote rows * 2
        want[x["idx"]] = x["rows"] * 2
    for x in r:
        np.testing.assert_array_equal(x["H"], want)                     # replicas bit-identical
        np.testing.assert_array_equal(x["H2"], want)
    np.testing.assert_array_equal(want[5], r[7]["rows"][0] * 2)         # eight writers: the highest rank's row stays
    np.testing.assert_array_equal(want[6], r[6]["rows"][1] * 2)
    np.testing.assert_array_equal(want[7], r[7]["rows"][2] * 2)


def test_default_packers_leaves_the_launching_thread_and_the_core_their_cores(monkeypatch):
    """scheduler.default_packers: three packers on a box of its own, what is left of a rank's share with 8 ranks on one
    host (none when the container grants 16 cores: 2 per rank)."""
    from stochastic_gcn_amd import scheduler
    monkeypatch.setattr(os, "sched_getaffinity", lambda _: set(range(16)), raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert 0 <= scheduler.default_packers() <= 3            # (3 unless the container's CPU quota is under five cores)
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert scheduler.default_packers() == 0
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert scheduler.default_packers() <= 2


def test_history_cap_decides_job_wide_whether_the_library_carries_the_exchange():
    """DataParallel.set_history_cap: the library's exchange moves a fixed cap x (d + 1) block per rank and layer;
    above HISTORY_FIXED_LIMIT_BYTES, or without a bound, every rank leaves the history exchange to torch.distributed --
    a function of job-wide constants only."""
    from stochastic_gcn_amd.parallel import DataParallel
    par = DataParallel(init=False)
    par.native = True                       # (as if the library's communicator were up)
    assert par.set_history_cap(512 * 2, 128) is True and par.native_history
    assert par.set_history_cap(232965, 128) is False and not par.native_history      # the bound saturated at the graph's size
    assert par.history_cap == 232965
    assert par.set_history_cap(None, 128) is False
    lim = DataParallel.HISTORY_FIXED_LIMIT_BYTES
    cap = lim // (129 * 4) // 4 * 4
    assert par.set_history_cap(cap, 128) is True and par.set_history_cap(cap + 4, 128) is False
    par.native = False
    assert par.set_history_cap(1024, 128) is False


def test_sharding_allreduce_broadcast(tmp_path):
    r = _run("w_shard_and_allreduce", tmp_path)
    ids = np.random.RandomState(0).permutation(1001)[:700]
    assert sorted(np.concatenate([r[0]["mine"], r[1]["mine"]]).tolist()) == sorted(ids.tolist())
    assert r[0]["lo_hi"].tolist() == [0, 500] and r[1]["lo_hi"].tolist() == [500, 1001]
    assert r[0]["mine"].max() < 500 <= r[1]["mine"].min()
    want = np.arange(10, dtype=np.float32) * 1.5           # mean of x*1 and x*2
    for x in r:
        np.testing.assert_allclose(x["flat"], want)
        np.testing.assert_array_equal(x["theta"], np.zeros(5))      # rank 0's weights everywhere
        assert x["mx"][0] == 4


def test_history_sync_is_replica_consistent_and_rank_ordered(tmp_path):
    r = _run("w_history", tmp_path)
    np.testing.assert_array_equal(r[0]["H"], r[1]["H"])
    want = np.zeros((50, 6), np.float32)
    for x in r:                                            # rank order: 0 then 1
        want[x["idx"]] = x["rows"]
    np.testing.assert_array_equal(r[0]["H"], want)
    np.testing.assert_array_equal(want[5], r[1]["rows"][0])         # higher rank wins the conflict
    np.testing.assert_array_equal(r[0]["H2"], want)                 # fixed-capacity path, -1 padding
    np.testing.assert_array_equal(r[1]["H2"], want)


def test_two_rank_training_step_matches_mean_gradient(tmp_path):
    r = _run("w_train_step", tmp_path)
    assert r[0]["n_train"][0] + r[1]["n_train"][0] == 64 * 3
    for step in range(2):
        mean = 0.5 * (r[0]["local_grad%d" % step] + r[1]["local_grad%d" % step])
        np.testing.assert_allclose(r[0]["avg_grad%d" % step], mean, rtol=1e-6, atol=1e-8)
        np.testing.assert_array_equal(r[0]["avg_grad%d" % step], r[1]["avg_grad%d" % step])
    np.testing.assert_array_equal(r[0]["theta"], r[1]["theta"])     # replicas stay in lock-step
    np.testing.assert_array_equal(r[0]["hist"], r[1]["hist"])
    assert np.abs(r[0]["hist"]).sum() > 0


def test_stop_decision_is_collective(tmp_path):
    r = _run("w_stop_verdict", tmp_path)
    for x in r:
        assert x["v"].tolist() == [0, 1, 0, 2]


def test_partition_rows_by_nnz_balances_power_law_rows():
    from stochastic_gcn_amd.parallel import partition_rows_by_nnz
    from stochastic_gcn_amd import synthetic
    a = synthetic.rmat_like(1 << 12, 40 << 12, seed=3)
    for world in (1, 2, 3, 8):
        b = partition_rows_by_nnz(a.indptr, world)
        assert b[0] == 0 and b[-1] == a.shape[0] and np.all(np.diff(b) >= 0)
        per = np.diff(a.indptr[b])
        assert per.sum() == a.nnz
        heaviest_row = int(np.diff(a.indptr).max())
        assert per.max() <= a.nnz / world + heaviest_row          # within one row of the ideal
    # more ranks than rows: empty ranges, still a partition
    b = partition_rows_by_nnz(np.array([0, 3, 4]), 5)
    assert b[0] == 0 and b[-1] == 2 and np.all(np.diff(b) >= 0)


def test_sharded_spmm_balances_load_not_nonzeros_alone():
    """ShardedSpMM's row blocks (kernel=None: the partition alone, no device): with the column sweep a row costs what
    ROW_WEIGHT of its nonzeros do, so the blocks of a skewed graph carry equal LOAD (nonzeros of both directions + the weight
    per row and direction) -- the hub block gets more nonzeros, the sparse tail fewer rows; row_weight = 0 is the plain
    nonzero balance."""
    import types
    from stochastic_gcn_amd import synthetic
    from stochastic_gcn_amd.parallel import ShardedSpMM
    a = synthetic.rmat_like(1 << 14, 30 << 14, seed=4)
    at = a.T.tocsr()
    world = 8
    both = np.diff(a.indptr) + np.diff(at.indptr)
    for w in (0, ShardedSpMM.ROW_WEIGHT):
        bounds, loads, nnzs = [0], [], []
        for r in range(world):
            sh = ShardedSpMM(types.SimpleNamespace(rank=r, world=world, active=False), a, 'cpu', kernel=None, row_weight=w)
            assert sh.row_weight == w and sh.lo == bounds[-1]
            bounds.append(sh.hi)
            loads.append(int(both[sh.lo:sh.hi].sum()) + 2 * w * (sh.hi - sh.lo))
            nnzs.append(int(both[sh.lo:sh.hi].sum()))
        assert bounds[-1] == a.shape[0]
        ideal = (int(both.sum()) + 2 * w * a.shape[0]) / world
        assert max(loads) <= ideal + int(both.max()) + 2 * w
        if w:
            assert nnzs[0] > 1.1 * nnzs[-2]           # the hub block takes more nonzeros than a sparse one
    # the default: the weight for the column sweep, none for the row-gather kernel's blocks
    ns = types.SimpleNamespace(rank=0, world=2, active=False)
    assert ShardedSpMM(ns, a, 'cpu', kernel=None).row_weight == 0
