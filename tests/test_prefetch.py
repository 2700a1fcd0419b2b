"""Host-side batch pipelines of stochastic_gcn_amd/train.py (no GPU): the in-order multi-sampler
prefetcher of the non-parity fast mode, and the wrap-around batch walk."""
import numpy as np

from stochastic_gcn_amd import synthetic
from stochastic_gcn_amd.scheduler import PyScheduler, StagingSlot
from stochastic_gcn_amd.train import ParallelPrefetcher, epoch_batches

PH = {'adj': ['a0', 'a1'], 'madj': ['m0', 'm1'], 'fadj': ['f0', 'f1'], 'fields': ['x0', 'x1', 'x2'],
      'ffields': ['ff0', 'ff1'], 'scales': ['s0', 's1'], 'labels': 'l'}


def _graph(n=600):
    a = synthetic.rmat_like(n, 12 * n, seed=2)
    labels = np.eye(4, dtype=np.float32)[np.arange(n) % 4]
    return a, labels


def test_epoch_batches_wraps_like_minibatch():
    data = np.arange(10, dtype=np.int32)
    b = epoch_batches(data, 4, 5)
    assert [x.tolist() for x in b] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9], [0, 1, 2, 3], [4, 5, 6, 7]]


def test_parallel_prefetcher_delivers_valid_batches_in_order():
    a, labels = _graph()
    L, deg = 2, np.array([3, 3], dtype=np.int32)
    schs = [PyScheduler(a, labels, L, deg, PH, 7 + 1000 * k, cv=True) for k in range(3)]
    ids = np.random.RandomState(0).permutation(a.shape[0]).astype(np.int32)
    batches = epoch_batches(ids, 32, 25)
    slots = [StagingSlot(pin=False) for _ in range(2 * 3 + 3)]
    pre = ParallelPrefetcher(schs, batches, 0, slots, depth=2)
    indptr, indices = a.indptr, a.indices
    for i, want in enumerate(batches):
        pb = pre.next()
        np.testing.assert_array_equal(pb.field(L), want)              # batch i, in order
        fd = pb.feed_dict(PH)
        for l in range(L):
            idx, w, shape = fd[PH['adj'][l]]
            f_in, f_out = fd[PH['fields'][l]], fd[PH['fields'][l + 1]]
            assert shape[0] == len(f_out) and shape[1] == len(f_in)
            # every sampled edge is a real edge of the graph
            src, dst = f_out[idx[:, 0]], f_in[idx[:, 1]]
            for s_, d_ in zip(src[:50], dst[:50]):
                assert s_ == d_ or d_ in indices[indptr[s_]:indptr[s_ + 1]]
    assert pre.next() is None


def test_single_sampler_prefetcher_is_the_sequential_sequence():
    a, labels = _graph()
    deg = np.array([2], dtype=np.int32)
    ids = np.arange(a.shape[0], dtype=np.int32)
    batches = epoch_batches(ids, 50, 6)
    seq = PyScheduler(a, labels, 1, deg, PH, 3, cv=True)
    ref = [seq.batch_packed(b) for b in batches]
    one = PyScheduler(a, labels, 1, deg, PH, 3, cv=True)
    pre = ParallelPrefetcher([one], batches, 0, [StagingSlot(pin=False) for _ in range(5)], depth=2)
    for r in ref:
        pb = pre.next()
        assert pb.n_i == r.n_i and pb.n_f == r.n_f
        np.testing.assert_array_equal(np.asarray(pb.ibuf[:pb.n_i]), np.asarray(r.ibuf[:r.n_i]))
        np.testing.assert_array_equal(np.asarray(pb.fbuf[:pb.n_f]), np.asarray(r.fbuf[:r.n_f]))


def test_native_prefetcher_is_the_sequential_sequence_and_survives_small_slots():
    """The C++ sampler threads (sgcn_prefetch_*) yield exactly the batches the synchronous loop yields -- one thread
    sampling and packing (packers = 0) or the sampler's core on one thread feeding 1-3 packer threads -- also when a
    batch outgrows its staging slot (spill path) and when the consumer stops early."""
    from stochastic_gcn_amd.scheduler import NativePrefetcher
    a, labels = _graph()
    deg = np.array([2, 3], dtype=np.int32)
    ids = np.random.RandomState(3).permutation(a.shape[0]).astype(np.int32)
    batches = epoch_batches(ids, 40, 17)
    seq = PyScheduler(a, labels, 2, deg, PH, 5, cv=True)
    ref = [seq.batch_packed(b) for b in batches]
    for words, packers in ((1 << 20, 0), (300, 0), (1 << 20, 1), (1 << 20, 2), (300, 2), (1 << 20, 3), (1 << 20, None)):
        sch = PyScheduler(a, labels, 2, deg, PH, 5, cv=True)      # roomy slots / every batch spills
        sch._slot_words = words
        pre = NativePrefetcher(sch, batches, 0, depth=2, pin=False, packers=packers)
        assert pre.packers == (3 if packers is None else packers)
        for r in ref:
            pb = pre.next()
            assert (pb.n_i, pb.n_f) == (r.n_i, r.n_f)
            np.testing.assert_array_equal(pb.meta, r.meta)
            np.testing.assert_array_equal(np.asarray(pb.ibuf[:pb.n_i]), np.asarray(r.ibuf[:r.n_i]))
            np.testing.assert_array_equal(np.asarray(pb.fbuf[:pb.n_f]), np.asarray(r.fbuf[:r.n_f]))
        assert pre.next() is None and pre.next() is None
        if words == 300:
            assert sch._slot_words > 300                  # the next epoch gets bigger slots
        assert set(pre.stats) >= {'wait_slot_s', 'pack_s', 'copy_s', 'sample_s'} and (pre.stats['sample_s'] > 0) == (pre.packers > 0)
    # early stop: the threads are joined without draining the epoch
    for packers in (0, 2):
        sch = PyScheduler(a, labels, 2, deg, PH, 5, cv=True)
        pre = NativePrefetcher(sch, batches, 0, depth=2, pin=False, packers=packers)
        assert pre.next() is not None
        pre.close()
        # and the sampler is usable again afterwards
        assert sch.batch_packed(batches[0]) is not None


def test_native_multi_sampler_delivers_valid_batches_in_order():
    """N sampler threads in C++ (the non-parity fast mode): batch i is the i-th id slice, each
    batch equals what the sampler that built it yields when run alone on its own slices."""
    from stochastic_gcn_amd.scheduler import NativePrefetcher
    a, labels = _graph()
    deg = np.array([3], dtype=np.int32)
    ids = np.random.RandomState(9).permutation(a.shape[0]).astype(np.int32)
    batches = epoch_batches(ids, 32, 23)
    N = 3
    mk = lambda: [PyScheduler(a, labels, 1, deg, PH, 11 + 1000 * k, cv=True) for k in range(N)]   # noqa: E731
    solo = mk()
    want = {}
    for k in range(N):                                   # sampler k alone on batches k, k+N, ...
        for i in range(k, len(batches), N):
            want[i] = solo[k].batch_packed(batches[i])
    pre = NativePrefetcher(mk(), batches, 0, depth=2, pin=False)
    for i in range(len(batches)):
        pb = pre.next()
        np.testing.assert_array_equal(pb.field(1), batches[i])
        r = want[i]
        assert (pb.n_i, pb.n_f) == (r.n_i, r.n_f)
        np.testing.assert_array_equal(np.asarray(pb.ibuf[:pb.n_i]), np.asarray(r.ibuf[:r.n_i]))
        np.testing.assert_array_equal(np.asarray(pb.fbuf[:pb.n_f]), np.asarray(r.fbuf[:r.n_f]))
    assert pre.next() is None


def test_native_prefetcher_stress_no_deadlock():
    """Many short epochs with random early stops, sampler counts and depths: every epoch either
    drains or closes cleanly (pytest-timeout guards against a hang)."""
    from stochastic_gcn_amd.scheduler import NativePrefetcher
    a, labels = _graph(300)
    deg = np.array([2], dtype=np.int32)
    rng = np.random.RandomState(0)
    schs = [PyScheduler(a, labels, 1, deg, PH, 5 + k, cv=True) for k in range(4)]
    ids = np.arange(a.shape[0], dtype=np.int32)
    for it in range(120):
        nb = int(rng.randint(0, 25))
        batches = epoch_batches(ids, int(rng.randint(1, 40)), nb)
        n = int(rng.randint(1, 5))
        pre = NativePrefetcher(schs[:n] if n > 1 else schs[0], batches, 0, depth=int(rng.randint(1, 4)), pin=False,
                               lag=int(rng.randint(0, 3)), packers=int(rng.randint(0, 4)))
        take = nb if rng.rand() < 0.5 else int(rng.randint(0, nb + 1))
        for i in range(take):
            pb = pre.next()
            assert pb is not None
            np.testing.assert_array_equal(pb.field(1), batches[i])
        if take == nb:
            assert pre.next() is None
        pre.close()


def test_sampler_error_reaches_the_consumer_through_every_producer_shape():
    """A batch the importance sampler cannot expand (only isolated vertices: 'Prob is empty', gcn/mult.cpp:17-18) is an error
    of the call that asks for it -- also when the sampler's core and the packer run on different threads."""
    import pytest
    import scipy.sparse as sp
    from stochastic_gcn_amd._ffi import SgcnError
    from stochastic_gcn_amd.scheduler import NativePrefetcher
    n = 64
    rows = np.arange(0, 40)
    a = sp.csr_matrix((np.ones(40, np.float32), (rows, (rows + 1) % 40)), shape=(n, n))      # vertices 40..63 are isolated
    labels = np.eye(4, dtype=np.float32)[np.arange(n) % 4]
    ph = {'adj': ['a0'], 'madj': ['m0'], 'fadj': ['f0'], 'fields': ['x0', 'x1'], 'ffields': ['ff0'], 'scales': ['s0'], 'labels': 'l'}
    batches = [np.arange(0, 8, dtype=np.int32), np.arange(8, 16, dtype=np.int32), np.arange(48, 56, dtype=np.int32),
               np.arange(16, 24, dtype=np.int32)]
    for packers in (0, 1, 3):
        sch = PyScheduler(a, labels, 1, np.array([2], dtype=np.int32), ph, 3, cv=False, importance=True)
        pre = NativePrefetcher(sch, batches, 0, depth=2, pin=False, packers=packers)
        assert pre.next() is not None and pre.next() is not None
        with pytest.raises(SgcnError) as e:
            pre.next()
        assert "Prob is empty" in str(e.value)
        pre.close()
