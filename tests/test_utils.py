"""Host-side metric bookkeeping (no GPU)."""
import numpy as np
import pytest

from stochastic_gcn_amd.utils import calc_f1, f1_from_classes


@pytest.mark.parametrize("seed", range(6))
def test_f1_from_classes_is_sklearns_f1_score(seed):
    """The single-label F1 pair from class indices (what an evaluation sweep now brings to the host: the loss kernel's
    argmax(pred) and argmax(labels) per row) against gcn/utils.py:521-529's sklearn.metrics.f1_score on the one-hot /
    probability matrices: the same doubles."""
    rng = np.random.RandomState(seed)
    n, c = (5000, 41) if seed < 4 else (50, 7)
    labels = np.zeros((n, c), np.float32)
    present = rng.choice(c, size=max(2, c - seed), replace=False)          # some classes never occur
    labels[np.arange(n), rng.choice(present, n)] = 1
    pred = rng.rand(n, c).astype(np.float32)
    pred[np.arange(n), labels.argmax(1)] += 0.6 * (rng.rand(n) < 0.7)      # ~70 % correct
    if seed == 5:
        pred[:] = 0
        pred[:, 3] = 1                                                     # one predicted class only
    want = calc_f1(pred, labels, False)
    got = f1_from_classes(labels.argmax(1), pred.argmax(1))
    assert got[0] == pytest.approx(want[0], abs=1e-15) and got[1] == pytest.approx(want[1], abs=1e-15)
    assert isinstance(got[0], float) and isinstance(got[1], float)


def test_cached_graph_verifies_the_digest_it_stores(tmp_path, monkeypatch):
    """A cache file under $TMPDIR is builder-writable state: a file whose arrays no longer hash to the stored
    SHA-256 -- or, for a graph with a committed digest, to THAT digest -- is rebuilt, never trusted."""
    import scipy.sparse as sp
    from stochastic_gcn_amd import synthetic
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    calls = []

    def build():
        calls.append(1)
        return synthetic.rmat_like(300, 2000, seed=3)

    a = synthetic.cached_graph("t_small", build)
    b = synthetic.cached_graph("t_small", build)                     # second call: from the file
    assert len(calls) == 1 and synthetic.graph_digest(a) == synthetic.graph_digest(b)
    path = tmp_path / "sgcn_graphs" / "t_small.npz"
    z = dict(np.load(path))
    assert str(z["sha256"]) == synthetic.graph_digest(a)
    z["data"] = z["data"].copy(); z["data"][7] += 1e-3               # same shape, same nnz, one value off
    np.savez(open(path, "wb"), **z)
    c = synthetic.cached_graph("t_small", build)
    assert len(calls) == 2 and synthetic.graph_digest(c) == synthetic.graph_digest(a)
    # a self-consistent file of the WRONG graph does not pass a committed digest
    monkeypatch.setitem(synthetic.KNOWN_DIGESTS, "t_small", synthetic.graph_digest(a))
    other = synthetic.rmat_like(300, 2000, seed=4)
    np.savez(open(path, "wb"), data=other.data, indices=other.indices, indptr=other.indptr,
             shape=np.array(other.shape, np.int64), sha256=np.array(synthetic.graph_digest(other)))
    e = synthetic.cached_graph("t_small", build)
    assert len(calls) == 3 and synthetic.graph_digest(e) == synthetic.graph_digest(a)
    # and a generator that no longer reproduces the committed digest is an error, not a silent new graph
    monkeypatch.setitem(synthetic.KNOWN_DIGESTS, "t_small", "0" * 64)
    with pytest.raises(RuntimeError):
        synthetic.cached_graph("t_small", build)
