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
