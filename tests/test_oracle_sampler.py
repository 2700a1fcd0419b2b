"""Pins the pure-Python sampler restatement (oracle/sampler.py) against the golden vectors
captured from the real reference C++ -- the oracle is only trusted because of this file."""
import numpy as np

import golden_util as gu
from oracle import sampler as osamp


def _run(z, gname, max_cases):
    adj = gu.graph(z, gname)
    labels = np.zeros((adj.shape[0], 2), np.float32)
    n = 0
    for case, nit in gu.cases(z, gname)[:max_cases]:
        c = gu.parse_case(case)
        sch = osamp.PyScheduler(adj, labels, c['L'], [c['deg']] * c['L'], gu.placeholders(c['L']),
                                c['seed'], cv=c['cv'], importance=c['imp'])
        for it in range(nit):
            prefix = "%s/%s/it%d" % (gname, case, it)
            gu.check_feed_against_golden(z, prefix, sch.batch(z[prefix + "/ids"]), False)
            n += 1
        np.testing.assert_array_equal(z["%s/%s/adj_i_after" % (gname, case)],
                                      np.asarray(sch.c_sch.adj_i, dtype=np.int32))
    return n


def test_oracle_sampler_tree():
    assert _run(gu.load("sampler_small.npz"), "tree", 10 ** 6) >= 200


def test_oracle_sampler_rand50():
    assert _run(gu.load("sampler_small.npz"), "rand50", 10 ** 6) >= 200


def test_oracle_mt19937_matches_numpy_legacy_seeding():
    for seed in (0, 1, 123, 5489):
        g = osamp.Mt19937(seed)
        ours = [g.next() for _ in range(1500)]
        ref = np.random.RandomState(seed)._bit_generator.random_raw(1500)
        assert ours == [int(x) for x in ref]


def test_oracle_mult_golden():
    z = gu.load("mult.npz")
    for n in sorted({k.split("/")[0] for k in z.files}):
        m = osamp.Mult(z[n + "/prob"])
        assert gu.bits_equal(np.asarray(m.bit, np.float32), z[n + "/bit"]), n
        assert [m.query_u(u) for u in z[n + "/u"]] == z[n + "/query_u"].tolist()
        assert [m.query() for _ in range(len(z[n + "/prob"]))] == z[n + "/draws"].tolist()
