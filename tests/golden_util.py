"""Helpers shared by the golden-vector tests."""
import hashlib
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def placeholders(L):
    return {'adj': ['adj_%d' % i for i in range(L)], 'madj': ['madj_%d' % i for i in range(L)],
            'fadj': ['fadj_%d' % i for i in range(L)],
            'fields': ['fields_%d' % i for i in range(L + 1)],
            'ffields': ['ffields_%d' % i for i in range(L + 1)],
            'scales': ['scales_%d' % i for i in range(L)], 'labels': 'labels'}


def graph(z, gname):
    ip, ix, d = z[gname + "/indptr"], z[gname + "/indices"], z[gname + "/data"]
    n = ip.shape[0] - 1
    return sp.csr_matrix((d, ix, ip), shape=(n, n))


def cases(z, gname):
    """{case: [it0, it1, it2]} -> sorted list of (case, n_iters)."""
    out = {}
    pre = gname + "/"
    for k in z.files:
        if not k.startswith(pre) or "/it" not in k:
            continue
        case = k[len(pre):].split("/")[0]
        it = int(k[len(pre):].split("/")[1][2:])
        out[case] = max(out.get(case, 0), it + 1)
    return sorted(out.items())


def parse_case(case):
    parts = case.split("_")
    return dict(seed=int(parts[0][1:]), cv=bool(int(parts[1][2:])), imp=bool(int(parts[2][2:])),
                L=int(parts[3][1:]), deg=int(parts[4][1:]))


def flatten_feed(fd):
    flat = {}
    for k, v in fd.items():
        if isinstance(k, tuple):       # ('csr', placeholder): product-only extra
            continue
        if isinstance(v, tuple):
            flat["%s/idx" % k] = np.asarray(v[0], dtype=np.int32).reshape(-1, 2)
            flat["%s/w" % k] = np.asarray(v[1], dtype=np.float32)
            flat["%s/shape" % k] = np.asarray(v[2], dtype=np.int64)
        else:
            flat[str(k)] = np.asarray(v)
    return flat


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def check_feed_against_golden(z, prefix, fd, big):
    """Bit-exact comparison of a feed-dict with the stored reference output."""
    flat = flatten_feed(fd)
    stored = [k for k in z.files if k.startswith(prefix + "/") and not k.endswith("/ids")]
    names = set()
    for k in stored:
        names.add(k[len(prefix) + 1:].split("#")[0])
    assert names == set(flat.keys()), (prefix, sorted(names ^ set(flat.keys())))
    for name, arr in flat.items():
        if big:
            assert tuple(z["%s/%s#shape" % (prefix, name)]) == arr.shape, (prefix, name)
            assert np.array_equal(z["%s/%s#sha" % (prefix, name)], digest(_canon(arr, name))), (prefix, name)
        else:
            ref = z["%s/%s" % (prefix, name)]
            assert bits_equal(_canon(arr, name), ref), (prefix, name, arr, ref)


def _canon(arr, name):
    if name.endswith("/idx"):
        return np.ascontiguousarray(arr, dtype=np.int32).reshape(-1, 2)
    if name.endswith("/w"):
        return np.ascontiguousarray(arr, dtype=np.float32)
    if name.endswith("/shape"):
        return np.ascontiguousarray(arr, dtype=np.int64)
    return np.ascontiguousarray(arr)
