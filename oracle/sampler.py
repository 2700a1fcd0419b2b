"""oracle/sampler.py -- TEST INFRASTRUCTURE ONLY.

Pure-Python restatement of the reference neighbour sampler, for small cases:
  * ``Mt19937`` / ``u01``      std::mt19937 + libstdc++ uniform_real_distribution<float>
                               (gcn/scheduler.cpp:8, gcn/scheduler.h:27)
  * ``Mult``                   gcn/mult.h:8-27, gcn/mult.cpp:7-51
  * ``Scheduler``              gcn/scheduler.h:6-28, gcn/scheduler.cpp:11-189
  * ``PyScheduler``            gcn/_scheduler.pyx:28-148

Pinned against the golden vectors produced by the real reference C++
(tests/golden/*.npz, tests/test_oracle_sampler.py).  Every fp32 expression is evaluated with
numpy.float32 scalars so each operation rounds exactly once, as in the reference's SSE build.
"""
import numpy as np

f32 = np.float32


class Mt19937(object):
    """std::mt19937 (32-bit Mersenne twister), seeded like mt19937::seed(value)."""

    def __init__(self, seed=5489):
        self.seed(seed)

    def seed(self, seed):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt, self.idx = mt, 624

    def _refill(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            v = mt[(i + 397) % 624] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            mt[i] = v
        self.idx = 0

    def next(self):
        if self.idx >= 624:
            self._refill()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def u01(self):
        """generate_canonical<float,24> on a 32-bit engine: one draw, float(x)/2^32 with
        float(x) rounded to nearest; results >= 1 become nextafter(1, 0)."""
        r = f32(self.next()) / f32(4294967296.0)
        return f32(0.99999994) if r >= f32(1.0) else r


class Mult(object):
    """gcn/mult.cpp:7-51."""

    def __init__(self, prob):
        self.prob = [f32(p) for p in prob]
        n = len(self.prob)
        N = n
        while N != (N & -N):                      # mult.cpp:9
            N += N & -N
        self.N = N
        self.bit = [f32(0)] * (N + 1)
        self.sum = f32(0)
        for i, p in enumerate(self.prob):
            self.add(i + 1, p)
        if n == 0:
            raise RuntimeError("Prob is empty")   # mult.cpp:17-18
        self.max_result = n - 1
        self.generator = Mt19937()                # default-seeded member (mult.h:25-26)

    def add(self, idx, val):
        while idx <= self.N:
            self.bit[idx] = f32(self.bit[idx] + val)
            idx += idx & -idx
        self.sum = f32(self.sum + val)

    def query_u(self, u):                         # mult.cpp:38-51
        u = f32(u)
        cur, step = 0, self.N
        while step > 0:
            if cur + step > self.N or self.bit[cur + step] > u:
                step //= 2
            else:
                u = f32(u - self.bit[cur + step])
                cur += step
                step //= 2
        return cur

    def query(self):                              # mult.cpp:29-36
        u = f32(self.generator.u01() * self.sum)
        r = min(self.query_u(u), self.max_result)
        self.add(r + 1, f32(-self.prob[r]))
        self.prob[r] = f32(0)
        return r


class Scheduler(object):
    """gcn/scheduler.cpp:11-189."""

    def __init__(self, adj_w, adj_i, adj_p, num_data, num_edges, L, cv, is_):
        self.cv, self.is_ = bool(cv), bool(is_)
        self.adj_w = [f32(x) for x in adj_w[:num_edges]]
        self.adj_i = [int(x) for x in adj_i[:num_edges]]
        self.adj_p = [int(x) for x in adj_p[:num_data]] + [num_edges]     # :16,20
        self.num_data = num_data
        self.visited = [-1] * num_data
        self.fvisited = [-1] * num_data
        if self.is_:                                                       # :22-25
            imp = [f32(1e-6)] * num_data
            for i in range(num_data):
                for p in range(int(adj_p[i]), int(adj_p[i + 1])):
                    w = f32(adj_w[p])
                    imp[int(adj_i[p])] = f32(imp[int(adj_i[p])] + f32(w * w))
            self.importance = imp
        else:
            self.importance = [f32(1)] * num_data                          # :32-33
        self.generator = Mt19937()
        self.field, self.ffield = [], []
        self._clear()

    def _clear(self):
        self.edg_s, self.edg_t, self.edg_w, self.medg_w = [], [], [], []
        self.fedg_s, self.fedg_t, self.fedg_w, self.scales = [], [], [], []

    def seed(self, s):
        self.generator.seed(int(s))

    def start_batch(self, data):
        self.field = [int(x) for x in data]

    def expand(self, degree):
        field, visited = self.field, self.visited
        new_field = list(field)                                            # :50
        self.ffield = []
        for i, v in enumerate(new_field):
            visited[v] = i                                                 # :51-52
        self._clear()
        if self.is_:
            return self._expand_is(degree, new_field)
        for i, s in enumerate(field):                                      # :126
            lo, hi = self.adj_p[s], self.adj_p[s + 1]
            adj_range = hi - lo
            adj_size = min(adj_range, degree)
            scale = f32(1) if adj_range == 0 else f32(f32(adj_range) / f32(adj_size))   # :132-133
            self.scales.append(f32(1.0 / float(np.sqrt(scale))))           # :134 (fp32 sqrt, fp64 div)
            for it in range(adj_size):
                num_remaining = adj_range - it
                pos = f32(f32(it) + f32(f32(num_remaining) * self.generator.u01()))     # :141
                idx = min(int(pos), adj_range - 1)
                a, b = lo + it, lo + idx
                self.adj_i[a], self.adj_i[b] = self.adj_i[b], self.adj_i[a]             # :144
                self.adj_w[a], self.adj_w[b] = self.adj_w[b], self.adj_w[a]             # :145
                t = self.adj_i[a]
                w = f32(self.adj_w[a] * scale)
                if visited[t] == -1:
                    visited[t] = len(new_field)
                    new_field.append(t)
                self.edg_s.append(i)
                self.edg_t.append(visited[t])
                self.edg_w.append(w)
                if self.cv:
                    self.medg_w.append(f32(self.adj_w[a] * w))                          # :164
            if self.cv:                                                                 # :167-179
                for it in range(adj_range):
                    t = self.adj_i[lo + it]
                    if self.fvisited[t] == -1:
                        self.fvisited[t] = len(self.ffield)
                        self.ffield.append(t)
                    self.fedg_s.append(i)
                    self.fedg_t.append(self.fvisited[t])
                    self.fedg_w.append(self.adj_w[lo + it])
        self.field = new_field
        for s in self.field:
            visited[s] = -1
        if self.cv:
            for s in self.ffield:
                self.fvisited[s] = -1

    def _expand_is(self, degree, new_field):                              # :63-123
        field, visited = self.field, self.visited
        neighbors, probs = [], []
        v2 = [False] * self.num_data
        times = [0] * self.num_data
        total = f32(0)
        for i in field:
            for p in range(self.adj_p[i], self.adj_p[i + 1]):
                t = self.adj_i[p]
                if not v2[t]:
                    v2[t] = True
                    neighbors.append(t)
                    total = f32(total + self.importance[t])
                    probs.append(self.importance[t])
        mult = Mult(probs)
        num_samples = min(len(field) * degree, len(neighbors))
        for _ in range(num_samples):
            t = neighbors[mult.query()]
            times[t] += 1
            if visited[t] == -1:
                visited[t] = len(new_field)
                new_field.append(t)
        for i, s in enumerate(field):
            for p in range(self.adj_p[s], self.adj_p[s + 1]):
                t = self.adj_i[p]
                if times[t]:
                    num = f32(f32(f32(times[t]) * self.adj_w[p]) * total)
                    den = f32(self.importance[t] * f32(num_samples))
                    self.edg_s.append(i)
                    self.edg_t.append(visited[t])
                    self.edg_w.append(f32(num / den))
        self.field = new_field
        for s in self.field:
            visited[s] = -1


class PyScheduler(object):
    """gcn/_scheduler.pyx:28-148."""

    def __init__(self, adj, labels, L, degrees, placeholders, seed, data=None, cv=False,
                 importance=False):
        self.c_sch = Scheduler(adj.data, adj.indices, adj.indptr, labels.shape[0],
                               adj.data.shape[0], L, cv, importance)
        self.c_sch.seed(seed)
        self.labels, self.data, self.degrees, self.L = labels, data, degrees, L
        self.placeholders, self.start = placeholders, 0

    def batch(self, data):
        s = self.c_sch
        i32 = lambda x: np.asarray(x, dtype=np.int32)           # noqa: E731
        fl = lambda x: np.asarray(x, dtype=np.float32)          # noqa: E731
        fields, ffields, adjs, madjs, fadjs, scales = [i32(data)], [], [], [], [], []
        s.start_batch(data)
        for l in range(self.L):
            s.expand(int(self.degrees[self.L - l - 1]))
            fields.append(i32(s.field))
            scales.append(fl(s.scales))
            edg_i = np.stack([i32(s.edg_s), i32(s.edg_t)], axis=1).reshape(-1, 2)
            shape = (fields[-2].shape[0], fields[-1].shape[0])
            adjs.append((edg_i, fl(s.edg_w), shape))
            if s.cv:
                ffields.append(i32(s.ffield))
                fedg_i = np.stack([i32(s.fedg_s), i32(s.fedg_t)], axis=1).reshape(-1, 2)
                madjs.append((edg_i.copy(), fl(s.medg_w), np.copy(shape)))
                fadjs.append((fedg_i, fl(s.fedg_w), (fields[-2].shape[0], ffields[-1].shape[0])))
        for lst in (fields, ffields, adjs, madjs, fadjs, scales):
            lst.reverse()
        ph = self.placeholders
        fd = {ph['adj'][i]: adjs[i] for i in range(self.L)}
        fd.update({ph['scales'][i]: scales[i] for i in range(len(scales))})
        if s.cv:
            fd.update({ph['madj'][i]: madjs[i] for i in range(len(madjs))})
            fd.update({ph['fadj'][i]: fadjs[i] for i in range(len(fadjs))})
            fd.update({ph['ffields'][i]: ffields[i] for i in range(len(ffields))})
        fd[ph['labels']] = self.labels[fields[-1]]
        for i in range(self.L + 1):
            fd[ph['fields'][i]] = fields[i]
        return fd
