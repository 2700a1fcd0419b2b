"""L2 model of the column sweep: replays the B-row cache lines one XCD's 512 resident waves request during one
launch, in clock-locked order, through a set-associative LRU of the XCD's L2 size (profiles/l2_sweep_sim.cpp) and
reports how many times each line is fetched.  It reproduces the hardware counters of the two-group plan WITHOUT
alignment (1.4 fetches per line at 4 MiB, 2.4 at an effective 2 MiB -- 1.7 / 2.7 with unaligned 2,408-byte rows;
measured 1.6 clock-locked, 2.6 at speed)
and shows the cause: the k-th entries of a wave's two bins sit 2,600 columns apart on average (p99 9,500), so the
wave's two halves gather from windows that far apart; with `align` <= 2048 the multiplicity is 1.00.

    g++ -O2 -o /tmp/l2_sweep_sim profiles/l2_sweep_sim.cpp && python profiles/l2_sweep_sim.py
"""
import numpy as np, ctypes as C, sys, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_gcn_amd import synthetic
from stochastic_gcn_amd._ffi import lib, check
data = synthetic.reddit_like(seed=1, with_features=False)
a = data[2].tocsr()
rowptr = np.ascontiguousarray(a.indptr, dtype=np.int32); col = np.ascontiguousarray(a.indices, dtype=np.int32)
val = np.ascontiguousarray(a.data, dtype=np.float32)
M = rowptr.shape[0]-1
PITCH = 2432           # bench layout: 608 floats, 19 whole lines per row
SIM = os.environ.get('L2_SWEEP_SIM', '/tmp/l2_sweep_sim')

def plan_g1():
    nt, nfix, nslots = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.sgcn_csplan_count(rowptr.ctypes.data, M, 16, 0, None, C.byref(nt), C.byref(nfix), C.byref(nslots)))
    tile_ptr = np.empty(nt.value + 1, dtype=np.int64); colrow = np.empty(col.shape[0], dtype=np.int32)
    valout = np.empty(col.shape[0], dtype=np.float32); tr = np.empty(nt.value*16, np.int32); ts = np.empty(nt.value*16, np.int32)
    fix = np.empty((max(nfix.value,1),3), np.int32)
    check(lib.sgcn_csplan_fill(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, 16, 0, None, tile_ptr.ctypes.data,
          colrow.ctypes.data, valout.ctypes.data, tr.ctypes.data, ts.ctypes.data, fix.ctypes.data))
    return tile_ptr, colrow & ((1<<28)-1)

def plan_g2(align):
    nt, ne, nfix, nslots = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.sgcn_csplan2_count(rowptr.ctypes.data, col.ctypes.data, M, 0, 4096, align, C.byref(nt), C.byref(ne), C.byref(nfix), C.byref(nslots)))
    tile_ptr = np.empty(nt.value + 1, dtype=np.int64); colrow = np.empty(ne.value, dtype=np.int32)
    valout = np.empty(ne.value, dtype=np.float32); tr = np.empty(nt.value*32, np.int32); ts = np.empty(nt.value*32, np.int32)
    fix = np.empty((max(nfix.value,1),3), np.int32)
    check(lib.sgcn_csplan2_fill(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, 0, 4096, align, tile_ptr.ctypes.data,
          colrow.ctypes.data, valout.ctypes.data, tr.ctypes.data, ts.ctypes.data, fix.ctypes.data))
    return tile_ptr, colrow & ((1<<28)-1)

def stream(tile_ptr, cols, G, U, slack, piece_off, piece_bytes, xcd=0, rnd=0, jitter=0.0, seed=0):
    """access stream of XCD `xcd` in launch `rnd`, clock-locked: a batch of U steps issues when the clock
    reaches (first column of the batch - slack); + optional random per-wave lag (columns)"""
    rng = np.random.default_rng(seed)
    times, lines = [], []
    for b in range(xcd, 1024, 8):
        for w in range(4):
            t = rnd*4096 + 4*b + w
            if t + 1 >= tile_ptr.shape[0]: continue
            c = cols[tile_ptr[t]:tile_ptr[t+1]].astype(np.int64)
            if c.size == 0: continue
            if G == 2:
                steps = c.size // 2
                first = c[0::2][:steps]                      # group 0's column of each step
            else:
                steps = c.size
                first = c
            # batch index of each step (chunks of 64 entries restart the batching like the kernel)
            per_chunk = 64 // G
            k = np.arange(steps)
            batch_first = (k // per_chunk) * per_chunk + ((k % per_chunk) // U) * U
            tt = first[batch_first].astype(np.float64) - slack
            tt = np.maximum.accumulate(tt)
            if jitter: tt = tt + rng.uniform(0, jitter)
            tt = np.repeat(tt, G)
            addr0 = c[:steps*G] * PITCH + piece_off
            l0 = addr0 // 128; l1 = (addr0 + piece_bytes - 1) // 128
            nl = (l1 - l0 + 1)
            mx = int(nl.max())
            for j in range(mx):
                m = nl > j
                times.append(tt[m] + 1e-6 * j); lines.append((l0 + j)[m])
    times = np.concatenate(times); lines = np.concatenate(lines)
    o = np.argsort(times, kind='stable')
    return lines[o].astype(np.uint64)

def run(name, s, cap=32768, ways=16, hash=0):
    s.tofile('/tmp/l2_sweep_stream.bin')
    out = subprocess.check_output([SIM, '/tmp/l2_sweep_stream.bin', str(cap), str(ways), str(hash)]).decode().strip()
    uniq = np.unique(s).size
    tot, miss = int(out.split()[1]), int(out.split()[3])
    print("%-40s accesses %9d unique %8d misses %9d  multiplicity %.2f" % (name, tot, uniq, miss, miss/uniq), flush=True)

if __name__ == '__main__':
    tp1, c1 = plan_g1()
    tp2, c2 = plan_g2(0)
    d = []
    for t in range(0, 4096, 37):
        c = c2[tp2[t]:tp2[t+1]].astype(np.int64)
        d.append(np.abs(c[0::2] - c[1::2]))
    d = np.concatenate(d)
    print("G=2, no alignment: |column of bin 0 - column of bin 1| per step: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d"
          % (d.mean(), np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max()))
    plans = {al: plan_g2(al) for al in (1024, 2048, 4096)}
    for cap in (32768, 16384):
        for jit in (0, 2000):
            print("L2 capacity %d lines, per-wave lag up to %d columns" % (cap, jit))
            run("  G=1, 1216-byte pieces", stream(tp1, c1, 1, 4, 512, 0, 1216, jitter=jit), cap)
            run("  G=2 unaligned, 512-byte pieces", stream(tp2, c2, 2, 8, 512, 0, 512, jitter=jit), cap)
            for al, (tp, c) in plans.items():
                run("  G=2 align %d" % al, stream(tp, c, 2, 8, 512, 0, 512, jitter=jit), cap)
