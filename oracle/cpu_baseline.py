#!/usr/bin/env python
"""oracle/cpu_baseline.py -- TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg runs it as a
subprocess; never the product).

Times the oracle's OpenMP C restatement of the CSR x dense product (oracle/oracle_c.c, loop order of the
authors' own abandoned CPU kernel, gcn/history.cpp:10-48) on the host cores at several thread counts, as
a REAL CPU number: one process per thread count with OMP_NUM_THREADS / OMP_PROC_BIND=spread /
OMP_PLACES=cores set before the OpenMP runtime starts (threads pinned, spread over the sockets), the
dense operand first-touched page-interleaved over the NUMA nodes, a warm-up pass, then whole passes for a
bounded time.  Prints one JSON object.

    python oracle/cpu_baseline.py sample.npz d seconds threads
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    path, d, seconds, threads = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    lib = C.CDLL(os.path.join(HERE, "liboracle_c.so"))
    lib.oracle_max_threads.restype = C.c_int32
    P = C.c_void_p
    lib.oracle_first_touch_interleaved.argtypes = [P, C.c_int64]
    lib.oracle_spmm_csr_f32.argtypes = [P, P, P, C.c_int32, C.c_int32, P, C.c_int64, P, C.c_int64, C.c_float]
    z = np.load(path)
    rowptr, col, val, K = z["indptr"].astype(np.int32), z["indices"].astype(np.int32), z["data"].astype(np.float32), int(z["K"])
    M, nnz = rowptr.shape[0] - 1, int(col.shape[0])
    B = np.empty((K, d), dtype=np.float32)
    lib.oracle_first_touch_interleaved(B.ctypes.data, B.size)            # NUMA placement BEFORE the values
    rng = np.random.RandomState(0)
    for lo in range(0, K, 16384):                                        # fill in place (keeps the page placement)
        B[lo:lo + 16384] = rng.standard_normal((min(16384, K - lo), d)).astype(np.float32)
    Cm = np.empty((M, d), dtype=np.float32)
    lib.oracle_first_touch_interleaved(Cm.ctypes.data, Cm.size)

    def run():
        lib.oracle_spmm_csr_f32(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, M, d, B.ctypes.data, d,
                                Cm.ctypes.data, d, 0.0)
    run()                                                                # warm-up: threads, page tables
    t0, reps = time.perf_counter(), 0
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 200:
            break
    print(json.dumps({"threads": int(lib.oracle_max_threads()), "requested": threads, "edges_per_s": nnz * reps / el,
                      "reps": reps, "seconds": el, "rows": M, "edges": nnz, "checksum": float(Cm[::97].sum())}))


if __name__ == "__main__":
    main()
