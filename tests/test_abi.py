"""The C-ABI library loads without a GPU and exports exactly what include/sgcn.h declares."""
import ctypes
import os
import re

from stochastic_gcn_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "sgcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(sgcn_[a-z0-9_]+)\s*\(", src))


def test_header_symbols_are_exported_and_bound():
    syms = _header_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libsgcn.so does not export %s" % s
    assert syms == set(_ffi.SIGNATURES.keys()), sorted(syms ^ set(_ffi.SIGNATURES.keys()))


def test_status_codes_and_error_message_without_gpu():
    assert _ffi.lib.sgcn_abi_version() == _ffi.ABI_VERSION == 16
    rc = _ffi.lib.sgcn_tune(b"no_such_knob", 1)
    assert rc == -1
    assert b"unknown key" in _ffi.lib.sgcn_last_error()
    assert _ffi.lib.sgcn_tune_get(b"step_fuse") == 127 and _ffi.lib.sgcn_tune(b"step_fuse", 128) == -1
    # argument validation happens before any HIP call, so it is testable on CPU
    rc = _ffi.lib.sgcn_spmm_csr_f32(None, None, None, 4, 4, 8, None, 8, None, None, None, None, 8,
                                    0.0, None, None)
    assert rc == -1 and b"null operand" in _ffi.lib.sgcn_last_error()
    rc = _ffi.lib.sgcn_gather_rows_f32(None, 8, None, -1, 8, None, 8, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    import numpy as np
    import pytest
    import scipy.sparse as sp
    import torch
    from stochastic_gcn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback|HBM"):
        ops.gather_rows(torch.zeros(4, 8), torch.zeros(2, dtype=torch.int32))
    a = sp.identity(4, format='csr', dtype=np.float32)
    A = ops.DeviceCSR.from_scipy(a, torch.device("cpu"), with_plan=False)
    with pytest.raises(RuntimeError, match="HBM"):
        ops.spmm(A, torch.zeros(4, 8))


def test_step_fill_is_the_numpy_expression_it_replaces():
    """sgcn_step_fill (host only): the slot table of a minibatch -- sizes and addresses affine in the descriptor table, the
    capacity checks in front of it, the step's dropout keys (ops.dropout_key) and the Adam step size -- against the NumPy
    form the launching thread evaluated before ABI v8."""
    import numpy as np
    from stochastic_gcn_amd import ops
    rng = np.random.RandomState(0)
    meta = rng.randint(0, 5000, 64).astype(np.int64)
    n, nslots = 40, 47
    idx = rng.randint(0, 64, n).astype(np.int64)
    mul = rng.choice([0, 1, 4, 512], n).astype(np.int64)
    base = rng.randint(0, 3, n).astype(np.int64)
    cap_idx, cap_max = np.array([3, 9], np.int64), np.array([meta[3], meta[9] + 5], np.int64)
    ws_idx, ws_ld = np.array([11], np.int64), np.array([128], np.int64)
    key_slot, key_layer = np.array([40, 41, 42, 43], np.int64), np.array([0, 2, 5, 4096 + 3], np.int64)
    f = _ffi.StepFill(n=n, idx=idx.ctypes.data, mul=mul.ctypes.data, base=base.ctypes.data,
                      n_cap=2, cap_idx=cap_idx.ctypes.data, cap_max=cap_max.ctypes.data,
                      n_ws=1, ws_idx=ws_idx.ctypes.data, ws_ld=ws_ld.ctypes.data, ws_floats=int(meta[11]) * 128,
                      n_keys=4, key_slot=key_slot.ctypes.data, key_layer=key_layer.ctypes.data, lr_slot=46)
    ip, fp = 0x7F0012340000, 0x7F0012380000
    for seed, step, lr in ((123, 0, 1e-2), (-7, 977, 3.3e-3), (2 ** 31 + 5, 2 ** 20, 0.0)):
        slots = np.full(nslots, -1, np.int64)
        rc = _ffi.lib.sgcn_step_fill(ctypes.byref(f), meta.ctypes.data, 64, ip, fp, seed, step, lr, slots.ctypes.data, nslots)
        assert rc == 0
        want = meta[idx] * mul + (base == 1) * ip + (base == 2) * fp
        assert np.array_equal(slots[:n], want)
        assert [int(slots[s]) for s in key_slot] == [ops.dropout_key(seed, int(li), step) for li in key_layer]
        assert int(slots[46]) == int(np.float32(lr).view(np.uint32)) and slots[44] == slots[45] == -1
    # a minibatch over one of the capacities: refused, the table untouched
    for k, arr in ((0, cap_max), (0, None)):
        slots = np.full(nslots, -1, np.int64)
        if arr is not None:
            arr[k] -= 1
        else:
            f.ws_floats -= 1
        assert _ffi.lib.sgcn_step_fill(ctypes.byref(f), meta.ctypes.data, 64, ip, fp, 1, 1, 0.1, slots.ctypes.data, nslots) == 1
        assert (slots == -1).all()
        if arr is not None:
            arr[k] += 1
        else:
            f.ws_floats += 1
    # an index outside the descriptor table is an error, not a read
    assert _ffi.lib.sgcn_step_fill(ctypes.byref(f), meta.ctypes.data, 32, ip, fp, 1, 1, 0.1, slots.ctypes.data, nslots) == -1
    assert b"descriptor table" in _ffi.lib.sgcn_last_error()
    assert _ffi.lib.sgcn_copy_h2d_async(None, None, -1, None) == -1 and _ffi.lib.sgcn_copy_h2d_async(None, None, 0, None) == 0
