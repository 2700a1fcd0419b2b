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
    assert _ffi.lib.sgcn_abi_version() == _ffi.ABI_VERSION == 7
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
