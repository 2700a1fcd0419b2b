"""``history.slice`` / ``history.dense_slice`` -- drop-in for the reference's Cython module
(gcn/_history.pyx:25-62) on device memory.

``dense_slice(a, r)`` gathers rows of an HBM-resident matrix; ``slice(a, r)`` row-slices an
HBM-resident CSR and returns the reference's ``(indices[nnz,2] int32, data f32,
dense_shape int32[2])`` triple as device tensors (or an empty DeviceCSR when nnz == 0, the
counterpart of the reference's empty ``csr_matrix`` edge case, gcn/_history.pyx:34-35).
"""
import numpy as np
import torch

from . import ops


def dense_slice(a, r):
    if not isinstance(r, torch.Tensor):
        r = torch.from_numpy(np.ascontiguousarray(r, dtype=np.int32)).to(a.device)
    return ops.gather_rows(a, r)


def slice(a, r):  # noqa: A001  (reference name)
    r_host = r.cpu().numpy() if isinstance(r, torch.Tensor) else np.ascontiguousarray(r, dtype=np.int32)
    s = ops.csr_slice(a, r_host, with_coo_rows=True)
    if s.nnz == 0:
        return s
    indices = torch.stack([s.coo_rows, s.col], dim=1)
    shape = torch.tensor([s.shape[0], s.shape[1]], dtype=torch.int32)
    return indices, s.val, shape
