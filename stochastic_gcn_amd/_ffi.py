"""ctypes binding of libsgcn.so (C-ABI declared in include/sgcn.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C stochastic_gcn_amd/csrc``).
There is NO fallback: if the library is missing or a symbol declared in the header is not
exported, importing this module raises -- the product path never silently degrades to a
CPU / PyTorch implementation.
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be the one and
# only HIP runtime in the process: tensors, streams and our kernels have to share a context.
# Importing torch first makes the loader resolve libsgcn.so's libamdhip64 dependency to the
# already-loaded copy instead of pulling /opt/rocm's second runtime in.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsgcn.so")

ABI_VERSION = 16         # include/sgcn.h sgcn_abi_version(): bumped on any signature change

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)


class Seg(C.Structure):
    _fields_ = [("row", C.c_int32), ("start", C.c_int32), ("end", C.c_int32), ("slot", C.c_int32)]


class Fix(C.Structure):
    _fields_ = [("row", C.c_int32), ("first_slot", C.c_int32), ("nslots", C.c_int32)]


class CsPlan(C.Structure):
    _fields_ = [("R", C.c_int32), ("ntiles", C.c_int64), ("dev_tile_ptr", C.c_void_p),
                ("dev_colrow", C.c_void_p), ("dev_val", C.c_void_p), ("dev_tile_rows", C.c_void_p),
                ("dev_tile_slots", C.c_void_p), ("dev_fix", C.c_void_p), ("nfix", C.c_int64),
                ("nslots", C.c_int64), ("dev_ws", C.c_void_p), ("ws_elems", C.c_int64),
                ("round_tiles", C.c_int64), ("host_tile_nnz_hint", C.c_void_p),
                ("pace_ns_per_nnz", C.c_int32), ("G", C.c_int32), ("xcd_map", C.c_int32),
                ("dev_warp", C.c_void_p), ("warp_shift", C.c_int32)]


class LdsPlan(C.Structure):
    """include/sgcn.h sgcn_ldsplan_t"""
    _fields_ = [("VW", C.c_int32), ("NW", C.c_int32), ("RW", C.c_int32), ("S", C.c_int32), ("U", C.c_int32),
                ("nparts", C.c_int32), ("unit", C.c_int32), ("xcd_tile_ptr", C.c_int32 * 9), ("ntiles", C.c_int64), ("nchunks", C.c_int64), ("nent", C.c_int64),
                ("dev_tile_chunk_ptr", C.c_void_p), ("dev_chunk_cols", C.c_void_p), ("dev_chunk_hdr", C.c_void_p),
                ("dev_ent_ptr", C.c_void_p), ("dev_words", C.c_void_p), ("dev_vals", C.c_void_p), ("dev_row_fold", C.c_void_p),
                ("dev_tile_rows", C.c_void_p), ("dev_tile_slots", C.c_void_p),
                ("dev_fix", C.c_void_p), ("nfix", C.c_int64), ("nslots", C.c_int64), ("dev_ws", C.c_void_p),
                ("ws_elems", C.c_int64)]


STEP_MAX_ARGS = 48


class StepOp(C.Structure):
    """include/sgcn.h sgcn_step_op_t"""
    _fields_ = [("op", C.c_int32), ("nargs", C.c_int32), ("mul", C.c_int64 * STEP_MAX_ARGS),
                ("slot", C.c_int32 * STEP_MAX_ARGS), ("add", C.c_int64 * STEP_MAX_ARGS)]


class StepFill(C.Structure):
    """include/sgcn.h sgcn_step_fill_t"""
    _fields_ = [("n", C.c_int64), ("idx", C.c_void_p), ("mul", C.c_void_p), ("base", C.c_void_p),
                ("n_cap", C.c_int64), ("cap_idx", C.c_void_p), ("cap_max", C.c_void_p),
                ("n_ws", C.c_int64), ("ws_idx", C.c_void_p), ("ws_ld", C.c_void_p), ("ws_floats", C.c_int64),
                ("n_keys", C.c_int64), ("key_slot", C.c_void_p), ("key_layer", C.c_void_p), ("lr_slot", C.c_int64)]


class Dropout(C.Structure):
    """include/sgcn.h sgcn_dropout_t"""
    _fields_ = [("key", C.c_uint32), ("keep", C.c_float), ("rows", C.c_int32), ("width", C.c_int32)]


class Plan(C.Structure):
    _fields_ = [("dev_seg", C.c_void_p), ("nseg", C.c_int64),
                ("dev_fix", C.c_void_p), ("nfix", C.c_int64),
                ("nslots", C.c_int64), ("dev_ws", C.c_void_p), ("ws_elems", C.c_int64)]


P = C.c_void_p  # raw address (host or device), passed as integers from data_ptr()/ctypes.data

# name -> (restype, argtypes).  Must list every symbol of include/sgcn.h
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "sgcn_last_error": (C.c_char_p, []),
    "sgcn_abi_version": (C.c_int, []),
    "sgcn_plan_count": (C.c_int, [P, C.c_int32, C.c_int32, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sgcn_plan_fill": (C.c_int, [P, C.c_int32, C.c_int32, P, P]),
    "sgcn_spmm_csr_f32": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P,
                                    P, P, P, C.c_int64, C.c_float, C.POINTER(Plan), P]),
    "sgcn_spmm_csr_add_f32": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P,
                                    P, P, P, C.c_int64, C.c_float, C.POINTER(Plan), P, C.c_int64, C.c_int32, P]),
    "sgcn_tune": (C.c_int, [C.c_char_p, C.c_int64]),
    "sgcn_tune_get": (C.c_int64, [C.c_char_p]),
    "sgcn_csplan_count": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, P, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sgcn_csplan_fill": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, P, P, P, P, P, P, P]),
    "sgcn_csplang_count": (C.c_int, [P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, C.c_int32,
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sgcn_csplang_fill": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, C.c_int32,
                                    P, P, P, P, P, P]),
    "sgcn_csplan_build": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P, C.c_int32,
                                    C.c_int32, C.POINTER(P)]),
    "sgcn_csbuild_sizes": (C.c_int, [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sgcn_csbuild_export": (C.c_int, [P, P, P, P, P, P, P]),
    "sgcn_csbuild_free": (None, [P]),
    "sgcn_cs_warp_table": (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, P,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sgcn_csr_transpose_host": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, P, P, P]),
    "sgcn_host_threads": (C.c_int32, []),
    "sgcn_reorder_lp": (C.c_int, [P, P, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, P, C.POINTER(C.c_int32)]),
    "sgcn_spmm_cs_variant": (C.c_int, [C.POINTER(CsPlan), C.c_int32, C.c_char_p, C.c_int32]),
    "sgcn_spmm_cs_f32": (C.c_int, [C.POINTER(CsPlan), C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P,
                                   P, P, P, C.c_int64, C.c_float, P]),
    "sgcn_ldsplan_create": (C.c_int, [P, P, P, C.c_int32, C.c_int32, P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.POINTER(C.c_void_p)]),
    "sgcn_ldsplan_sizes": (C.c_int, [C.c_void_p, P]),
    "sgcn_ldsplan_export": (C.c_int, [C.c_void_p, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "sgcn_ldsplan_destroy": (None, [C.c_void_p]),
    "sgcn_lds_profile_buffer": (C.c_int, [P]),
    "sgcn_spmm_lds_f32": (C.c_int, [C.POINTER(LdsPlan), C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P, P,
                                    C.c_int64, C.c_float, P]),
    "sgcn_vr_aggregate_f32": (C.c_int, [P, P, P, P, P, P, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, P, P, C.c_int64, P, C.c_int64, P, P, P, P, P,
                                        C.c_int64, C.c_int32, C.c_int32, C.POINTER(Plan), P]),
    "sgcn_vr_aggregate_pre_f32": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P, P,
                                            C.POINTER(Plan), P]),
    "sgcn_vr_aggregate_post_f32": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, P, P, C.c_int64, P, C.c_int64,
                                             P, P, P, P, C.c_int64, C.c_int32, C.c_int32, P, P]),
    "sgcn_gather_rows_f32": (C.c_int, [P, C.c_int64, P, C.c_int32, C.c_int32, P, C.c_int64, P]),
    "sgcn_scatter_rows_f32": (C.c_int, [P, C.c_int64, P, C.c_int32, C.c_int32, P, C.c_int64, P]),
    "sgcn_coll_available": (C.c_int, [C.POINTER(C.c_int32)]),
    "sgcn_coll_retain": (C.c_int, []),
    "sgcn_coll_abort": (C.c_int, []),
    "sgcn_coll_async_error": (C.c_int, []),
    "sgcn_coll_unique_id": (C.c_int, [P]),
    "sgcn_coll_init": (C.c_int, [P, C.c_int32, C.c_int32]),
    "sgcn_coll_world": (C.c_int, []),
    "sgcn_coll_destroy": (C.c_int, []),
    "sgcn_coll_allreduce_avg_f32": (C.c_int, [P, C.c_int64, P]),
    "sgcn_coll_allgather_i32": (C.c_int, [P, P, C.c_int64, P]),
    "sgcn_coll_init_exchange": (C.c_int, [P]),
    "sgcn_coll_has_exchange": (C.c_int, []),
    "sgcn_coll_allgather_x_i32": (C.c_int, [P, P, C.c_int64, P]),
    "sgcn_hist_pack_f32": (C.c_int, [P, C.c_int32, P, C.c_int64, C.c_int32, C.c_int32, P, P]),
    "sgcn_hist_apply_f32": (C.c_int, [P, C.c_int64, P, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "sgcn_csr_slice_indptr": (C.c_int, [C.c_int32, P, P, P]),
    "sgcn_csr_slice_indptr_dev": (C.c_int, [C.c_int32, P, P, P, P]),
    "sgcn_scale_rows_f32": (C.c_int, [P, C.c_int64, P, C.c_int32, C.c_int32, P, C.c_int64, P]),
    "sgcn_csr_slice_f32": (C.c_int, [C.c_int32, P, P, P, P, P, P, P, P, P]),
    "sgcn_ln_act_fwd_f32": (C.c_int, [P, C.c_int64, P, P, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                      P, C.c_int64, P, P, P]),
    "sgcn_ln_act_bwd_ws_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "sgcn_ln_act_bwd_f32": (C.c_int, [P, C.c_int64, P, C.c_int64, P, P, P, C.c_int32, C.c_int32,
                                      C.c_int32, P, C.c_int64, P, P, P, P]),
    "sgcn_gemm_ws_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "sgcn_gemm_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P,
                                C.c_int64, P, C.c_int64, C.c_int32, P, P, P, P]),
    "sgcn_dense_fwd_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P, C.c_int64, C.c_int32,
                                     P, C.c_int64, P, P, C.c_float, C.c_int32, P, C.c_int64, P, P, P, P, P, P, P]),
    "sgcn_dense_bwd_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, P, C.c_int64, P, C.c_int64, P, P, P,
                                     C.c_int32, P, C.c_int64, P, C.c_int64, P, C.c_int64, P, P, P, C.c_int64,
                                     P, P, P, P, P]),
    "sgcn_det_pre_f32": (C.c_int, [P, P, C.c_int64, C.c_float, P, P]),
    "sgcn_det_pre_bwd_f32": (C.c_int, [P, P, C.c_int64, C.c_float, P, P, P]),
    "sgcn_square_f32": (C.c_int, [P, C.c_int64, C.c_float, P, P]),
    "sgcn_addmul_f32": (C.c_int, [P, P, P, C.c_int64, C.c_float, P]),
    "sgcn_det_lnvar_fwd_f32": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_float, P, P]),
    "sgcn_det_lnvar_bwd_f32": (C.c_int, [P, P, P, P, P, C.c_int32, C.c_int32, C.c_float, P, P, P, P, P]),
    "sgcn_det_relu_fwd_f32": (C.c_int, [P, P, C.c_int64, P, P, P]),
    "sgcn_det_relu_bwd_f32": (C.c_int, [P, P, P, P, C.c_int64, P, P, P]),
    "sgcn_gauss_sample_f32": (C.c_int, [P, P, C.c_int64, C.c_uint32, P, P]),
    "sgcn_gauss_sample_bwd_f32": (C.c_int, [P, P, C.c_int64, C.c_uint32, P, P]),
    "sgcn_det_agg_prep_f32": (C.c_int, [P, P, P, P, C.c_int64, P, C.c_int32, C.c_int32, P, P, P, P, P, P]),
    "sgcn_det_agg_prep_bwd_f32": (C.c_int, [P, P, P, P, P, C.c_int32, C.c_int32, P, C.c_int64, C.c_int32, P, P]),
    "sgcn_relu_eps_f32": (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, C.c_float, P, C.c_int64, P]),
    "sgcn_gate_f32": (C.c_int, [P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int32, P, P]),
    "sgcn_dropout_f32": (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, P, P, C.c_int64, P]),
    "sgcn_softmax_ce_f32": (C.c_int, [P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int32, P, C.c_int64,
                                      P, C.c_int64, P, P, P]),
    "sgcn_sigmoid_ce_f32": (C.c_int, [P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int32, P, C.c_int64,
                                      P, C.c_int64, P, P, P]),
    "sgcn_l2_penalty_f32": (C.c_int, [P, C.c_int64, C.c_int64, C.c_float, P, P, P]),
    "sgcn_csr_transpose_ws_ints": (C.c_int64, [C.c_int32, C.c_int64]),
    "sgcn_csr_transpose_index": (C.c_int, [C.c_int32, C.c_int64, P, P, P, P, P, P, P]),
    "sgcn_gather_f32": (C.c_int, [P, P, C.c_int64, P, P]),
    "sgcn_step_run": (C.c_int, [C.POINTER(StepOp), C.c_int32, P, C.c_int32, P]),
    "sgcn_step_fill": (C.c_int, [C.POINTER(StepFill), P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float,
                                 P, C.c_int64]),
    "sgcn_copy_h2d_async": (C.c_int, [P, P, C.c_int64, P]),
    "sgcn_adam_f32": (C.c_int, [P, P, P, P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, P]),
    "sgcn_sched_create": (C.c_int, [P, P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.POINTER(C.c_void_p)]),
    "sgcn_sched_destroy": (None, [C.c_void_p]),
    "sgcn_sched_seed": (C.c_int, [C.c_void_p, C.c_int32]),
    "sgcn_sched_start_batch": (C.c_int, [C.c_void_p, C.c_int32, P]),
    "sgcn_sched_expand": (C.c_int, [C.c_void_p, C.c_int32]),
    "sgcn_sched_view_i32": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(c_i32p), C.POINTER(C.c_int64)]),
    "sgcn_sched_view_f32": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(c_f32p), C.POINTER(C.c_int64)]),
    "sgcn_sched_batch_packed": (C.c_int, [C.c_void_p, C.c_int32, P, C.c_int32, P, P, C.c_int32,
                                          C.c_int32, P, C.c_int64, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64)]),
    "sgcn_sched_batch_packed_into": (C.c_int, [C.c_void_p, C.c_int32, P, C.c_int32, P, P, C.c_int32,
                                               C.c_int32, P, C.c_int64, P, C.c_int64,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sgcn_sched_packed_meta_len": (C.c_int64, [C.c_int32]),
    "sgcn_prefetch_start": (C.c_int, [P, C.c_int32, C.c_int32, P, P, C.c_int32, P, P, C.c_int32, C.c_int32,
                                      C.c_int32, P, P, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "sgcn_prefetch_next": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), P, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]),
    "sgcn_prefetch_release": (C.c_int, [C.c_void_p, C.c_int32]),
    "sgcn_prefetch_stats": (C.c_int, [C.c_void_p, P]),
    "sgcn_prefetch_stop": (None, [C.c_void_p]),
    "sgcn_sched_packed_copy": (C.c_int, [C.c_void_p, P, P]),
    "sgcn_mult_create": (C.c_int, [P, C.c_int32, C.POINTER(C.c_void_p)]),
    "sgcn_mult_destroy": (None, [C.c_void_p]),
    "sgcn_mult_tree": (C.c_int, [C.c_void_p, C.POINTER(c_f32p), C.POINTER(C.c_int64)]),
    "sgcn_mult_query_u": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(C.c_int32)]),
    "sgcn_mult_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
}


class SgcnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsgcn error %d: %s" % (code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "stochastic_gcn_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C stochastic_gcn_amd/csrc`). There is no CPU/PyTorch fallback."
            % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("libsgcn.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.sgcn_abi_version() != ABI_VERSION:      # same symbols, different signatures: never call into it
        raise ImportError("libsgcn.so has ABI version %d, this binding needs %d (stale build: run "
                          "`python -c 'import __graft_entry__ as g; g.build()'`)"
                          % (lib.sgcn_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def check(code):
    """Raise SgcnError on a non-zero status."""
    if code != 0:
        raise SgcnError(code, (lib.sgcn_last_error() or b"").decode("utf-8", "replace"))
    return code


def tune(key, value):
    check(lib.sgcn_tune(key.encode(), int(value)))
