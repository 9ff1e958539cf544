"""ctypes binding of libmrgcn_hip.so (the C ABI declared in include/mrgcn_hip.h).

There is NO fallback: if the library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  It must be the one
# resident in the process before libmrgcn_hip.so is loaded, so that the kernels, torch's
# allocator and torch's streams share ONE runtime (loading ours first pulls in the system
# runtime, which then sees no device next to torch's HSA stack).
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmrgcn_hip.so")

# enums of include/mrgcn_hip.h
OK = 0
VAL_I8, VAL_F32 = 0, 1
PLAN_PRUNE_ZEROS, PLAN_REPLICATE, PLAN_NO_REPLICATE, PLAN_LEAN = 1, 2, 4, 8
VIEW_LITERAL, VIEW_COMPACT, VIEW_TRANSPOSED = 0, 1, 2
ABI_VERSION = 5  # include/mrgcn_hip.h: MRGCN_ABI_VERSION
SPMM_RELU, SPMM_PAD_WRITABLE, SPMM_TWO_PASS = 1, 2, 4  # flag word of mrgcn_spmm_f32 / _bf16 (`relu` argument)
(ARR_ROWPTR, ARR_LCOL, ARR_CCOL, ARR_VAL, ARR_CPTR, ARR_CROW, ARR_CVAL, ARR_UREL, ARR_UNODE,
 ARR_NPTR, ARR_ROWIDX, ARR_ULCOL, ARR_RPERM, ARR_RELPTR, ARR_MPOS, ARR_MCOL, ARR_MVAL, ARR_ROWMAP,
 ARR_PTR3) = range(19)
FLOAT_ARRAYS = (ARR_VAL, ARR_CVAL, ARR_MVAL)


class MrgcnError(RuntimeError):
    pass


class PlanInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "num_rows", "num_nodes", "num_relations", "nnz", "ncols", "max_row_nnz", "max_col_nnz",
        "long_rows", "long_cols", "device_bytes", "operand_rows", "replicas")]


class SupportInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "live_cols", "live_entries", "live_nodes", "device_bytes", "chunks_wide", "chunks_narrow", "flagged_rows")]


(SUP_COL_FLAGS, SUP_NODE_FLAGS, SUP_LCOL, SUP_LREL, SUP_NLPTR, SUP_LPTR, SUP_LROW, SUP_LVAL, SUP_LNODE,
 SUP_LPERM, SUP_FROW, SUP_FPTR, SUP_FCOL, SUP_FVAL, SUP_LNODE_ORD, SUP_ROWRANK) = range(16)
SUPPORT_FORWARD = 1

_p = C.c_void_p
_i32, _i64, _u32 = C.c_int32, C.c_int64, C.c_uint32

# name -> (restype, argtypes); kept in one table so that tests can compare it with the header
SIGNATURES = {
    "mrgcn_abi_version": (C.c_int, []),
    "mrgcn_config_count": (_i32, []),
    "mrgcn_config_name": (C.c_char_p, [_i32]),
    "mrgcn_config_doc": (C.c_char_p, [_i32]),
    "mrgcn_config_get": (C.c_int, [C.c_char_p, C.POINTER(_i64)]),
    "mrgcn_config_set": (C.c_int, [C.c_char_p, _i64]),
    "mrgcn_arch": (C.c_char_p, []),
    "mrgcn_last_error": (C.c_char_p, []),
    "mrgcn_plan_create": (C.c_int, [C.POINTER(_p), _i64, _i64, _i32, _i64, _p, _p, _p, _i32, _u32, _p]),
    "mrgcn_plan_create_csr": (C.c_int, [C.POINTER(_p), _i64, _i64, _i32, _i64, _p, _p, _p, _i32, _u32, _p]),
    "mrgcn_plan_create_hinted": (C.c_int, [C.POINTER(_p), _i64, _i64, _i32, _i64, _p, _p, _p, _i32, _u32, _p, _i32, _p]),
    "mrgcn_plan_create_csr_hinted": (C.c_int, [C.POINTER(_p), _i64, _i64, _i32, _i64, _p, _p, _p, _i32, _u32, _p, _i32,
                                               _p]),
    "mrgcn_plan_destroy": (C.c_int, [_p]),
    "mrgcn_plan_destroy_after": (C.c_int, [_p, _p]),
    "mrgcn_plan_info": (C.c_int, [_p, C.POINTER(PlanInfo)]),
    "mrgcn_plan_export": (C.c_int, [_p, _i32, _p, _i64]),
    "mrgcn_plan_array": (C.c_int, [_p, _i32, C.POINTER(_p), C.POINTER(_i64)]),
    "mrgcn_spmm_f32": (C.c_int, [_p, _i32, _p, _i64, _i32, _p, _i64, _p, _i32, _p, _p]),
    "mrgcn_operand_replicate": (C.c_int, [_p, _p, _i64, _p]),
    "mrgcn_basis_mix_fwd_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _i64, _p, _i64, _p]),
    "mrgcn_gather_rows_f32": (C.c_int, [_p, _p, _i32, _p, _i64, _p, _i64, _p]),
    "mrgcn_rel_transform_fwd_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, _p, _i64, _i32, _p]),
    "mrgcn_spmm_bf16": (C.c_int, [_p, _i32, _p, _i64, _i32, _p, _i64, _p, _i32, _p, _p]),
    "mrgcn_basis_mix_fwd_bf16": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _i64, _p, _i64, _p]),
    "mrgcn_gather_rows_bf16": (C.c_int, [_p, _p, _i32, _p, _i64, _p, _i64, _p]),
    "mrgcn_rel_transform_fwd_bf16": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, _p, _i64, _i32, _p]),
    "mrgcn_cast_rows_bf16": (C.c_int, [_p, _i64, _i64, _i32, _p, _i64, _p]),
    "mrgcn_rel_transform_xbf16_supported": (_i32, [_p, _i32, _i32, _i64, _i64]),
    "mrgcn_rel_transform_fwd_xbf16": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, _p, _i64, _i32, _i32, _p]),
    "mrgcn_basis_mix_fwd_abf16": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _i64, _p, _i64, _i32, _p]),
    "mrgcn_support_rel_transform_bwd_xbf16": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _p, _i32, _p, _i64, _p, _p, _i64,
                                                        _p]),
    "mrgcn_rel_transform_bwd_workspace": (C.c_int64, [_p, _i32, _i32, _i32, _i32]),
    "mrgcn_basis_mix_bwd_f32": (C.c_int, [_p, _p, _i64, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p]),
    "mrgcn_adam_step_rows_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p, C.c_float, C.c_float, C.c_float,
                                           C.c_float, _i64, _p, _p, _p]),
    "mrgcn_adam_step_index_rows_f32": (C.c_int, [_p, _p, _i64, _p, _p, _p, _i64, _i32, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, _i64, _p, _p, _p]),
    "mrgcn_adam_rows_fused_supported": (_i32, [_p, _i32, _i32]),
    "mrgcn_adam_step_rows_fused_f32": (C.c_int, [_p, _p, _i64, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, C.c_float,
                                                 C.c_float, C.c_float, C.c_float, _i64, _p, _p, _p]),
    "mrgcn_rel_transform_bwd_f32": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _p, _i32, _p, _i64, _p, _p, _i64, _p]),
    "mrgcn_rel_transform_bwd_live_f32": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _i32, _p, _i32, _p, _i64, _p, _p,
                                                   _i64, _p]),
    "mrgcn_rows_nonzero_f32": (C.c_int, [_p, _i64, _i32, _i64, _p, _p]),
    "mrgcn_softmax_xent_rows_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _p, _p, _p]),
    "mrgcn_softmax_xent_bwd_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _i64, _i64, _p, _p]),
    "mrgcn_basis_contract_f32": (C.c_int, [_p, _p, _i32, _i32, _i64, _p, _p]),
    "mrgcn_basis_contract_bwd_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _i64, _p, _p, _p]),
    "mrgcn_sumsq_accum_multi_f32": (C.c_int, [_i32, _p, _p, _p, _p]),
    "mrgcn_sumsq_clip_multi_f32": (C.c_int, [_i32, _p, _p, _i32, _p, _p, _p, C.c_float, _p, _p, _p, _p, C.c_float,
                                             C.c_float, _p, _p]),
    "mrgcn_adam_step_multi_f32": (C.c_int, [_i32, _p, _p, _p, _p, _p, _p, _p, C.c_float, C.c_float, C.c_float, _i64,
                                            _p, _p, _p]),
    "mrgcn_spmm_transposed_live_scratch": (C.c_int64, [_p]),
    "mrgcn_spmm_transposed_live_flagged_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _p, _p, _p, _i32, _p, _p, _p]),
    "mrgcn_rel_transform_bwd_masked_supported": (C.c_int32, [_p, _i32, _i32, _i64]),
    "mrgcn_rel_transform_bwd_masked_f32": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _i32, _p, _i32, _p, _i64, _p, _p,
                                                     _i64, _i32, _p, _p, _p]),
    "mrgcn_spmm_transposed_live_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _p, _p, _p, _i32, _p]),
    "mrgcn_colsum_rows_workspace": (_i64, [_i32]),
    "mrgcn_colsum_rows_f32": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _p, _i64, _p]),
    "mrgcn_scatter_rows_zero_fill_f32": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _p, _i64, _p]),
    "mrgcn_probe_copy_f32": (C.c_int, [_p, _p, _i64, _p]),
    "mrgcn_probe_triad_f32": (C.c_int, [_p, _p, _p, _i64, _p]),
    "mrgcn_probe_triad_persistent_f32": (C.c_int, [_p, _p, _p, _i64, _p]),
    "mrgcn_probe_copy_persistent_f32": (C.c_int, [_p, _p, _i64, _p]),
    "mrgcn_relu_bwd_f32": (C.c_int, [_p, _p, _i64, _p, _p]),
    "mrgcn_relu_bwd_rows_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _i64, _p]),
    "mrgcn_softmax_xent_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _p, _p, _i64, _i64, _p]),
    "mrgcn_sumsq_accum_f32": (C.c_int, [_p, _i64, _p, _p]),
    "mrgcn_clip_coef_f32": (C.c_int, [_p, C.c_float, _p, _p, _p]),
    "mrgcn_adam_step_f32": (C.c_int, [_p, _p, _p, _p, _i64, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_float, _i64, _p, _p]),
    "mrgcn_distmult_score_f32": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p, _p]),
    "mrgcn_distmult_score_bwd_f32": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _i64, _p]),
    "mrgcn_distmult_score_bwd_sorted_f32": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p, _p, _p, _p, _p, _i64,
                                                      _p, _i64, _p]),
    "mrgcn_bce_logits_f32": (C.c_int, [_p, _p, _i64, _p, _p, _p]),
    "mrgcn_random_subset_i64": (C.c_int, [_i64, _i64, _p, _p, _p]),
    "mrgcn_corrupt_triples_i64": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _p, _p]),
    "mrgcn_distmult_orders_workspace": (C.c_int64, [_i64]),
    "mrgcn_distmult_orders": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _p]),
    "mrgcn_distmult_orders_counting_workspace": (C.c_int64, [_i64, _i64]),
    "mrgcn_distmult_orders_counting": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _p]),
    "mrgcn_distmult_ranks_workspace": (C.c_int64, [_i64, _i32, _i64]),
    "mrgcn_distmult_ranks": (C.c_int, [_p, _i64, _i64, _p, _i64, _i32, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "mrgcn_adam_bias_f32": (C.c_int, [_p, C.c_float, C.c_float, _p, _p]),
    "mrgcn_adam_step_dev_f32": (C.c_int, [_p, _p, _p, _p, _i64, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, _p, _p, _p]),
    "mrgcn_mlp_fused_supported": (C.c_int32, [_i32, _p]),
    "mrgcn_mlp_gate_scatter_fwd_f32": (C.c_int, [_i32, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i32, _p]),
    "mrgcn_mlp_gate_scatter_bwd_f32": (C.c_int, [_i32, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i32, _p, _p, _p,
                                                 _p]),
    "mrgcn_gemm_f32": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i32, _p,
                                 C.c_float, _p, _p]),
    "mrgcn_gemm_bf16mm_f32": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i32, _p,
                                 C.c_float, _p, _p]),
    "mrgcn_colsum_f32": (C.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "mrgcn_channel_sum_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "mrgcn_bn_running_stats_f32": (C.c_int, [_p, _p, _i32, _i64, C.c_float, _p, _p, _p]),
    "mrgcn_pool_out_len": (_i32, [_i32, _i32, _i32]),
    "mrgcn_bn_workspace_bytes": (C.c_size_t, [_i32]),
    "mrgcn_bn_relu_pool_fwd_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, C.c_float, _i32, _p, _p, _i32, _i32, _p, _p,
                                             _p, _p]),
    "mrgcn_bn_relu_pool_bwd_f32": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, C.c_float, _i32, _i32, _i32,
                                             _p, _p, _p, _p, _p, _p]),
    "mrgcn_bn_relu_pool_bwd_sum_f32": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, C.c_float, _i32, _i32,
                                                 _i32, _p, _p, _p, _p, _p, _p, _p]),
    "mrgcn_frontier_workspace_bytes": (C.c_size_t, [_i64, _i64]),
    "mrgcn_frontier_count": (C.c_int, [_p, _p, _i64, _p, _i64, _p, _p, _p, C.c_size_t, _p]),
    "mrgcn_frontier_emit": (C.c_int, [_p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "mrgcn_plan_entry_relations": (C.c_int, [_p, _p, _p]),
    "mrgcn_wide_input_bwd_supported": (_i32, [_p, _i32, _i32]),
    "mrgcn_wide_input_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i32, _i32, _p, _p, _p]),
    "mrgcn_wide_input_bwd_det_workspace": (_i64, [_p, _i64, _i32, _i32]),
    "mrgcn_wide_input_bwd_det_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _i64, _p, _p, _i32,
                                               _i32, _p, _p, _p, _i64, _p]),
    "mrgcn_support_create": (C.c_int, [C.POINTER(_p), _p, _p, _p]),
    "mrgcn_support_create_ex": (C.c_int, [C.POINTER(_p), _p, _p, _u32, _p]),
    "mrgcn_support_create_chain": (C.c_int, [C.POINTER(_p), _i32, _p, _p, _u32, _p]),
    "mrgcn_support_destroy": (C.c_int, [_p]),
    "mrgcn_support_destroy_ordered": (C.c_int, [_p]),
    "mrgcn_support_spmm_fwd_f32": (C.c_int, [_p, _i32, _p, _i64, _i32, _p, _i64, _p, _i32, _p]),
    "mrgcn_support_spmm_t_compact_f32": (C.c_int, [_p, _i32, _p, _i64, _i32, _p, _i64, _p]),
    "mrgcn_support_mix_fwd_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _i64, _p]),
    "mrgcn_support_literal_rows_f32": (C.c_int, [_p, _i32, _p, _i32, _p, _i64, _p]),
    "mrgcn_support_rel_transform_supported": (_i32, [_p, _i32, _i32, _i32]),
    "mrgcn_support_rel_transform_fwd_f32": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _i32, _p, _i64, _p]),
    "mrgcn_support_rel_transform_bwd_compact_f32": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _i32, _p, _i32, _p, _i64, _p,
                                                              _p, _i64, _i32, _p]),
    "mrgcn_support_info": (C.c_int, [_p, C.POINTER(SupportInfo)]),
    "mrgcn_support_array": (C.c_int, [_p, _i32, C.POINTER(_p), C.POINTER(_i64)]),
    "mrgcn_support_spmm_t_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _p]),
    "mrgcn_support_mix_bwd_workspace": (C.c_int64, [_p, _i32]),
    "mrgcn_support_mix_bwd_f32": (C.c_int, [_p, _p, _i64, _p, _p, _i32, _i32, _p, _i32, _p, _p, _p, _i64, _p]),
    "mrgcn_support_adam_rows_fused_f32": (C.c_int, [_p, _p, _i64, _p, _i32, _i32, _p, _p, _p, _p, C.c_float, C.c_float,
                                                    C.c_float, C.c_float, _i64, _p, _p, _i32, _p]),
    "mrgcn_support_rel_transform_bwd_workspace": (C.c_int64, [_p, _i32, _i32, _i32, _i32]),
    "mrgcn_support_rel_transform_bwd_f32": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _p, _i32, _p, _i64, _p, _p, _i64,
                                                      _i32, _p]),
    "mrgcn_softmax_xent_bwd_rows_f32": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _i64, _p]),
    "mrgcn_event_create": (C.c_int, [C.POINTER(_p)]),
    "mrgcn_event_destroy": (C.c_int, [_p]),
    "mrgcn_event_record": (C.c_int, [_p, _p]),
    "mrgcn_event_elapsed_ms": (C.c_int, [_p, _p, C.POINTER(C.c_float)]),
}

_lib = None


def load():
    """Loads the shared library (once).  Raises MrgcnError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MrgcnError(
            f"{LIB_PATH} is missing: build it with `python -m mrgcn_amd.build` "
            "(mrgcn_amd has no CPU or PyTorch fallback for its HIP kernels)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mrgcn_abi_version() != ABI_VERSION:
        raise MrgcnError("libmrgcn_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != OK:
        msg = load().mrgcn_last_error().decode(errors="replace")
        raise MrgcnError(f"{what or 'mrgcn call'} failed (code {rc}): {msg}")


def config() -> dict:
    """The library's configuration table as {name: value} (include/mrgcn_hip.h: mrgcn_config_*)."""
    lib = load()
    out = {}
    for i in range(lib.mrgcn_config_count()):
        name = lib.mrgcn_config_name(i)
        v = _i64()
        check(lib.mrgcn_config_get(name, C.byref(v)), "mrgcn_config_get")
        out[name.decode()] = int(v.value)
    return out


def set_config(**kw) -> dict:
    """Sets library switches (`set_config(adam_list=0)`); returns the previous values.  Takes effect at the next call of
    the entry points concerned; plans keep what they were built with."""
    lib = load()
    prev = {}
    for k, v in kw.items():
        old = _i64()
        check(lib.mrgcn_config_get(k.encode(), C.byref(old)), "mrgcn_config_get")
        check(lib.mrgcn_config_set(k.encode(), int(v)), "mrgcn_config_set")
        prev[k] = int(old.value)
    return prev
