"""
ctypes binding of libtensorrec_hip.so (include/tensorrec_hip.h).

PyTorch is used for device memory and streams only: every entry point takes raw device pointers
(``tensor.data_ptr()``) and the current HIP stream.  There is NO fallback: if the library is missing, fails to
load, or a tensor is not on a GPU, the call raises -- the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime that the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TREC_HIP_LIB: another build of the same library (A/B runs of kernel variants on one box); default: the in-tree build
LIB_PATH = os.environ.get("TREC_HIP_LIB") or os.path.join(_HERE, "libtensorrec_hip.so")

_vp, _i32, _i64, _u32, _u64, _f = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32,
                                   ctypes.c_uint64, ctypes.c_float)

_RETURNS_I64 = ("trec_csr_split_workspace_bytes", "trec_rank_rows_workspace_bytes", "trec_user_prep_alloc_rows",
                "trec_user_prep_workspace_bytes", "trec_group_pairs_staged_bytes", "trec_group_pairs_binned_bytes",
                "trec_fit_step_coop_workspace_floats")      # sizing queries that return a byte count

# name -> argtypes, in the order of include/tensorrec_hip.h
SIGNATURES = {
    "trec_abi_version": [],
    "trec_device_cu_count": [],
    "trec_set_tuning": [ctypes.c_char_p, _i32],
    "trec_get_tuning": [ctypes.c_char_p, ctypes.c_int],
    "trec_spmm_csr": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp],
    "trec_spmm_one_per_row": [_vp, _vp, _i64, _vp, _i32, _vp, _vp],
    "trec_spmm_csr_filter": [_vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "trec_absmax": [_vp, _i64, _vp, _vp],
    "trec_spmm_csr_packed": [_vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "trec_spmv_csr": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "trec_csr_split_workspace_bytes": [_i64, _i32],
    "trec_spmm_csr_split": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _vp],
    "trec_spmv_csr_split": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp],
    "trec_pair_euclid_coef": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "trec_csr_to_dense": [_vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "trec_row_l2norm_fwd": [_vp, _i64, _i32, _vp, _vp, _vp],
    "trec_row_l2norm_bwd": [_vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "trec_relu_bwd": [_vp, _vp, _i64, _vp, _vp],
    "trec_colsum": [_vp, _i64, _i32, _vp, _vp, _i32, _vp],
    "trec_gemm_f32": [_i32, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp],
    "trec_gemm_f32_split_bf16": [_i32, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp, _vp],
    "trec_score_kpad": [_i32],
    "trec_score_rows_per_workgroup": [_i32, _i32],
    "trec_score_tile_rows": [_i32, _i32],
    "trec_score_topk_capacity": [_i32],
    "trec_score_prep": [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "trec_score_gemm_store": [_vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp],
    "trec_score_topk_parts": [_i32, _i32, _i64, _i32],
    "trec_score_gemm_topk": [_vp, _vp, _i32, _i32, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp,
                             _i32, _vp],
    "trec_score_gemm_blockmax": [_vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _i32,
                                 _vp],
    "trec_topk_select_blocks": [_vp, _i32, _i64, _i64, _i32, _vp, _vp, _vp, _vp],
    "trec_topk_select_blocks_ex": [_vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp],
    "trec_score_prep_filter": [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_score_prep_i8": [_vp, _i64, _i32, _i32, _i32, _f, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_score_row_scale_i8": [_vp, _i64, _i32, _vp, _vp, _vp],
    "trec_score_prep_i8_users": [_vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp],
    "trec_score_bias_i8_classes": [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "trec_score_blockmax_i8_rows_per_workgroup": [_i32],
    "trec_score_user_err_i8": [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _i32, _vp, _vp],
    "trec_score_gemm_blockmax_i8": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _i32,
                                    _vp, _vp, _i32, _vp],
    "trec_user_prep_alloc_rows": [_i64, _i32],
    "trec_user_prep_workspace_bytes": [_i64],
    "trec_user_prep_sorted": [_vp, _i64, _i32, _i32, _i32, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_fill_zero": [_vp, _i64, _vp],
    "trec_topk_cascade_floor": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_rows_user_blocks": [_i64],
    "trec_topk_rows_count": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "trec_topk_rows_fill": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "trec_topk_prerefine_max_superblocks": [],
    "trec_topk_prerefine_rows": [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_prerefine_tau": [_vp, _vp, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp],
    "trec_topk_prerefine_rows_pos": [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_prerefine_tau_listed": [_vp, _vp, _vp, _i32, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "trec_score_gemm_refine_candidates_marked": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp,
                                                 _i32, _i32, _vp, _i32, _vp, _vp],
    "trec_topk_rows_collect": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "trec_score_gemm_blockmax_grouped": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp],
    "trec_topk_rows_hot": [_vp, _i32, _i32, _i64, _vp, _i32, _i64, _vp, _vp],
    "trec_score_gemm_blockmax_hot": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _i32, _vp, _i64, _vp],
    "trec_topk_dense_users": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "trec_score_gemm_refine_candidates": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp,
                                          _i32, _i32, _vp, _i32, _vp],
    "trec_topk_rows_wg_map": [_vp, _i32, _i32, _vp, _vp, _i64, _vp],
    "trec_topk_rows_wg_map_ex": [_vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp],
    "trec_score_gemm_refine_candidates_resident": [_vp, _vp, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp,
                                                   _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "trec_score_gemm_refine_candidates_hot": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _vp,
                                              _i32, _i32, _vp],
    "trec_topk_candidates_finish": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _i64, _i32, _vp,
                                    _vp, _vp, _vp, _vp, _i32, _vp],
    "trec_topk_candidates_finish_mixed": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _i64, _i32,
                                          _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "trec_topk_candidates_finish_wide": [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp,
                                         _vp, _vp],
    "trec_topk_euclid_certify": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_filter_floor": [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp],
    "trec_topk_collect_blocks": [_vp, _i32, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_filter_floor_ex": [_vp, _vp, _vp, _vp, _i32, _i64, _f, _vp, _vp, _vp, _vp],
    "trec_topk_scan_blocks": [_vp, _i32, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_prune_candidates": [_vp, _vp, _vp, _i32, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_collect_blocks_masked": [_vp, _i32, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_topk_filter_finish": [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _i64, _i32, _vp, _vp,
                                _vp, _vp, _vp],
    "trec_topk_filter_finish_wide": [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _i64, _i32, _vp, _vp,
                                     _vp, _vp, _vp],
    "trec_topk_group_keys": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "trec_topk_pad_counts": [_vp, _i32, _i32, _vp, _vp],
    "trec_exclusive_scan_i32": [_vp, _i64, _vp, _vp, _vp],
    "trec_topk_fill_groups": [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp],
    "trec_score_gemm_topk_grouped": [_vp, _vp, _i32, _i32, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp,
                                     _vp, _i32, _vp, _vp, _i32, _vp, _vp],
    "trec_topk_fill_groups_index": [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp],
    "trec_topk_merge": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp],
    "trec_pair_score_fwd": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "trec_pair_score_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "trec_group_pairs_by_item": [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "trec_group_pairs_lds_runs": [_i64, _i64],
    "trec_group_pairs_by_item_lds": [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "trec_group_pairs_staged_bytes": [_i64, _i32],
    "trec_group_pairs_binned_bytes": [_i64, _i64],
    "trec_group_pairs_by_item_binned": [_vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "trec_group_pairs_by_item_staged": [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "trec_rank_rows": [_vp, _i64, _i64, _i64, _vp, _i64, _vp],
    "trec_rank_rows_workspace_bytes": [_i64, _i64],
    "trec_rank_rows_chunked": [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp],
    "trec_score_rankcount_max_targets": [],
    "trec_pair_score_exact": [_vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp],
    "trec_score_gemm_rankcount": [_vp, _vp, _i32, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _i32, _vp, _vp],
    "trec_rank_of_pairs": [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "trec_rank_of_pairs_by_user": [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp],
    "trec_wmrb_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp],
    "trec_wmrb_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp],
    "trec_wmrb_fused_lds_bytes": [_i32, _i32, _i32],
    "trec_wmrb_fused_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                             _vp, _vp, _vp, _vp, _vp],
    "trec_wmrb_tiled_lds_bytes": [_i32, _i32, _i32],
    "trec_wmrb_tiled_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp,
                             _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "trec_fit_step_coop_workspace_floats": [_i64, _i64, _i32, _i32, _i32],
    "trec_fit_step_coop": [_vp] * 12 + [_vp] * 6 + [_vp] * 5 + [_i64, _i64, _i64, _i32, _i32, _i32, _i64, _u64, _u32, _f, _f, _f, _f, _f,
                                                            _vp, _i64, _vp, _vp, _vp],
    "trec_item_weighted_hist": [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _vp],
    "trec_dense_loss_fwd": [_i32, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "trec_dense_loss_fwd_phase": [_i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "trec_dense_loss_bwd": [_i32, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "trec_gram_f64": [_vp, _i64, _i32, _i64, _vp, _vp],
    "trec_dense_loss_factored_phase": [_i32, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "trec_dense_loss_factored_bwd": [_i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "trec_rmse_fwd": [_vp, _vp, _i64, _vp, _i32, _vp, _vp],
    "trec_rmse_bwd": [_vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "trec_sample_items": [_i64, _i64, _i32, _i32, _i32, _u64, _u32, _vp, _vp],
    "trec_crc32c": [_vp, _u64, _u32],
    "trec_collapse_tastes_fwd": [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "trec_collapse_tastes_bwd": [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp],
    "trec_adam_tf_step": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _vp],
    "trec_adam_schedule_advance": [_vp, _f, _f, _f, _i32, _vp],
    "trec_adam_tf_step_dev": [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _vp],
    "trec_sample_items_dev": [_i64, _i64, _i32, _i32, _i32, _u64, _vp, _vp, _vp],
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises NativeLibraryError if the .so is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C tensorrec_amd/csrc`.  tensorrec_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, exc))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise NativeLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name in _RETURNS_I64 else ctypes.c_int
    lib.trec_last_error.argtypes = []
    lib.trec_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def require_gpu():
    if not torch.cuda.is_available():
        raise NativeLibraryError("tensorrec_amd needs an MI355X (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors and non-contiguous views."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeLibraryError("expected a GPU tensor, got device=%s" % t.device)
    if not t.is_contiguous():
        raise NativeLibraryError("expected a contiguous tensor")
    if t.device.index != torch.cuda.current_device():
        raise NativeLibraryError("tensor lives on %s but the current device (whose stream the kernels are launched on) "
                                 "is cuda:%d -- call through the model's public methods or use torch.cuda.device(...)"
                                 % (t.device, torch.cuda.current_device()))
    return ctypes.c_void_p(t.data_ptr())


def stream():
    """The current stream of the current device (ptr() checks that every tensor of the call lives on that device)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    """Invoke an entry point on the current torch stream; raise with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (name, rc, lib.trec_last_error().decode()))


def set_tuning(name, value):
    load().trec_set_tuning(name.encode(), int(value))


def query(name, *args):
    """Entry points that return a value instead of a status (no stream argument)."""
    return getattr(load(), name)(*args)
