"""ctypes loader for libsglk.so (the C ABI declared in include/sglk.h).

The library is the product: there is no CPU or torch fallback.  If it is missing, loading fails
loudly with instructions to build it.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGLK_LIB") or os.path.join(_PKG, "libsglk.so")  # SGLK_LIB: a development build

F32, F16, BF16, FP8_E4M3 = 0, 1, 2, 3

_lib = None

p, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, [argtypes]); mirrors include/sglk.h one to one
_SIGS = {
    "sglk_last_error": (C.c_char_p, []),
    "sglk_version": (i32, []),
    "sglk_device_cu_count": (i32, []),
    "sglk_create_kv_indices": (i32, [p, i64, p, i32, p, i32, p, p, i32, p, i32, p]),
    "sglk_kv_indptr_cumsum": (i32, [p, i32, p, i32, p]),
    "sglk_get_num_kv_splits": (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, p]),
    "sglk_decode_metadata": (i32, [p, i64, p, i32, p, i32, p, p, p, i32, i32, i32, i32, i32, p]),
    "sglk_decode_attention": (i32, [p, p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32,
                                    i64, i64, i64, i64, i64, i64, i64, i64, f32, f32, i32, p]),
    "sglk_top_k_renorm_probs": (i32, [p, p, p, i32, i32, i32, i64, i64, p]),
    "sglk_top_p_renorm_probs": (i32, [p, p, p, f32, i32, i32, i64, i64, p]),
    "sglk_sampling_from_probs": (i32, [p, p, p, p, i32, p, f32, p, f32, i32, i32, i32, i64, p]),
    "sglk_build_tree_efficient": (i32, [p] * 8 + [i32] * 4 + [i64, p]),
    "sglk_verify_tree_greedy": (i32, [p] * 8 + [i32] * 4 + [p]),
    "sglk_tree_speculative_sampling_target_only": (i32, [p] * 10 + [i32] * 5 + [f32, f32, p]),
    "sglk_gptq_repack": (i32, [p, p, p, p, p, i32, i32, i32, i32, p]),
    "sglk_segment_packbits": (i32, [p, p, p, p, i32, p]),
    "sglk_moe_fused_gate": (i32, [p, p, p, p, i32, i32, i32, i32, i32, i32, f32, i64, i32, p]),
    "sglk_mla_rope_prepare": (i32, [p] * 8 + [i32] * 5 + [i64] * 7 + [i32, i32, p]),
    "sglk_prepare_for_decode": (i32, [p, i64, p, i32, p, i32, p, p, i32, i32, p]),
    "sglk_argmax": (i32, [p, p, i64, i32, i64, i32, p]),
    "sglk_decode_shared_plan_bytes": (i64, [i32, i32, i32, i32]),
    "sglk_decode_shared_plan": (i32, [p, i64, p, i32, p, i32, p, p, i32, i32, i32, i32, i32, p, p]),
    "sglk_decode_shared_plan_from_groups": (i32, [p, p, p, p, i32, i32, i32, i32, i32, p, p]),
    "sglk_decode_metadata_shared": (i32, [p, i64, p, i32, p, i32, p, p, p, i32, i32, i32, i32, i32, i32, p, p]),
    "sglk_decode_attention_shared_planned": (i32, [p] * 8 + [i32] * 6 + [i64] * 8 + [f32, f32, i32, p, i64, p, p]),
    "sglk_decode_attention_shared_planned_rope": (i32, [p, i32, i64, p, p, p, i32, i32, p, i32, p, i64, p, p, p, p, p, p, i32,
                                                        p] + [i32] * 6 + [i64] * 6 + [f32, f32, i32, p, i64, p]),
    "sglk_decode_shared_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "sglk_decode_shared_workspace_init": (i32, [p, i32, i32, p]),
    "sglk_decode_shared_workspace_status": (i32, [p, i32, i32]),
    "sglk_decode_attention_shared_prefix": (i32, [p, p, p, p, p, p, p, p, i32, p, i32, i32, i32, i32, i32, i32,
                                                  i64, i64, i64, i64, i64, i64, i64, i64, f32, f32, i32, p]),
    "sglk_common_prefix_len": (i32, [p, i64, p, i32, p, i32, i32, p, p]),
    "sglk_extend_attention": (i32, [p, p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32,
                                    i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64,
                                    i32, f32, f32, i32, i32, p]),
    "sglk_extend_attention_masked": (i32, [p] * 11 + [i32] * 6 + [i64] * 12 + [i32, f32, f32, i32, p]),
    "sglk_extend_attention_split_workspace_bytes": (i64, [i64, i32, i32]),
    "sglk_extend_attention_split": (i32, [p] * 9 + [i32] * 5 + [i64] * 12 + [i32, f32, f32, i32, i32, p, i64, i32, p]),
    "sglk_extend_attention_window": (i32, [p] * 11 + [i32] * 7 + [i64] * 12 + [i32, f32, f32, i32, i32, p]),
    "sglk_prefill_attention_paged": (i32, [p] * 7 + [i32] * 5 + [i64] * 8 + [i32, f32, f32, i32, i32, p]),
    "sglk_merge_state": (i32, [p, p, p, p, p, p, i64, i32, i32, i32, p]),
    "sglk_set_kv_buffer": (i32, [p, p, p, p, p, i32, i64, i32, i32, i64, i64, i64, i64, i32, p]),
    "sglk_set_kv_buffer_fp8": (i32, [p, p, p, p, p, i32, i64, i32, i32, i64, i64, i64, i64, f32, f32, i32, i32, p]),
    "sglk_rmsnorm": (i32, [p, p, p, i64, i32, i64, i64, f32, i32, i32, p]),
    "sglk_fused_add_rmsnorm": (i32, [p, p, p, i64, i32, f32, i32, i32, p]),
    "sglk_act_and_mul": (i32, [p, p, i64, i32, i32, i32, p]),
    "sglk_rope": (i32, [p, p, p, p, i64, i32, i32, i32, i32, i64, i64, i32, i32, p]),
    "sglk_rope_and_store_kv": (i32, [p, p, p, p, p, p, i32, i64, i32, i32, i32, i32, i64, i64, i32, i32, p]),
    "sglk_awq_dequantize": (i32, [p, p, p, p, i32, i32, i32, i32, p]),
    "sglk_awq_dequantize_packed": (i32, [p, p, p, p, i32, i32, i32, i32, p]),
    "sglk_awq_repack": (i32, [p, p, i32, i32, p]),
    "sglk_awq_gemm_workspace_bytes": (i64, [i32, i32, i32]),
    "sglk_awq_gemm": (i32, [p, p, p, p, p, p, p, i32, i32, i32, i32, i64, i64, i32, p]),
    "sglk_awq_gemm_static_max_workspace_bytes": (i64, []),
    "sglk_awq_gemm_static_init": (i32, []),
    "sglk_awq_gemm_static": (i32, [p, p, p, p, p, p, i32, i32, i32, i32, i64, i64, i32, p]),
    "sglk_awq_gemm_xs": (i32, [p, p, p, p, p, p, p, i32, i32, i32, i32, i64, i64, i32, p, i64, p]),
    "sglk_awq_gemm_partial_splits": (i32, [i32, i32, i32, i32]),
    "sglk_awq_gemm_partial": (i32, [p, p, p, p, p, i64, p, i32, i32, i32, i32, i64, i32, p]),
    "sglk_fused_add_rmsnorm_ex": (i32, [p, p, p, i64, i32, f32, i32, i32, p, i32, i64, p, p, i64, p, i64, p]),
    "sglk_awq_gemm_silu_mul": (i32, [p, p, p, p, p, i32, i32, i32, i32, i64, i64, i32, p, i64, p, i64, p]),
    "sglk_act_and_mul_xs": (i32, [p, p, i64, i32, i32, i32, p, i64, p]),
    "sglk_rope_and_store_kv_partial": (i32, [p, p, i32, i64, p, p, p, p, p, p, i32, i64, i32, i32, i32, i32, i64, i64, i32, i32, p]),
    "sglk_decode_attention_xs": (i32, [p] * 9 + [i32] * 6 + [i64] * 8 + [f32, f32, i32, p, i64, p]),
    "sglk_decode_attention_shared_prefix_rope": (i32, [p, i32, i64, p, p, p, i32, i32, p, i32] + [p] * 7 + [i32, p] +
                                                 [i32] * 6 + [i64] * 6 + [f32, f32, i32, p, i64, p]),
    "sglk_decode_attention_shared_prefix_xs": (i32, [p] * 8 + [i32, p] + [i32] * 6 + [i64] * 8 + [f32, f32, i32, p, i64, p]),
    "sglk_dense_gemm_nt": (i32, [p, p, p, p, i32, i32, i32, i64, i64, i64, i32, p]),
    "sglk_dense_gemm_workspace_bytes": (i64, [i32, i32, i32]),
    "sglk_awq_gemm_prefill": (i32, [p, p, p, p, p, p, i32, i32, i32, i32, i64, i64, i32, p]),
    "sglk_dense_gemm_prefill": (i32, [p, p, p, p, i32, i32, i32, i64, i64, i64, i32, p]),
    "sglk_dense_repack": (i32, [p, p, i32, i32, i64, p]),
    "sglk_dense_gemm_packed": (i32, [p, p, p, p, i32, i32, i32, i64, i64, i32, p]),
    "sglk_dense_gemm_partial_splits": (i32, [i32, i32, i32]),
    "sglk_dense_gemm_packed_partial": (i32, [p, p, p, i32, i32, i32, i64, i32, p]),
    "sglk_dense_gemm_packed_ws": (i32, [p, p, p, p, i32, i32, i32, i64, i64, p, i64, i32, p]),
    "sglk_dense_gemm_packed_act": (i32, [p, p, p, i32, i32, i32, i64, i64, i32, p, i64, i32, p]),
    "sglk_dense_gemm_nt_ws": (i32, [p, p, p, p, i32, i32, i32, i64, i64, i64, p, i64, i32, p]),
    "sglk_per_token_quant_fp8": (i32, [p, p, p, i64, i32, i32, p]),
    "sglk_per_token_group_quant_fp8": (i32, [p, p, p, i64, i32, i32, f32, f32, f32, i32, i32, p]),
    "sglk_per_tensor_quant_fp8": (i32, [p, p, p, i64, i32, i32, p]),
    "sglk_per_tensor_quant_fp8_rows": (i32, [p, p, p, i64, i64, i64, i64, i32, i32, p]),
    "sglk_mla_rope_and_store": (i32, [p, p, p, p, p, i32, p, p] + [i32] * 5 + [i64] * 7 + [i32, i32, p]),
    "sglk_fp8_scaled_mm": (i32, [p, p, p, p, p, p, i32, i32, i32, i64, i64, i64, i32, p]),
    "sglk_fp8_blockwise_scaled_mm": (i32, [p, p, p, p, p, i32, i32, i32, i64, i64, i64, i64, i32, p]),
    "sglk_rmsnorm_quant_fp8": (i32, [p, p, p, p, p, i64, i32, f32, i32, i32, p]),
    "sglk_silu_and_mul_quant_fp8": (i32, [p, p, p, i64, i32, i32, p]),
    "sglk_rmsnorm_group_quant_fp8": (i32, [p, p, p, p, p, i64, i32, i64, f32, i32, i32, p]),
    "sglk_silu_and_mul_group_quant_fp8": (i32, [p, p, p, i64, i32, i32, p]),
    "sglk_fp8_mm_workspace_bytes": (i64, [i32, i32, i32]),
    "sglk_fp8_scaled_mm_ws": (i32, [p, p, p, p, p, p, i32, i32, i32, i64, i64, i64, p, i64, i32, p]),
    "sglk_fp8_repack": (i32, [p, p, i32, i32, i64, p]),
    "sglk_fp8_blockwise_scaled_mm_packed": (i32, [p, p, p, p, p, i32, i32, i32, i64, i64, i64, i64, p, i64, i32, p]),
    "sglk_fp8_mm_packed_partial_splits": (i32, [i32, i32, i32]),
    "sglk_fp8_mm_packed_partial": (i32, [p, p, p, i32, i32, i32, i64, p]),
    "sglk_silu_and_mul_quant_fp8_partial": (i32, [p, i32, i64, p, p, p, p, i64, i32, i32, p]),
    "sglk_rmsnorm_quant_fp8_partial": (i32, [p, i32, i64, p, p, p, p, p, p, p, i64, i32, f32, i32, i32, p]),
    "sglk_rope_and_store_kv_partial_scaled": (i32, [p, p, i32, i64, p, p, p, p, p, p, p, p, i32, i64, i32, i32, i32, i32,
                                               i64, i64, i32, i32, p]),
    "sglk_fp8_mm_packed_workspace_bytes": (i64, [i32, i32, i32]),
    "sglk_fp8_scaled_mm_packed": (i32, [p, p, p, p, p, p, i32, i32, i32, i64, i64, p, i64, i32, p]),
    "sglk_fp8_blockwise_scaled_mm_ws": (i32, [p, p, p, p, p, i32, i32, i32, i64, i64, i64, i64, p, i64, i32, p]),
    "sglk_bmm_fp8": (i32, [p, p, p, p, p, i32, i32, i32, i32, i64, i64, i64, i64, i64, i64, i32, i32, i32, p]),
    "sglk_moe_align_block_size": (i32, [p, i32, i64, i32, i32, p, p, p, p, p]),
    "sglk_topk_softmax": (i32, [p, p, p, p, i32, i32, i32, i64, i32, p]),
    "sglk_awq_chain_sync_bytes": (i64, []),
    "sglk_awq_chain_status": (i32, [p]),
    "sglk_awq_chain": (i32, [p, i32, p, i32, i32, i32, p]),
    "sglk_ar_meta_size": (i64, []),
    "sglk_ar_init": (i64, [p, i32, i32, i32, i64]),
    "sglk_ar_init_handles": (i64, [p, p, i64, p, p, i32, i32, i32, i64]),
    "sglk_ar_dispose": (i32, [i64]),
    "sglk_ar_register_buffer_handles": (i32, [i64, p, p, p]),
    "sglk_ar_graph_buffer_count": (i64, [i64]),
    "sglk_ar_get_graph_buffer_ipc_meta": (i32, [i64, p, p, i64]),
    "sglk_ar_register_graph_buffers": (i32, [i64, p, p, i64]),
    "sglk_ar_register_buffer": (i32, [i64, p, p]),
    "sglk_ar_all_reduce": (i32, [i64, p, p, i64, i32, i32, p]),
    "sglk_ar_all_reduce_unreg": (i32, [i64, p, p, p, i64, i32, i32, p]),
    "sglk_ar_stuck": (i32, [i64]),
    "sglk_ar_set_timeout_ms": (i32, [i64, i64]),
    "sglk_ar_all_gather": (i32, [i64, p, p, i64, p]),
    "sglk_ipc_handle_bytes": (i32, []),
    "sglk_ipc_get_handle": (i32, [p, p, p]),
    "sglk_ipc_open_handle": (p, [p]),
    "sglk_ipc_close": (i32, [p]),
    "sglk_alloc_uncached": (p, [i64]),
    "sglk_free_uncached": (i32, [p]),
    "sglk_radix_create": (i64, [i32]),
    "sglk_radix_destroy": (None, [i64]),
    "sglk_radix_reset": (None, [i64]),
    "sglk_radix_set_epoch": (None, [i64, i64]),
    "sglk_radix_match_prefix": (i64, [i64, p, i64, i64, p, i64, p]),
    "sglk_radix_insert": (i64, [i64, p, p, i64, i64, p, i64, p]),
    "sglk_radix_inc_lock_ref": (i64, [i64, i64]),
    "sglk_radix_dec_lock_ref": (i64, [i64, i64]),
    "sglk_radix_evict": (None, [i64, i64]),
    "sglk_radix_take_freed": (i64, [i64, p, i64]),
    "sglk_radix_evictable_size": (i64, [i64]),
    "sglk_radix_protected_size": (i64, [i64]),
    "sglk_radix_total_size": (i64, [i64]),
    "sglk_radix_root": (i64, [i64]),
    "sglk_radix_live_nodes": (i64, [i64]),
    "sglk_radix_node_set_privacy": (i32, [i64, i64, i32, i32]),
    "sglk_radix_node_get": (i32, [i64, i64, p]),
    "sglk_radix_node_key": (i64, [i64, i64, p, p, i64]),
    "sglk_radix_dump": (i64, [i64, p, i64]),
}


_PENDING = set()


class SglkError(RuntimeError):
    """A launcher rejected its arguments or a HIP call failed (the reference raises RuntimeError from
    TORCH_CHECK in the same situations)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP kernel library has not been built. "
                "Run `python -m sglang_security_amd.build` (needs hipcc). There is no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if name in _PENDING:
                continue
            fn = getattr(L, name)  # AttributeError if the library is stale: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().sglk_last_error().decode(errors="replace")
        raise SglkError(f"{what}: {msg}" if what else msg)


def dtype_code(dt) -> int:
    import torch

    try:
        return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16, torch.float8_e4m3fn: FP8_E4M3}[dt]
    except KeyError:
        raise SglkError(f"unsupported dtype {dt}")


def stream_ptr(device=None) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream
