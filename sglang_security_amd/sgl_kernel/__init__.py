"""Drop-in operator surface: the functions of the reference's `sgl_kernel` package that lie on the
RadixAttention + quantized-matmul decode path, with the reference's names, argument order, defaults and error
behaviour, executed by libsglk (HIP, gfx950).

    import sglang_security_amd.sgl_kernel as sgl_kernel        # or: sglang_security_amd.sgl_kernel.install()

Each function cites the reference wrapper it replaces (sgl-kernel/python/sgl_kernel/*.py).  Arguments that only
make sense on NVIDIA hardware (enable_pdl) are accepted and ignored.  There is no CPU fallback: CPU tensors
raise SglkError.
"""
from __future__ import annotations

import sys
from typing import List, Optional, Tuple

import torch

from .. import ops as _ops
from .._lib import SglkError  # noqa: F401

__all__ = [
    "rmsnorm", "fused_add_rmsnorm", "gemma_rmsnorm", "gemma_fused_add_rmsnorm", "silu_and_mul", "gelu_tanh_and_mul",
    "gelu_and_mul", "apply_rope_with_cos_sin_cache_inplace", "awq_dequantize", "fp8_scaled_mm",
    "fp8_blockwise_scaled_mm", "bmm_fp8", "sgl_per_token_quant_fp8", "sgl_per_token_group_quant_fp8",
    "sgl_per_tensor_quant_fp8", "moe_align_block_size", "topk_softmax", "top_k_renorm_probs", "top_k_renorm_prob",
    "top_p_renorm_probs", "top_p_renorm_prob", "top_p_sampling_from_probs", "top_k_top_p_sampling_from_probs",
    "min_p_sampling_from_probs", "build_tree_kernel_efficient", "verify_tree_greedy",
    "tree_speculative_sampling_target_only", "segment_packbits", "moe_fused_gate", "merge_state", "merge_state_v2", "install",
]


def _rows(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1])


# ---- elementwise.py ---------------------------------------------------------------------------------------
def rmsnorm(input: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, out: Optional[torch.Tensor] = None,
            enable_pdl: Optional[bool] = None) -> torch.Tensor:
    """elementwise.py:9-45."""
    if out is None:
        out = torch.empty_like(input)
    _ops.rmsnorm(_rows(out), _rows(input), weight, eps, gemma=False)
    return out


def fused_add_rmsnorm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
                      enable_pdl: Optional[bool] = None) -> None:
    """elementwise.py:48-82: residual += input; input = rmsnorm(residual) * weight, both in place."""
    _ops.fused_add_rmsnorm(_rows(input), _rows(residual), weight, eps, gemma=False)


def gemma_rmsnorm(input: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, out: Optional[torch.Tensor] = None,
                  enable_pdl: Optional[bool] = None) -> torch.Tensor:
    """elementwise.py:85-121: scale is (1 + weight)."""
    if out is None:
        out = torch.empty_like(input)
    _ops.rmsnorm(_rows(out), _rows(input), weight, eps, gemma=True)
    return out


def gemma_fused_add_rmsnorm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
                            enable_pdl: Optional[bool] = None) -> None:
    """elementwise.py:124-158."""
    _ops.fused_add_rmsnorm(_rows(input), _rows(residual), weight, eps, gemma=True)


def _gated(input: torch.Tensor, out: Optional[torch.Tensor], act: int) -> torch.Tensor:
    # elementwise.py:161-183: same checks, same messages
    if input.shape[-1] * input.dtype.itemsize % 16 != 0:
        raise ValueError("The pointers must be multiple of 16 bytes.")
    if out is not None:
        assert input.ndim == out.ndim, f"{input.ndim} != {out.ndim}"
        assert input.shape[:-1] == out.shape[:-1], f"{input.shape[:-1]} != {out.shape[:-1]}"
        assert input.shape[-1] == 2 * out.shape[-1], f"{input.shape[-1]} != {2 * out.shape[-1]}"
    else:
        out = torch.empty(input.shape[:-1] + (input.shape[-1] // 2,), device=input.device, dtype=input.dtype)
    _ops.act_and_mul(_rows(out), _rows(input), act)
    return out


def silu_and_mul(input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """elementwise.py:171-183."""
    return _gated(input, out, 0)


def gelu_tanh_and_mul(input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """elementwise.py:186-198."""
    return _gated(input, out, 1)


def gelu_and_mul(input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """elementwise.py:201-213."""
    return _gated(input, out, 2)


def apply_rope_with_cos_sin_cache_inplace(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                                          head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool = True) -> None:
    """elementwise.py:216-268: query [nnz, Hq*head_size], key [nnz, Hk*head_size], rotated in place."""
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    _ops.rope(query.view(query.shape[0], -1, head_size), key.view(key.shape[0], -1, head_size), cos_sin_cache,
              positions.long(), head_size, interleave=not is_neox)


# ---- gemm.py ------------------------------------------------------------------------------------------------
def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> torch.Tensor:
    """gemm.py:7-10."""
    return _ops.awq_dequantize(qweight, scales, qzeros)


def fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias=None):
    """gemm.py:34-42."""
    return _ops.fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias)


def fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype):
    """gemm.py:24-31."""
    return _ops.fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype)


def bmm_fp8(A, B, A_scale, B_scale, dtype, out=None):
    """gemm.py:66-82."""
    return _ops.bmm_fp8(A, B, A_scale, B_scale, dtype, out)


# ---- sampling.py ----------------------------------------------------------------------------------------------
def top_k_renorm_probs(probs: torch.Tensor, top_k) -> torch.Tensor:
    """sampling.py:21-52."""
    return _ops.top_k_renorm_probs(probs, top_k)


top_k_renorm_prob = top_k_renorm_probs  # sampling.py:52


def top_p_renorm_probs(probs: torch.Tensor, top_p) -> torch.Tensor:
    """sampling.py:69-102."""
    return _ops.top_p_renorm_probs(probs, top_p)


top_p_renorm_prob = top_p_renorm_probs  # sampling.py:102


def _uniform(probs, generator):
    return torch.rand(probs.shape[0], dtype=torch.float32, device=probs.device, generator=generator)


def top_p_sampling_from_probs(probs, top_p, indices=None, deterministic=True, generator=None, check_nan=False):
    """sampling.py:131-185 (indices: rows of probs to sample from, one per output)."""
    if check_nan and torch.any(torch.isnan(probs)):
        raise ValueError("Input probs contains NaN.")
    src = probs if indices is None else probs[indices.long()]
    return _ops.sampling_from_probs(src, _uniform(src, generator), 0, top_p, 0.0, True)


def top_k_top_p_sampling_from_probs(probs, top_k, top_p, indices=None, filter_apply_order="top_k_first",
                                    deterministic=True, generator=None, check_nan=False):
    """sampling.py:219-300: "top_k_first" renormalises after top-k and then cuts by top-p, "joint" applies both
    filters to the original distribution."""
    if filter_apply_order not in ("top_k_first", "joint"):
        raise ValueError(f"Invalid filter_apply_order: {filter_apply_order}")
    if check_nan and torch.any(torch.isnan(probs)):
        raise ValueError("Input probs contains NaN.")
    src = probs if indices is None else probs[indices.long()]
    return _ops.sampling_from_probs(src, _uniform(src, generator), top_k, top_p, 0.0, filter_apply_order == "joint")


def min_p_sampling_from_probs(probs, min_p, indices=None, deterministic=True, generator=None, check_nan=False):
    """sampling.py:303-360."""
    if check_nan and torch.any(torch.isnan(probs)):
        raise ValueError("Input probs contains NaN.")
    src = probs if indices is None else probs[indices.long()]
    return _ops.sampling_from_probs(src, _uniform(src, generator), 0, 1.0, min_p, True)


# ---- speculative.py ------------------------------------------------------------------------------------------
def build_tree_kernel_efficient(parent_list, selected_index, verified_seq_len, tree_mask, positions, retrive_index,
                                retrive_next_token, retrive_next_sibling, topk, depth, draft_token_num) -> None:
    """speculative.py:61-86."""
    _ops.build_tree_kernel_efficient(parent_list, selected_index, verified_seq_len, tree_mask, positions, retrive_index,
                                     retrive_next_token, retrive_next_sibling, topk, depth, draft_token_num)


def verify_tree_greedy(predicts, accept_index, accept_token_num, candidates, retrive_index, retrive_next_token,
                       retrive_next_sibling, target_predict) -> None:
    """speculative.py:38-58."""
    _ops.verify_tree_greedy(predicts, accept_index, accept_token_num, candidates, retrive_index, retrive_next_token,
                            retrive_next_sibling, target_predict)


def tree_speculative_sampling_target_only(predicts, accept_index, accept_token_num, candidates, retrive_index,
                                          retrive_next_token, retrive_next_sibling, uniform_samples, target_probs,
                                          draft_probs, threshold_single: float = 1.0, threshold_acc: float = 1.0,
                                          deterministic: bool = True) -> None:
    """speculative.py:5-35."""
    _ops.tree_speculative_sampling_target_only(predicts, accept_index, accept_token_num, candidates, retrive_index,
                                               retrive_next_token, retrive_next_sibling, uniform_samples, target_probs,
                                               draft_probs, threshold_single, threshold_acc, deterministic)


def segment_packbits(x, input_indptr, output_indptr, y) -> None:
    """speculative.py:89-101."""
    _ops.segment_packbits(x, input_indptr, output_indptr, y)


# ---- moe.py -------------------------------------------------------------------------------------------------
def moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad,
                         token_cnts_buffer, cumsum_buffer):
    """moe.py:4-23."""
    _ops.moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad,
                              token_cnts_buffer, cumsum_buffer)


def moe_fused_gate(input_tensor, bias, num_expert_group, topk_group, topk, n_share_experts_fusion=0,
                   routed_scaling_factor=0):
    """moe.py:37-62."""
    return _ops.moe_fused_gate(input_tensor, bias, num_expert_group, topk_group, topk, n_share_experts_fusion,
                               routed_scaling_factor)


def topk_softmax(topk_weights, topk_ids, token_expert_indices, gating_output) -> None:
    """moe.py:26-34."""
    _ops.topk_softmax(topk_weights, topk_ids, token_expert_indices, gating_output)


def sgl_per_token_group_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor, group_size: int,
                                  eps: float, fp8_min: float, fp8_max: float) -> None:
    """gemm.py:85-96."""
    _ops.per_token_group_quant_fp8(input, output_q, output_s, group_size, eps, fp8_min, fp8_max)


def sgl_per_tensor_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor,
                             is_static: bool) -> None:
    """gemm.py:113-121."""
    _ops.per_tensor_quant_fp8(input, output_q, output_s, is_static)


def sgl_per_token_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor) -> None:
    """gemm.py:124-129."""
    _ops.per_token_quant_fp8(input, output_q, output_s)


# ---- attention.py ---------------------------------------------------------------------------------------------
def merge_state_v2(v_a: torch.Tensor, s_a: torch.Tensor, v_b: torch.Tensor, s_b: torch.Tensor,
                   v_merged: Optional[torch.Tensor] = None,
                   s_merged: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """attention.py:31-51."""
    s_a = s_a.to(torch.float32)
    s_b = s_b.to(torch.float32)
    if v_merged is None:
        v_merged = torch.empty_like(v_a)
    if s_merged is None:
        s_merged = torch.empty_like(s_a)
    _ops.merge_state(v_a, s_a, v_b, s_b, v_merged, s_merged)
    return v_merged, s_merged


merge_state = merge_state_v2  # attention.py:12-28: same contract, older kernel


# ---- allreduce.py (ROCm branch, :5-50) --------------------------------------------------------------------------
_ALLREDUCE = ("init_custom_ar", "all_reduce_reg", "all_reduce_unreg", "dispose", "meta_size", "register_buffer",
              "get_graph_buffer_ipc_meta", "register_graph_buffers", "allocate_meta_buffer", "get_meta_buffer_ipc_handle")
__all__ += list(_ALLREDUCE)


def __getattr__(name: str):
    # The all-reduce functions live in the `allreduce` submodule, which loads common_ops.so (torch op registrations
    # from C++) when first imported; importing it lazily keeps `import sgl_kernel` free of that side effect until a
    # communicator is built, and `from sgl_kernel import init_custom_ar` / `sgl_kernel.allreduce.*` both work, as with
    # the reference package (python/sgl_kernel/__init__.py:15).
    if name in _ALLREDUCE or name == "allreduce":
        import importlib

        mod = importlib.import_module(__name__ + ".allreduce")
        return mod if name == "allreduce" else getattr(mod, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def install(name: str = "sgl_kernel") -> None:
    """Make `import sgl_kernel` resolve to this module (the hook a deployment uses to swap the kernels in)."""
    sys.modules[name] = sys.modules[__name__]
    import importlib

    sys.modules[name + ".allreduce"] = importlib.import_module(__name__ + ".allreduce")
