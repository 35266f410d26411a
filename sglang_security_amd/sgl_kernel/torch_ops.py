"""`torch.ops.sgl_kernel.*` registrations for the hot-path ops, with the reference's schemas
(sgl-kernel/csrc/common_extension.cc:48-121, torch_extension_rocm.cc:60-84; the reference registers them from C++ with TORCH_LIBRARY_FRAGMENT,
here torch.library does the same from Python and the implementations call libsglk through its C ABI).

    from sglang_security_amd.sgl_kernel.torch_ops import register_torch_ops
    register_torch_ops()        # idempotent; "CUDA" dispatch key = HIP device on ROCm
"""
from __future__ import annotations

import os

import torch

from .. import ops as _ops

COMMON_OPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "common_ops.so")

_SCHEMAS = {
    "awq_dequantize": "(Tensor qweight, Tensor scales, Tensor qzeros) -> Tensor",
    "fp8_scaled_mm": "(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype, Tensor? bias) -> Tensor",
    "fp8_blockwise_scaled_mm": "(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, ScalarType out_dtype) -> Tensor",
    "moe_align_block_size": "(Tensor topk_ids, int num_experts, int block_size, Tensor! sorted_token_ids, Tensor! experts_ids, Tensor! num_tokens_post_pad, Tensor! token_cnts_buffer, Tensor! cumsum_buffer) -> ()",
    "topk_softmax": "(Tensor! topk_weights, Tensor! topk_indices, Tensor! token_expert_indices, Tensor gating_output) -> ()",
    "sgl_per_token_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s) -> ()",
    "sgl_per_token_group_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s, int group_size, float eps, float fp8_min, float fp8_max) -> ()",
    "sgl_per_tensor_quant_fp8": "(Tensor input, Tensor output_q, Tensor output_s, bool is_static) -> ()",
    "merge_state_v2": "(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()",
    "merge_state": "(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, Tensor! s_merged) -> ()",
    "rmsnorm": "(Tensor! output, Tensor input, Tensor weight, float eps, bool enable_pdl) -> ()",
    "fused_add_rmsnorm": "(Tensor! input, Tensor! residual, Tensor weight, float eps, bool enable_pdl) -> ()",
    "gemma_rmsnorm": "(Tensor! output, Tensor input, Tensor weight, float eps, bool enable_pdl) -> ()",
    "gemma_fused_add_rmsnorm": "(Tensor! input, Tensor! residual, Tensor weight, float eps, bool enable_pdl) -> ()",
    "silu_and_mul": "(Tensor! out, Tensor input, int cuda_stream) -> ()",
    "gelu_tanh_and_mul": "(Tensor! out, Tensor input, int cuda_stream) -> ()",
    "gelu_and_mul": "(Tensor! out, Tensor input, int cuda_stream) -> ()",
    "apply_rope_pos_ids_cos_sin_cache": "(Tensor q, Tensor k, Tensor! q_rope, Tensor! k_rope, Tensor cos_sin_cache, Tensor pos_ids, bool interleave, int cuda_stream) -> ()",
    "moe_fused_gate": "(Tensor input, Tensor bias, int num_expert_group, int topk_group, int topk, int n_share_experts_fusion, float routed_scaling_factor) -> Tensor[]",
    "verify_tree_greedy": "(Tensor! predicts, Tensor! accept_index, Tensor! accept_token_num, Tensor candidates, Tensor retrive_index, Tensor retrive_next_token, Tensor retrive_next_sibling, Tensor target_predict, int cuda_stream) -> ()",
    "build_tree_kernel_efficient": "(Tensor parent_list, Tensor selected_index, Tensor verified_seq_len, Tensor! tree_mask, Tensor! positions, Tensor! retrive_index, Tensor! retrive_next_token, Tensor! retrive_next_sibling, int topk, int depth, int draft_token_num) -> ()",
    "tree_speculative_sampling_target_only": "(Tensor! predicts, Tensor! accept_index, Tensor! accept_token_num, Tensor candidates, Tensor retrive_index, Tensor retrive_next_token, Tensor retrive_next_sibling, Tensor uniform_samples, Tensor target_probs, Tensor! draft_probs, float threshold_single, float threshold_acc, bool deterministic, int cuda_stream) -> ()",
    # the fast path the reference does not have: fused dequantise + GEMM on the re-laid-out weight
    "awq_gemm": "(Tensor x, Tensor packed, Tensor scales, Tensor qzeros, Tensor? bias, Tensor workspace) -> Tensor",
}

_lib = None


def _rows(t):
    return t.reshape(-1, t.shape[-1])


def _rope(q, k, q_rope, k_rope, cos_sin_cache, pos_ids, interleave, cuda_stream):
    if q_rope.data_ptr() != q.data_ptr():
        q_rope.copy_(q)
    if k_rope.data_ptr() != k.data_ptr():
        k_rope.copy_(k)
    _ops.rope(q_rope, k_rope, cos_sin_cache, pos_ids.long(), q.shape[-1], interleave)


_IMPLS = {
    "awq_dequantize": lambda qweight, scales, qzeros: _ops.awq_dequantize(qweight, scales, qzeros),
    "fp8_scaled_mm": lambda a, b, sa, sb, dt, bias: _ops.fp8_scaled_mm(a, b, sa, sb, dt, bias),
    "fp8_blockwise_scaled_mm": lambda a, b, sa, sb, dt: _ops.fp8_blockwise_scaled_mm(a, b, sa, sb, dt),
    "moe_align_block_size": lambda ids, ne, bs, srt, ex, npp, cnt, cs: _ops.moe_align_block_size(ids, ne, bs, srt, ex, npp, cnt, cs),
    "topk_softmax": lambda w, ids, src, g: _ops.topk_softmax(w, ids, src, g),
    "sgl_per_token_quant_fp8": lambda x, q, s: _ops.per_token_quant_fp8(x, q, s),
    "sgl_per_token_group_quant_fp8": lambda x, q, s, g, eps, lo, hi: _ops.per_token_group_quant_fp8(x, q, s, g, eps, lo, hi),
    "sgl_per_tensor_quant_fp8": lambda x, q, s, st: _ops.per_tensor_quant_fp8(x, q, s, st),
    "merge_state_v2": lambda va, sa, vb, sb, vm, sm: _ops.merge_state(va, sa, vb, sb, vm, sm),
    "merge_state": lambda va, sa, vb, sb, vm, sm: _ops.merge_state(va, sa, vb, sb, vm, sm),
    "rmsnorm": lambda out, x, w, eps, pdl: _ops.rmsnorm(_rows(out), _rows(x), w, eps, False),
    "fused_add_rmsnorm": lambda x, r, w, eps, pdl: _ops.fused_add_rmsnorm(_rows(x), _rows(r), w, eps, False),
    "gemma_rmsnorm": lambda out, x, w, eps, pdl: _ops.rmsnorm(_rows(out), _rows(x), w, eps, True),
    "gemma_fused_add_rmsnorm": lambda x, r, w, eps, pdl: _ops.fused_add_rmsnorm(_rows(x), _rows(r), w, eps, True),
    "silu_and_mul": lambda out, x, stream: _ops.act_and_mul(_rows(out), _rows(x), 0),
    "gelu_tanh_and_mul": lambda out, x, stream: _ops.act_and_mul(_rows(out), _rows(x), 1),
    "gelu_and_mul": lambda out, x, stream: _ops.act_and_mul(_rows(out), _rows(x), 2),
    "apply_rope_pos_ids_cos_sin_cache": _rope,
    "moe_fused_gate": lambda x, b, ng, tg, k, ns, rs: list(_ops.moe_fused_gate(x, b, ng, tg, k, ns, rs)),
    "verify_tree_greedy": lambda pr, ai, an, c, ri, nt, ns, tp, stream: _ops.verify_tree_greedy(pr, ai, an, c, ri, nt, ns, tp),
    "build_tree_kernel_efficient": lambda pl, si, sl, m, pos, ri, nt, ns, topk, depth, n: _ops.build_tree_kernel_efficient(pl, si, sl, m, pos, ri, nt, ns, topk, depth, n),
    "tree_speculative_sampling_target_only": lambda pr, ai, an, c, ri, nt, ns, u, tp, dp, ts, ta, det, stream: _ops.tree_speculative_sampling_target_only(pr, ai, an, c, ri, nt, ns, u, tp, dp, ts, ta, det),
    "awq_gemm": lambda x, packed, scales, qzeros, bias, ws: _ops.awq_gemm(x, packed, scales, qzeros, bias, None, ws),
}


# Shape functions for the "Meta" key: what a tracer (torch.compile, FakeTensorMode, make_fx) needs to see through an
# op without running it.  The in-place ops (`Tensor!` arguments, no result) have nothing to compute.
def _meta_awq_dequantize(qweight, scales, qzeros):
    return scales.new_empty((qweight.shape[0], qweight.shape[1] * 8))


def _meta_scaled_mm(a, b, sa, sb, dt, bias=None):
    return a.new_empty((a.shape[0], b.shape[1]), dtype=dt)


def _meta_moe_fused_gate(x, bias, ng, tg, k, ns, rs):
    return [x.new_empty((x.shape[0], k), dtype=torch.float32), x.new_empty((x.shape[0], k), dtype=torch.int32)]


def _meta_awq_gemm(x, packed, scales, qzeros, bias, ws):
    return x.new_empty(x.shape[:-1] + (scales.shape[-1],))


_META = {
    "awq_dequantize": _meta_awq_dequantize,
    "fp8_scaled_mm": _meta_scaled_mm,
    "fp8_blockwise_scaled_mm": _meta_scaled_mm,
    "moe_fused_gate": _meta_moe_fused_gate,
    "awq_gemm": _meta_awq_gemm,
}


def register_torch_ops() -> None:
    global _lib
    if _lib is not None:
        return
    from .._lib import lib as _load_libsglk

    lib = torch.library.Library("sgl_kernel", "FRAGMENT")
    for name, schema in _SCHEMAS.items():
        lib.define(name + schema)
        lib.impl(name, _IMPLS[name], "CUDA")
        lib.impl(name, _META.get(name, lambda *a, **k: None), "Meta")
    _lib = lib  # keep the registrations alive
    # the all-reduce op set (torch_extension_rocm.cc:25-55) comes from C++: see csrc/torch/allreduce_ops.cc
    if not os.path.exists(COMMON_OPS):
        raise ImportError(f"{COMMON_OPS} is missing: run `python -m sglang_security_amd.build`")
    _load_libsglk()  # common_ops.so links against libsglk.so: have it mapped first
    torch.ops.load_library(COMMON_OPS)
