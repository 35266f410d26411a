"""Tensor-level entry points over the C ABI: argument checks, pointer extraction, launch on torch's
current stream.  torch is plumbing here (device memory + streams); all arithmetic is in libsglk.so.

Function names and argument order follow the reference callers they replace (cited per function).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib
from ._lib import SglkError, check, dtype_code, lib


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise SglkError("sglk ops run on the GPU only; got a CPU tensor (there is no CPU fallback)")


def _is64(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return 1
    if t.dtype == torch.int32:
        return 0
    raise SglkError(f"index tensor must be int32 or int64, got {t.dtype}")


# ------------------------------------------------------------------------------------------- A0
def create_kv_indices(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr, kv_start_idx, kv_indices):
    """create_flashinfer_kv_indices_triton[(bs,)](...) -- layers/attention/utils.py:5-41."""
    _need_gpu(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr, kv_indices)
    if req_to_token.dtype != torch.int32 or kv_indices.dtype != torch.int32 or kv_indptr.dtype != torch.int32:
        raise SglkError("req_to_token, kv_indptr and kv_indices must be int32")
    bs = req_pool_indices.shape[0]
    check(lib().sglk_create_kv_indices(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _is64(req_pool_indices),
        _ptr(page_kernel_lens), _is64(page_kernel_lens), _ptr(kv_indptr), _ptr(kv_start_idx),
        0 if kv_start_idx is None else _is64(kv_start_idx), _ptr(kv_indices), bs, _stream(req_to_token)),
        "create_kv_indices")


def kv_indptr_cumsum(seq_lens, kv_indptr):
    _need_gpu(seq_lens, kv_indptr)
    check(lib().sglk_kv_indptr_cumsum(_ptr(seq_lens), _is64(seq_lens), _ptr(kv_indptr), seq_lens.shape[0],
                                      _stream(seq_lens)), "kv_indptr_cumsum")


def get_num_kv_splits(num_kv_splits, seq_lens, num_head, num_kv_head, max_kv_splits, device_core_count):
    """TritonAttnBackend.get_num_kv_splits -- triton_backend.py:235-269."""
    _need_gpu(num_kv_splits, seq_lens)
    num_token, num_seq = num_kv_splits.shape[0], seq_lens.shape[0]
    num_group = num_token // num_seq
    assert num_group * num_seq == num_token, f"num_seq({num_seq}), num_token({num_token}), something goes wrong!"
    check(lib().sglk_get_num_kv_splits(_ptr(num_kv_splits), _ptr(seq_lens), _is64(seq_lens), num_seq, num_group,
                                       num_head, num_kv_head, max_kv_splits, device_core_count,
                                       _stream(seq_lens)), "get_num_kv_splits")


def decode_metadata(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, num_kv_splits, num_head,
                    num_kv_head, max_kv_splits, device_core_count):
    _need_gpu(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, num_kv_splits)
    check(lib().sglk_decode_metadata(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _is64(req_pool_indices),
        _ptr(seq_lens), _is64(seq_lens), _ptr(kv_indptr), _ptr(kv_indices), _ptr(num_kv_splits),
        seq_lens.shape[0], num_head, num_kv_head, max_kv_splits, device_core_count, _stream(seq_lens)),
        "decode_metadata")


def plan_kv_heads(Hkv: int, head_dim: int) -> int:
    """The kv head count a plan is laid out for: with head size 64 the shared-prefix kernel walks PAIRS of neighbouring
    kv heads as one 128-wide head (an even number of kv heads), so a plan for it counts pairs."""
    return Hkv // 2 if (head_dim == 64 and Hkv % 2 == 0) else Hkv


def shared_prefix_pays(Hkv: int, head_dim: int, v_head_dim: int, itemsize: int) -> bool:
    """Whether sharing a prefix in the one-launch kernel is worth its longer dependency chain (prefix blocks ->
    counter -> merge): it trades that chain for fetching the prefix rows once per group, which pays when a token's K and
    V rows are large.  Measured at batch 64, 1024 shared + 128 private tokens: Llama-3-8B (4 KiB of K + V per token and
    layer) gains 12 % of the step, Qwen2-0.5B (512 bytes) loses 9 %; the line is drawn at 1 KiB.  Absorbed MLA
    (576 / 512) never: its plan-driven form is 13-33 % slower than the per-request kernels at 16 and at 128 heads (the
    shared rows come from L2 either way, and the remainders' extra 512-wide partials cost more than the fetches saved)."""
    if head_dim == 576 and v_head_dim == 512:
        return False
    return Hkv * (head_dim + v_head_dim) * itemsize >= 1024


def decode_shared_plan_buffer(B, Hq, Hkv, max_kv_splits, device, head_dim=128) -> torch.Tensor:
    """An int32 buffer for the per-step plan of the shared-prefix decode kernel (up to B requests)."""
    n = int(lib().sglk_decode_shared_plan_bytes(B, Hq, plan_kv_heads(Hkv, head_dim), max_kv_splits))
    if n <= 0:
        raise SglkError("decode_shared_plan_buffer: bad shape")
    return torch.zeros(n // 4, dtype=torch.int32, device=device)


def _plan_room(plan, B, Hq, Hkv, max_kv_splits, what):
    need = int(lib().sglk_decode_shared_plan_bytes(B, Hq, Hkv, max_kv_splits))
    if plan.dtype != torch.int32 or plan.numel() * 4 < need:
        raise SglkError(f"{what}: plan must be an int32 buffer of at least {need} bytes")


def decode_shared_plan(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, Hq, Hkv, max_kv_splits, plan,
                       min_share=0, head_dim=128):
    """Find the groups of requests that share their first slots and lay the step's work out for
    decode_attention_shared_planned (device work only: no host synchronisation, fixed addresses).  head_dim: the
    model's head size (64: the plan counts pairs of kv heads, see plan_kv_heads)."""
    _need_gpu(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, plan)
    B = seq_lens.shape[0]
    Hkv = plan_kv_heads(Hkv, head_dim)
    _plan_room(plan, B, Hq, Hkv, max_kv_splits, "decode_shared_plan")
    check(lib().sglk_decode_shared_plan(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _is64(req_pool_indices), _ptr(seq_lens),
        _is64(seq_lens), _ptr(kv_indptr), _ptr(kv_indices), B, Hq, Hkv, max_kv_splits, int(min_share), _ptr(plan),
        _stream(seq_lens)), "decode_shared_plan")


def decode_shared_plan_from_groups(kv_indptr, kv_indices, group_indptr, prefix_lens, B, Hq, Hkv, max_kv_splits, plan,
                                   head_dim=128):
    _need_gpu(kv_indptr, kv_indices, group_indptr, prefix_lens, plan)
    Hkv = plan_kv_heads(Hkv, head_dim)
    _plan_room(plan, B, Hq, Hkv, max_kv_splits, "decode_shared_plan_from_groups")
    check(lib().sglk_decode_shared_plan_from_groups(
        _ptr(kv_indptr), _ptr(kv_indices), _ptr(group_indptr), _ptr(prefix_lens), group_indptr.shape[0] - 1, B, Hq, Hkv,
        max_kv_splits, _ptr(plan), _stream(kv_indptr)), "decode_shared_plan_from_groups")


def decode_metadata_shared(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, num_kv_splits, num_head,
                           num_kv_head, max_kv_splits, device_core_count, plan, min_share=0, head_dim=128):
    """decode_metadata + decode_shared_plan in one call."""
    _need_gpu(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, num_kv_splits, plan)
    if plan_kv_heads(num_kv_head, head_dim) != num_kv_head:
        # the split counts follow the model's kv heads, the plan counts pairs of them
        decode_metadata(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, num_kv_splits, num_head,
                        num_kv_head, max_kv_splits, device_core_count)
        if seq_lens.shape[0]:
            decode_shared_plan(req_to_token, req_pool_indices, seq_lens, kv_indptr, kv_indices, num_head, num_kv_head,
                               max_kv_splits, plan, min_share, head_dim)
        return
    _plan_room(plan, seq_lens.shape[0], num_head, num_kv_head, max_kv_splits, "decode_metadata_shared")
    check(lib().sglk_decode_metadata_shared(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _is64(req_pool_indices),
        _ptr(seq_lens), _is64(seq_lens), _ptr(kv_indptr), _ptr(kv_indices), _ptr(num_kv_splits),
        seq_lens.shape[0], num_head, num_kv_head, max_kv_splits, device_core_count, int(min_share), _ptr(plan),
        _stream(seq_lens)), "decode_metadata_shared")


def decode_shared_plan_read(plan, B):
    """Host view of a plan's grouping (synchronises; tests and logging): dict with num_groups, n_items, order,
    group_indptr, prefix_lens."""
    h = plan[:16].cpu().tolist()
    G = h[1]
    base = 16
    order = plan[base: base + B].cpu().tolist()
    gi0 = base + B + 4 * B
    return dict(n_items=h[0], num_groups=G, order=order, group_indptr=plan[gi0: gi0 + G + 1].cpu().tolist(),
                prefix_lens=plan[gi0 + B + 1: gi0 + B + 1 + G].cpu().tolist())


def decode_attention_shared_planned(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, plan, workspace, max_kv_splits,
                                    sm_scale, logit_cap=0.0, xsum_out=None, num_kv_splits=None):
    """decode_attention_shared_prefix with the groups and the work layout taken from a per-step plan.  The launch
    serves every batch (requests that share nothing walk their whole context as remainders); num_kv_splits matters to
    the absorbed-MLA form only, whose per-request kernels are then launched beside it, the plan deciding on the device
    which of the two runs."""
    _need_gpu(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, plan, workspace)
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    need = decode_shared_workspace_bytes(B, Hq, Dv, max_kv_splits)
    if workspace.numel() * workspace.element_size() < need:
        raise SglkError(f"decode_attention_shared_planned: workspace too small ({need} bytes needed)")
    _plan_room(plan, B, Hq, plan_kv_heads(Hkv, D), max_kv_splits, "decode_attention_shared_planned")
    check(lib().sglk_decode_attention_shared_planned(
        _ptr(q), _ptr(k_buffer), _ptr(v_buffer), _ptr(o), _ptr(kv_indptr), _ptr(kv_indices), _ptr(plan), _ptr(workspace),
        max_kv_splits, B, Hq, Hkv, D, Dv, q.stride(0), q.stride(1), k_buffer.stride(0), k_buffer.stride(1),
        v_buffer.stride(0), v_buffer.stride(1), o.stride(0), o.stride(1), float(sm_scale), float(logit_cap),
        _attn_dtype(q, k_buffer, v_buffer), _ptr(xsum_out), 0 if xsum_out is None else xsum_out.stride(0),
        _ptr(num_kv_splits), _stream(q)), "decode_attention_shared_planned")


def decode_attention_shared_planned_rope(partials, splits, part_stride, bias, cos_sin_cache, positions, loc, q_out,
                                         k_buffer, v_buffer, o, kv_indptr, kv_indices, plan, launch_in_step, workspace,
                                         max_kv_splits, sm_scale, logit_cap=0.0, xsum_out=None):
    """rope_and_store_kv_partial + decode_attention_shared_planned in one launch (the rotation as the launch's first
    blocks).  q_out [B, Hq, 128] (any row stride: a view of the qkv buffer) receives the rotated query rows."""
    _need_gpu(partials, cos_sin_cache, positions, loc, q_out, k_buffer, v_buffer, o, kv_indptr, kv_indices, plan, workspace)
    _same_type_pool("decode_attention_shared_planned_rope", o, k_buffer, v_buffer)
    B, Hq, D = q_out.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    if q_out.stride(1) != D or q_out.stride(2) != 1:
        raise SglkError("decode_attention_shared_planned_rope: the query heads of a row must be contiguous")
    need = decode_shared_workspace_bytes(B, Hq, Dv, max_kv_splits)
    if workspace.numel() * workspace.element_size() < need:
        raise SglkError(f"decode_attention_shared_planned_rope: workspace too small ({need} bytes needed)")
    _plan_room(plan, B, Hq, Hkv, max_kv_splits, "decode_attention_shared_planned_rope")
    check(lib().sglk_decode_attention_shared_planned_rope(
        _ptr(partials), int(splits), int(part_stride), _ptr(bias), _ptr(cos_sin_cache), _ptr(positions),
        cos_sin_cache.shape[1], 0, _ptr(loc), _is64(loc), _ptr(q_out), q_out.stride(0), _ptr(k_buffer), _ptr(v_buffer),
        _ptr(o), _ptr(kv_indptr), _ptr(kv_indices), _ptr(plan), int(launch_in_step), _ptr(workspace), max_kv_splits, B, Hq,
        Hkv, D, Dv, k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1), o.stride(0), o.stride(1),
        float(sm_scale), float(logit_cap), dtype_code(o.dtype), _ptr(xsum_out),
        0 if xsum_out is None else xsum_out.stride(0), _stream(o)), "decode_attention_shared_planned_rope")


def prepare_for_decode(req_to_token, req_pool_indices, seq_lens, positions, out_cache_loc):
    """positions = seq_lens; out_cache_loc = req_to_token[req_pool_indices, seq_lens]; seq_lens += 1 (one launch)."""
    _need_gpu(req_to_token, req_pool_indices, seq_lens, positions, out_cache_loc)
    if positions.dtype != torch.int64:
        raise SglkError("prepare_for_decode: positions must be int64")
    check(lib().sglk_prepare_for_decode(_ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices),
                                        _is64(req_pool_indices), _ptr(seq_lens), _is64(seq_lens), _ptr(positions),
                                        _ptr(out_cache_loc), _is64(out_cache_loc), seq_lens.shape[0], _stream(seq_lens)),
          "prepare_for_decode")


def argmax(x, out):
    """out[r] = argmax(x[r]) (first maximum), int64."""
    _need_gpu(x, out)
    if x.dim() != 2 or x.stride(1) != 1 or out.dtype != torch.int64 or not out.is_contiguous():
        raise SglkError("argmax: x must be [rows, cols] with unit inner stride, out contiguous int64")
    check(lib().sglk_argmax(_ptr(x), _ptr(out), x.shape[0], x.shape[1], x.stride(0), dtype_code(x.dtype), _stream(x)),
          "argmax")


def common_prefix_len(req_to_token, req_pool_indices, seq_lens, out):
    _need_gpu(req_to_token, req_pool_indices, seq_lens, out)
    check(lib().sglk_common_prefix_len(_ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices),
                                       _is64(req_pool_indices), _ptr(seq_lens), _is64(seq_lens),
                                       seq_lens.shape[0], _ptr(out), _stream(seq_lens)), "common_prefix_len")


# ------------------------------------------------------------------------------------------- A1
# fp8 KV cache: pool tensors of these types (memory_pool.py:119-123 keeps them as uint8 and hands out float8 views)
_KV_FORMATS = {torch.float8_e4m3fn: 1, torch.float8_e5m2: 2, torch.float8_e5m2fnuz: 3}
_KV_SHIFT = 8


def kv_format_code(dtype) -> int:
    """SGLK_KV_* of a pool dtype, 0 when the pool holds the model's own float type."""
    return _KV_FORMATS.get(dtype, 0)


def _attn_dtype(q, k_buffer, v_buffer=None) -> int:
    """dtype argument of the attention entry points: the query type, plus the pool's fp8 format when it has one."""
    code = dtype_code(q.dtype)
    if k_buffer is None or k_buffer.dtype == q.dtype:
        return code
    f = kv_format_code(k_buffer.dtype)
    if f == 0 or (v_buffer is not None and v_buffer.dtype != k_buffer.dtype):
        raise SglkError(f"attention: pool dtype {k_buffer.dtype} with queries of {q.dtype}: the pool holds the query type "
                        f"or one of {sorted(str(t) for t in _KV_FORMATS)}")
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise SglkError("attention: an fp8 pool needs fp16 / bf16 queries")
    return code | (f << _KV_SHIFT)


def _same_type_pool(name, rows, k_buffer, v_buffer):
    """The fused rotation + pool-write kernels store rows in the rows' own type: an fp8 (or otherwise retyped) pool goes
    through rope + set_kv_buffer, which round to the pool's format."""
    if k_buffer.dtype != rows.dtype or v_buffer.dtype != rows.dtype:
        raise SglkError(f"{name}: the pools hold {k_buffer.dtype}, the rows {rows.dtype}; use rope followed by "
                        f"set_kv_buffer for a pool of another type")


def _last_dim_contig(t, name):
    if t.stride(-1) != 1:
        raise SglkError(f"{name}: last dimension must be contiguous")


def decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits,
                         max_kv_splits, sm_scale, logit_cap=0.0):
    """Same signature as triton_ops/decode_attention.py:682-733."""
    _need_gpu(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits)
    assert max_kv_splits == attn_logits.shape[2]
    assert q.shape[0] <= kv_indptr.shape[0] - 1
    assert q.shape[0] <= attn_logits.shape[0]
    for t, n in ((q, "q"), (k_buffer, "k_buffer"), (v_buffer, "v_buffer"), (o, "o")):
        _last_dim_contig(t, n)
    if q.dtype != o.dtype or k_buffer.dtype != v_buffer.dtype:
        raise SglkError("decode_attention: q and o share a dtype, k_buffer and v_buffer share a dtype")
    if attn_logits.dtype != torch.float32 or attn_lse.dtype != torch.float32 or not attn_logits.is_contiguous():
        raise SglkError("decode_attention: attn_logits / attn_lse must be contiguous fp32")
    if kv_indptr.dtype != torch.int32 or kv_indices.dtype != torch.int32 or num_kv_splits.dtype != torch.int32:
        raise SglkError("decode_attention: kv_indptr, kv_indices, num_kv_splits must be int32")
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    if attn_logits.shape[1] != Hq or attn_logits.shape[3] != Dv:
        raise SglkError("decode_attention: attn_logits must be [B, Hq, max_kv_splits, Dv]")
    check(lib().sglk_decode_attention(
        _ptr(q), _ptr(k_buffer), _ptr(v_buffer), _ptr(o), _ptr(kv_indptr), _ptr(kv_indices), _ptr(attn_logits),
        _ptr(attn_lse), _ptr(num_kv_splits), max_kv_splits, B, Hq, Hkv, D, Dv, q.stride(0), q.stride(1),
        k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1), o.stride(0), o.stride(1),
        float(sm_scale), float(logit_cap), _attn_dtype(q, k_buffer, v_buffer), _stream(q)), "decode_attention")


def decode_shared_workspace_bytes(B, Hq, Dv, max_kv_splits) -> int:
    return int(lib().sglk_decode_shared_workspace_bytes(B, Hq, Dv, max_kv_splits))


def decode_shared_workspace(B, Hq, Dv, max_kv_splits, device) -> torch.Tensor:
    """A workspace for decode_attention_shared_prefix with its ticket counters (the first B * Hq words) cleared; the
    kernels leave them cleared, so one tensor serves every later call that does not overlap another one in time."""
    ws = torch.empty(decode_shared_workspace_bytes(B, Hq, Dv, max_kv_splits) // 4, dtype=torch.float32, device=device)
    check(lib().sglk_decode_shared_workspace_init(_ptr(ws), B, Hq, torch.cuda.current_stream(ws.device).cuda_stream),
          "decode_shared_workspace_init")
    return ws


def decode_shared_workspace_status(ws, B, Hq) -> int:
    """1 if a launch on this workspace gave up waiting for its prefix partials (its outputs are NaN), else 0.
    Synchronises with the device."""
    return int(lib().sglk_decode_shared_workspace_status(_ptr(ws), B, Hq))


def decode_attention_shared_prefix(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, group_indptr, prefix_lens,
                                   workspace, max_kv_splits, sm_scale, logit_cap=0.0):
    _need_gpu(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, group_indptr, prefix_lens, workspace)
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    G = group_indptr.shape[0] - 1
    need = decode_shared_workspace_bytes(B, Hq, Dv, max_kv_splits)
    if workspace.numel() * workspace.element_size() < need:
        raise SglkError(f"decode_attention_shared_prefix: workspace too small ({need} bytes needed)")
    check(lib().sglk_decode_attention_shared_prefix(
        _ptr(q), _ptr(k_buffer), _ptr(v_buffer), _ptr(o), _ptr(kv_indptr), _ptr(kv_indices), _ptr(group_indptr),
        _ptr(prefix_lens), G, _ptr(workspace), max_kv_splits, B, Hq, Hkv, D, Dv, q.stride(0), q.stride(1),
        k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1), o.stride(0), o.stride(1),
        float(sm_scale), float(logit_cap), _attn_dtype(q, k_buffer, v_buffer), _stream(q)),
        "decode_attention_shared_prefix")


# ------------------------------------------------------------------------------------------- A2
def extend_attention_fwd(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr,
                         kv_indices, custom_mask, is_causal, mask_indptr, max_len_extend, sm_scale=None,
                         logit_cap=0.0, skip_prefix_custom_mask=True, sliding_window_size=-1, total_prefix=None):
    """Same signature as triton_ops/extend_attention.py:306-438.  With sliding_window_size > 0 the caller lists only the
    window's prefix tokens in kv_indices (triton_backend.py:75-103) and the reference's window mask applies to them.
    total_prefix (an addition): the number of valid entries of kv_indices (kv_indptr[-1], which the caller knows on the
    host); without it the size of kv_indices stands in, which overstates the prefix of an over-allocated index buffer
    and may pick the split + merge form where the plain kernel would do (same result, one launch more)."""
    _need_gpu(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices)
    T, Hq, D = q_extend.shape
    Hkv, Dv = v_extend.shape[1], v_extend.shape[2]
    if sm_scale is None:
        sm_scale = 1.0 / (D ** 0.5)
    B = qo_indptr.shape[0] - 1
    m8 = None
    if custom_mask is not None:
        _need_gpu(custom_mask, mask_indptr)
        m8 = custom_mask.view(torch.uint8) if custom_mask.dtype == torch.bool else custom_mask
        if m8.dtype != torch.uint8 or not m8.is_contiguous() or mask_indptr.dtype != torch.int64 or mask_indptr.numel() != B + 1:
            raise SglkError("extend_attention: custom_mask must be contiguous uint8 / bool, mask_indptr int64 [B + 1]")
    if sliding_window_size is not None and sliding_window_size > 0:
        check(lib().sglk_extend_attention_window(
            _ptr(q_extend), _ptr(k_extend), _ptr(v_extend), _ptr(o_extend), _ptr(k_buffer), _ptr(v_buffer),
            _ptr(qo_indptr), _ptr(kv_indptr), _ptr(kv_indices), _ptr(m8) if m8 is not None else None,
            _ptr(mask_indptr) if m8 is not None else None, 1 if skip_prefix_custom_mask else 0, int(sliding_window_size),
            B, Hq, Hkv, D, Dv, q_extend.stride(0), q_extend.stride(1), k_extend.stride(0), k_extend.stride(1),
            v_extend.stride(0), v_extend.stride(1), o_extend.stride(0), o_extend.stride(1),
            k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1),
            int(max_len_extend), float(sm_scale), float(logit_cap), 1 if is_causal else 0,
            _attn_dtype(q_extend, k_buffer, v_buffer), _stream(q_extend)), "extend_attention_window")
        return
    if custom_mask is not None:
        check(lib().sglk_extend_attention_masked(
            _ptr(q_extend), _ptr(k_extend), _ptr(v_extend), _ptr(o_extend), _ptr(k_buffer), _ptr(v_buffer),
            _ptr(qo_indptr), _ptr(kv_indptr), _ptr(kv_indices), _ptr(m8), _ptr(mask_indptr),
            1 if skip_prefix_custom_mask else 0, B, Hq, Hkv, D, Dv,
            q_extend.stride(0), q_extend.stride(1), k_extend.stride(0), k_extend.stride(1),
            v_extend.stride(0), v_extend.stride(1), o_extend.stride(0), o_extend.stride(1),
            k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1),
            int(max_len_extend), float(sm_scale), float(logit_cap), _attn_dtype(q_extend, k_buffer, v_buffer),
            _stream(q_extend)), "extend_attention_masked")
        return
    if total_prefix is None:
        total_prefix = kv_indices.numel() if kv_indptr.numel() > 1 else 0
    splits = extend_attention_split_plan(T, Hq, B, max_len_extend, int(total_prefix), D,
                                         q_extend.dtype) if (D == Dv and k_buffer is not None) else 1
    if splits > 1:
        return extend_attention_split(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr,
                                      kv_indices, max_len_extend, sm_scale, logit_cap, is_causal, splits)
    check(lib().sglk_extend_attention(
        _ptr(q_extend), _ptr(k_extend), _ptr(v_extend), _ptr(o_extend), _ptr(k_buffer), _ptr(v_buffer),
        _ptr(qo_indptr), _ptr(kv_indptr), _ptr(kv_indices), B, Hq, Hkv, D, Dv,
        q_extend.stride(0), q_extend.stride(1), k_extend.stride(0), k_extend.stride(1),
        v_extend.stride(0), v_extend.stride(1), o_extend.stride(0), o_extend.stride(1),
        k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1),
        int(max_len_extend), float(sm_scale), float(logit_cap), 1 if is_causal else 0,
        _attn_dtype(q_extend, k_buffer, v_buffer), _stream(q_extend)), "extend_attention")


def extend_attention_split_plan(T, Hq, B, max_len_extend, total_prefix, head_dim, dtype) -> int:
    """How many ways to split the cached prefix (1 = the plain kernel): only where one block per (sequence, query block,
    head) would leave most CUs without work and the prefixes are long enough to be worth a merge launch."""
    if head_dim != 128 or dtype not in (torch.float16, torch.bfloat16) or B <= 0 or total_prefix <= 0:
        return 1
    blocks = ((int(max_len_extend) + 63) // 64) * Hq * B
    avg_prefix = total_prefix // B
    if blocks > 128 or avg_prefix < 256:
        return 1
    return int(max(1, min(8, 512 // blocks, avg_prefix // 128)))


def extend_attention_split(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                           max_len_extend, sm_scale, logit_cap, is_causal, kv_splits, workspace=None):
    """extend_attention_fwd with the cached prefix split kv_splits ways and a merge launch (sglk_extend_attention_split)."""
    _need_gpu(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices)
    T, Hq, D = q_extend.shape
    Hkv, Dv = v_extend.shape[1], v_extend.shape[2]
    B = qo_indptr.shape[0] - 1
    need = int(lib().sglk_extend_attention_split_workspace_bytes(T, Hq, kv_splits))
    if workspace is None:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=q_extend.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise SglkError(f"extend_attention_split: workspace too small ({need} bytes needed)")
    check(lib().sglk_extend_attention_split(
        _ptr(q_extend), _ptr(k_extend), _ptr(v_extend), _ptr(o_extend), _ptr(k_buffer), _ptr(v_buffer),
        _ptr(qo_indptr), _ptr(kv_indptr), _ptr(kv_indices), B, Hq, Hkv, D, Dv,
        q_extend.stride(0), q_extend.stride(1), k_extend.stride(0), k_extend.stride(1),
        v_extend.stride(0), v_extend.stride(1), o_extend.stride(0), o_extend.stride(1),
        k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1),
        int(max_len_extend), float(sm_scale), float(logit_cap), 1 if is_causal else 0, int(kv_splits), _ptr(workspace), T,
        _attn_dtype(q_extend, k_buffer, v_buffer), _stream(q_extend)), "extend_attention_split")


# ------------------------------------------------------------------------------------------- A3
def merge_state(v_a, s_a, v_b, s_b, v_merged, s_merged):
    _need_gpu(v_a, s_a, v_b, s_b, v_merged)
    n, h, d = v_a.shape
    for t in (v_a, v_b, v_merged, s_a, s_b):
        if not t.is_contiguous():
            raise SglkError("merge_state: tensors must be contiguous")
    check(lib().sglk_merge_state(_ptr(v_a), _ptr(s_a), _ptr(v_b), _ptr(s_b), _ptr(v_merged), _ptr(s_merged), n, h, d,
                                 dtype_code(v_a.dtype), _stream(v_a)), "merge_state")


# ------------------------------------------------------------------------------------------- P1
def set_kv_buffer(k_buffer, v_buffer, loc, cache_k, cache_v, k_scale=None, v_scale=None):
    """k_buffer[loc] = cache_k; v_buffer[loc] = cache_v -- memory_pool.py:378-412.  Into an fp8 pool (float8 views or
    their uint8 storage with kv_format given by the float8 dtype) the rows are divided by the scales, if any, and
    rounded as Tensor.to(float8) rounds (:389-396)."""
    _need_gpu(k_buffer, v_buffer, loc, cache_k, cache_v)
    fmt = kv_format_code(k_buffer.dtype)
    if fmt and cache_k.dtype != k_buffer.dtype:
        if v_buffer.dtype != k_buffer.dtype or cache_v.dtype != cache_k.dtype:
            raise SglkError("set_kv_buffer: K and V pools (and K and V rows) must share a dtype")
        T = loc.shape[0]
        ck, cv = cache_k.reshape(T, -1), cache_v.reshape(T, -1)
        if ck.stride(1) != 1 or cv.stride(1) != 1:
            raise SglkError("set_kv_buffer: cache_k/cache_v rows must be contiguous")
        kb, vb = k_buffer.view(k_buffer.shape[0], -1), v_buffer.view(v_buffer.shape[0], -1)
        check(lib().sglk_set_kv_buffer_fp8(_ptr(kb), _ptr(vb), _ptr(ck), _ptr(cv), _ptr(loc), _is64(loc), T, ck.shape[1],
                                           cv.shape[1], kb.stride(0), vb.stride(0), ck.stride(0), cv.stride(0),
                                           float(k_scale) if k_scale is not None else 0.0,
                                           float(v_scale) if v_scale is not None else 0.0, dtype_code(cache_k.dtype), fmt,
                                           _stream(k_buffer)), "set_kv_buffer_fp8")
        return
    if k_scale is not None or v_scale is not None:
        raise SglkError("set_kv_buffer: scales belong to an fp8 pool")
    if cache_k.dtype != k_buffer.dtype:
        cache_k = cache_k.to(k_buffer.dtype)
        cache_v = cache_v.to(v_buffer.dtype)
    T = loc.shape[0]
    ck = cache_k.reshape(T, -1)
    cv = cache_v.reshape(T, -1)
    if ck.stride(1) != 1 or cv.stride(1) != 1:
        raise SglkError("set_kv_buffer: cache_k/cache_v rows must be contiguous")
    kb = k_buffer.view(k_buffer.shape[0], -1)
    vb = v_buffer.view(v_buffer.shape[0], -1)
    check(lib().sglk_set_kv_buffer(_ptr(kb), _ptr(vb), _ptr(ck), _ptr(cv), _ptr(loc), _is64(loc), T, ck.shape[1],
                                   cv.shape[1], kb.stride(0), vb.stride(0), ck.stride(0), cv.stride(0),
                                   k_buffer.element_size(), _stream(k_buffer)), "set_kv_buffer")


# ------------------------------------------------------------------------------------------- E1
def rmsnorm(out, x, weight, eps, gemma=False):
    _need_gpu(out, x, weight)
    x2 = x.view(-1, x.shape[-1])
    o2 = out.view(-1, out.shape[-1])
    if x2.stride(1) != 1 or o2.stride(1) != 1 or not weight.is_contiguous():
        raise SglkError("rmsnorm: rows must be contiguous")
    check(lib().sglk_rmsnorm(_ptr(o2), _ptr(x2), _ptr(weight), x2.shape[0], x2.shape[1], x2.stride(0), o2.stride(0),
                             float(eps), 1 if gemma else 0, dtype_code(x.dtype), _stream(x)), "rmsnorm")


def fused_add_rmsnorm(x, residual, weight, eps, gemma=False):
    _need_gpu(x, residual, weight)
    if not (x.is_contiguous() and residual.is_contiguous() and weight.is_contiguous()):
        raise SglkError("fused_add_rmsnorm: tensors must be contiguous")
    hidden = x.shape[-1]
    check(lib().sglk_fused_add_rmsnorm(_ptr(x), _ptr(residual), _ptr(weight), x.numel() // hidden, hidden,
                                       float(eps), 1 if gemma else 0, dtype_code(x.dtype), _stream(x)),
          "fused_add_rmsnorm")


def act_and_mul(out, x, act: int):
    _need_gpu(out, x)
    if not (x.is_contiguous() and out.is_contiguous()):
        raise SglkError("act_and_mul: tensors must be contiguous")
    d = x.shape[-1] // 2
    check(lib().sglk_act_and_mul(_ptr(out), _ptr(x), x.numel() // (2 * d), d, act, dtype_code(x.dtype), _stream(x)),
          "act_and_mul")


def rope(q, k, cos_sin_cache, positions, head_size, interleave):
    """q:[T,Hq*hs] (or [T,Hq,hs]) and k rotated in place."""
    _need_gpu(q, cos_sin_cache, positions)
    if positions.dtype != torch.int64:
        positions = positions.long()
    T = q.shape[0]
    q2 = q.view(T, -1)
    k2 = None if k is None else k.view(T, -1)
    check(lib().sglk_rope(_ptr(q2), _ptr(k2), _ptr(cos_sin_cache), _ptr(positions), T, q2.shape[1] // head_size,
                          0 if k2 is None else k2.shape[1] // head_size, head_size, cos_sin_cache.shape[1],
                          q2.stride(0), 0 if k2 is None else k2.stride(0), 1 if interleave else 0,
                          dtype_code(q.dtype), _stream(q)), "rope")


def rope_and_store_kv(qkv, cos_sin_cache, positions, k_buffer, v_buffer, loc, q_heads, kv_heads, head_size,
                      interleave=False):
    _need_gpu(qkv, cos_sin_cache, positions, k_buffer, v_buffer, loc)
    if not qkv.is_contiguous() or positions.dtype != torch.int64:
        raise SglkError("rope_and_store_kv: qkv must be contiguous and positions int64")
    _same_type_pool("rope_and_store_kv", qkv, k_buffer, v_buffer)
    kb = k_buffer.view(k_buffer.shape[0], -1)
    vb = v_buffer.view(v_buffer.shape[0], -1)
    check(lib().sglk_rope_and_store_kv(_ptr(qkv), _ptr(cos_sin_cache), _ptr(positions), _ptr(kb), _ptr(vb), _ptr(loc),
                                       _is64(loc), qkv.shape[0], q_heads, kv_heads, head_size,
                                       cos_sin_cache.shape[1], kb.stride(0), vb.stride(0), 1 if interleave else 0,
                                       dtype_code(qkv.dtype), _stream(qkv)), "rope_and_store_kv")


# ------------------------------------------------------------------------------------------- G1
def awq_dequantize(qweight, scales, qzeros):
    """-> [K, N] in scales.dtype; sgl-kernel/csrc/gemm/awq_kernel.cu:184-219."""
    _need_gpu(qweight, scales, qzeros)
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32:
        raise SglkError("awq_dequantize: qweight/qzeros must be int32")
    if not (qweight.is_contiguous() and scales.is_contiguous() and qzeros.is_contiguous()):
        raise SglkError("awq_dequantize: tensors must be contiguous")
    rows, qcols = qweight.shape
    group = rows // scales.shape[0]
    out = torch.empty((rows, qcols * 8), dtype=scales.dtype, device=scales.device)
    check(lib().sglk_awq_dequantize(_ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(out), rows, qcols, group,
                                    dtype_code(scales.dtype), _stream(qweight)), "awq_dequantize")
    return out


def awq_repack(qweight):
    _need_gpu(qweight)
    if qweight.dtype != torch.int32 or not qweight.is_contiguous():
        raise SglkError("awq_repack: qweight must be contiguous int32")
    packed = torch.empty_like(qweight)
    check(lib().sglk_awq_repack(_ptr(qweight), _ptr(packed), qweight.shape[0], qweight.shape[1], _stream(qweight)),
          "awq_repack")
    return packed


def gptq_repack(qweight, qzeros, group_size, zero_offset=1):
    """GPTQ qweight int32 [K/8, N] / qzeros int32 [K/G, N/8] -> (packed [K, N/8], AWQ-order zero words): the operands
    of awq_gemm / awq_dequantize_packed.  Raises if a zero point does not fit four bits."""
    _need_gpu(qweight, qzeros)
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32 or not (qweight.is_contiguous() and qzeros.is_contiguous()):
        raise SglkError("gptq_repack: qweight / qzeros must be contiguous int32")
    K, N = qweight.shape[0] * 8, qweight.shape[1]
    if tuple(qzeros.shape) != (K // group_size, N // 8):
        raise SglkError("gptq_repack: qzeros must be [K / group_size, N / 8]")
    packed = torch.empty((K, N // 8), dtype=torch.int32, device=qweight.device)
    zeros = torch.empty_like(qzeros)
    flag = torch.zeros(1, dtype=torch.int32, device=qweight.device)
    check(lib().sglk_gptq_repack(_ptr(qweight), _ptr(qzeros), _ptr(packed), _ptr(zeros), _ptr(flag), K, N, int(group_size),
                                 int(zero_offset), _stream(qweight)), "gptq_repack")
    if int(flag.item()):
        raise SglkError("gptq_repack: a zero point of 16 does not fit the 4-bit zero words (asymmetric v1 checkpoint)")
    return packed, zeros


def awq_dequantize_packed(packed, scales, qzeros, out=None):
    """awq_dequantize from the re-laid-out copy: [K, N] in the scale dtype (into `out` when given)."""
    _need_gpu(packed, scales, qzeros)
    K, N = packed.shape[0], packed.shape[1] * 8
    group = K // scales.shape[0]
    if out is None:
        out = torch.empty((K, N), dtype=scales.dtype, device=packed.device)
    elif out.shape != (K, N) or out.dtype != scales.dtype or not out.is_contiguous() or out.device != packed.device:
        raise SglkError("awq_dequantize_packed: out must be a contiguous [K, N] tensor of the scale dtype")
    check(lib().sglk_awq_dequantize_packed(_ptr(packed), _ptr(scales), _ptr(qzeros), _ptr(out), K, N, group,
                                           dtype_code(scales.dtype), _stream(packed)), "awq_dequantize_packed")
    return out


def awq_gemm_workspace_bytes(M, N, K) -> int:
    return int(lib().sglk_awq_gemm_workspace_bytes(M, N, K))


def awq_gemm(x, packed, scales, qzeros, bias=None, out=None, workspace=None):
    """x:[M,K] @ W -> [M,N]; packed from awq_repack(qweight)."""
    _need_gpu(x, packed, scales, qzeros)
    M, K = x.shape
    N = scales.shape[1]
    group = K // scales.shape[0]
    if x.dtype != scales.dtype:
        raise SglkError("awq_gemm: activation and scale dtypes differ")
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    need = awq_gemm_workspace_bytes(M, N, K)
    if workspace is None:
        # the head of the workspace holds split-K ticket counters: zero before first use, left zero by every call
        workspace = torch.zeros(need, dtype=torch.uint8, device=x.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise SglkError(f"awq_gemm: workspace too small ({need} bytes needed)")
    check(lib().sglk_awq_gemm(_ptr(x), _ptr(packed), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(out),
                              _ptr(workspace), M, N, K, group, x.stride(0), out.stride(0), dtype_code(x.dtype),
                              _stream(x)), "awq_gemm")
    return out


def awq_gemm_prefill(x, packed, scales, qzeros, bias=None, out=None):
    """Large-M x @ W for a packed AWQ weight, dequantised inside the GEMM (K % 128 == 0, N % 128 == 0)."""
    _need_gpu(x, packed, scales, qzeros)
    M, K = x.shape
    N = scales.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    check(lib().sglk_awq_gemm_prefill(_ptr(x), _ptr(packed), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(out), M, N, K,
                                      K // scales.shape[0], x.stride(0), out.stride(0), dtype_code(x.dtype), _stream(x)),
          "awq_gemm_prefill")
    return out


def dense_gemm_prefill(x, w, bias=None, out=None):
    """Large-M x:[M,K] @ w:[N,K]^T on the hand-written tile kernel (K % 128 == 0, N % 128 == 0)."""
    _need_gpu(x, w)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    check(lib().sglk_dense_gemm_prefill(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N, K, x.stride(0), w.stride(0),
                                        out.stride(0), dtype_code(x.dtype), _stream(x)), "dense_gemm_prefill")
    return out


def dense_repack(w):
    """[N, K] row-major 16-bit weight -> the streaming layout dense_gemm_packed reads (same shape and bytes)."""
    _need_gpu(w)
    N, K = w.shape
    packed = torch.empty(N, K, dtype=w.dtype, device=w.device)
    check(lib().sglk_dense_repack(_ptr(w), _ptr(packed), N, K, w.stride(0), _stream(w)), "dense_repack")
    return packed


def dense_gemm_packed(x, packed, bias=None, out=None):
    """x:[M,K] @ W^T for a weight laid out by dense_repack ([N, K] in shape, not in meaning)."""
    _need_gpu(x, packed)
    M, K = x.shape
    N = packed.shape[0]
    if packed.shape[1] != K or not packed.is_contiguous():
        raise SglkError("dense_gemm_packed: the packed weight must be a contiguous [N, K] tensor from dense_repack")
    if out is None:
        out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    check(lib().sglk_dense_gemm_packed(_ptr(x), _ptr(packed), _ptr(bias), _ptr(out), M, N, K, x.stride(0), out.stride(0),
                                       dtype_code(x.dtype), _stream(x)), "dense_gemm_packed")
    return out


def dense_gemm_partial_splits(M, N, K) -> int:
    """How many fp32 slabs dense_gemm_packed_partial leaves for an [M, K] x [N, K]^T product (0: shape not covered)."""
    return int(lib().sglk_dense_gemm_partial_splits(M, N, K))


def dense_gemm_packed_partial(x, packed, partial):
    """Split-K partial sums partial[split][M][N] (fp32) of x @ W^T for a dense_repack weight; the consumer kernel
    (fused_add_rmsnorm_ex, rope_and_store_kv_partial) adds the splits up."""
    _need_gpu(x, packed, partial)
    M, K = x.shape
    N = packed.shape[0]
    if packed.shape[1] != K or not packed.is_contiguous():
        raise SglkError("dense_gemm_packed_partial: the packed weight must be a contiguous [N, K] tensor from dense_repack")
    splits = dense_gemm_partial_splits(M, N, K)
    if splits <= 0 or partial.dtype != torch.float32 or partial.numel() < splits * M * N:
        raise SglkError("dense_gemm_packed_partial: shape not covered or partial buffer too small")
    check(lib().sglk_dense_gemm_packed_partial(_ptr(x), _ptr(packed), _ptr(partial), M, N, K, x.stride(0),
                                               dtype_code(x.dtype), _stream(x)), "dense_gemm_packed_partial")
    return splits


def dense_gemm_packed_ws(x, packed, bias, out, workspace):
    """dense_gemm_packed with K split over enough blocks to fill the chip (fp32 slabs in `workspace`, summed by a second
    launch); the unsplit form when the shape does not split."""
    _need_gpu(x, packed, out)
    M, K = x.shape
    N = packed.shape[0]
    if packed.shape[1] != K or not packed.is_contiguous():
        raise SglkError("dense_gemm_packed_ws: the packed weight must be a contiguous [N, K] tensor from dense_repack")
    check(lib().sglk_dense_gemm_packed_ws(_ptr(x), _ptr(packed), _ptr(bias), _ptr(out), M, N, K, x.stride(0), out.stride(0),
                                          _ptr(workspace), 0 if workspace is None else workspace.numel() * workspace.element_size(),
                                          dtype_code(x.dtype), _stream(x)), "dense_gemm_packed_ws")
    return out


def dense_gemm_packed_act(x, packed, out, act, workspace):
    """out[M, d] = act(x @ Wg^T) * (x @ Wu^T) for the merged [2d, K] gate_up weight in the dense_repack layout: the
    gated activation runs in the launch that sums the K splits (act: 0 silu, 1 gelu tanh, 2 gelu)."""
    _need_gpu(x, packed, out, workspace)
    M, K = x.shape
    N = packed.shape[0]
    if packed.shape[1] != K or not packed.is_contiguous() or out.shape != (M, N // 2):
        raise SglkError("dense_gemm_packed_act: packed is a contiguous [2d, K] tensor from dense_repack, out is [M, d]")
    check(lib().sglk_dense_gemm_packed_act(_ptr(x), _ptr(packed), _ptr(out), M, N, K, x.stride(0), out.stride(0), int(act),
                                           _ptr(workspace), workspace.numel() * workspace.element_size(),
                                           dtype_code(x.dtype), _stream(x)), "dense_gemm_packed_act")
    return out


def dense_gemm_workspace_bytes(M, N, K) -> int:
    return int(lib().sglk_dense_gemm_workspace_bytes(M, N, K))


def dense_gemm_nt(x, w, bias=None, out=None, workspace=None):
    """x:[M,K] @ w:[N,K]^T.  With a workspace (dense_gemm_workspace_bytes) narrow outputs are split along K."""
    _need_gpu(x, w)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if workspace is not None:
        _need_gpu(workspace)
        check(lib().sglk_dense_gemm_nt_ws(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N, K, x.stride(0), w.stride(0),
                                          out.stride(0), _ptr(workspace), workspace.numel() * workspace.element_size(),
                                          dtype_code(x.dtype), _stream(x)), "dense_gemm_nt")
        return out
    check(lib().sglk_dense_gemm_nt(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N, K, x.stride(0), w.stride(0),
                                   out.stride(0), dtype_code(x.dtype), _stream(x)), "dense_gemm_nt")
    return out


# ---- fp8 ----------------------------------------------------------------------------------------------
def per_token_quant_fp8(x, out_q, out_s):
    """sgl_per_token_quant_fp8(input, output_q, output_s) (sgl-kernel/python/sgl_kernel/gemm.py:131-136)."""
    _need_gpu(x, out_q, out_s)
    if not (x.is_contiguous() and out_q.is_contiguous() and out_s.is_contiguous()):
        raise SglkError("per_token_quant_fp8: tensors must be contiguous")
    if out_q.dtype != torch.float8_e4m3fn or out_s.dtype != torch.float32:
        raise SglkError("per_token_quant_fp8: output_q must be float8_e4m3fn and output_s float32")
    hidden = x.shape[-1]
    rows = x.numel() // hidden if hidden else 0
    check(lib().sglk_per_token_quant_fp8(_ptr(x), _ptr(out_q), _ptr(out_s), rows, hidden, dtype_code(x.dtype),
                                         _stream(x)), "per_token_quant_fp8")


def per_token_group_quant_fp8(x, out_q, out_s, group_size, eps, fp8_min, fp8_max):
    """sgl_per_token_group_quant_fp8 (sgl-kernel/python/sgl_kernel/gemm.py:84-95).  A column-major scale
    tensor (stride(0) == 1 with more than one row) is written through its strides."""
    _need_gpu(x, out_q, out_s)
    if not (x.is_contiguous() and out_q.is_contiguous()):
        raise SglkError("per_token_group_quant_fp8: input and output_q must be contiguous")
    if out_q.dtype != torch.float8_e4m3fn or out_s.dtype != torch.float32:
        raise SglkError("per_token_group_quant_fp8: output_q must be float8_e4m3fn and output_s float32")
    hidden = x.shape[-1]
    rows = x.numel() // hidden
    ng = hidden // group_size
    col_major = 0
    if out_s.dim() == 2 and rows > 1 and ng > 1 and out_s.stride(0) == 1 and out_s.stride(1) == rows:
        col_major = 1
    elif not out_s.is_contiguous():
        raise SglkError("per_token_group_quant_fp8: output_s must be row-major or exactly column-major")
    check(lib().sglk_per_token_group_quant_fp8(_ptr(x), _ptr(out_q), _ptr(out_s), rows, hidden, group_size, eps,
                                               fp8_min, fp8_max, col_major, dtype_code(x.dtype), _stream(x)),
          "per_token_group_quant_fp8")


def per_tensor_quant_fp8(x, out_q, out_s, is_static):
    """sgl_per_tensor_quant_fp8 (sgl-kernel/python/sgl_kernel/gemm.py:119-128); for the dynamic form output_s
    must hold 0 on entry (the kernel takes a running maximum into it)."""
    _need_gpu(x, out_q, out_s)
    if not (x.is_contiguous() and out_q.is_contiguous()):
        raise SglkError("per_tensor_quant_fp8: tensors must be contiguous")
    check(lib().sglk_per_tensor_quant_fp8(_ptr(x), _ptr(out_q), _ptr(out_s), x.numel(), 1 if is_static else 0,
                                          dtype_code(x.dtype), _stream(x)), "per_tensor_quant_fp8")


def per_tensor_quant_fp8_rows(x, out_q, out_s, is_static=False):
    """per_tensor_quant_fp8 of a strided view: x [..., cols] whose leading dimensions collapse to rows one pitch apart
    (e.g. q[..., :nope] of [B, H, nope + rope]), out_q the same shape with its own pitch; one scale for the view."""
    _need_gpu(x, out_q, out_s)
    cols = x.shape[-1]
    rows = x.numel() // max(cols, 1)

    def pitch(t):
        if t.stride(-1) != 1 and cols > 1:
            raise SglkError("per_tensor_quant_fp8_rows: the last dimension must be contiguous")
        if t.dim() == 1:
            return cols
        st = t.stride(-2)
        for d in range(t.dim() - 2):  # the leading dimensions must walk the rows at one pitch
            if t.shape[d] > 1 and t.stride(d) != st * math.prod(t.shape[d + 1:-1]):
                raise SglkError("per_tensor_quant_fp8_rows: rows must lie one pitch apart")
        return st

    if tuple(out_q.shape) != tuple(x.shape) or out_q.dtype != torch.float8_e4m3fn or out_s.dtype != torch.float32:
        raise SglkError("per_tensor_quant_fp8_rows: out_q must be float8_e4m3fn of x's shape and out_s float32")
    check(lib().sglk_per_tensor_quant_fp8_rows(_ptr(x), _ptr(out_q), _ptr(out_s), rows, cols, pitch(x), pitch(out_q),
                                               1 if is_static else 0, dtype_code(x.dtype), _stream(x)),
          "per_tensor_quant_fp8_rows")


def mla_rope_and_store(q_pe, q_out, latent, kv_buffer, loc, cos_sin_cache, positions, kv_lora_rank, rotary_dim=None,
                       is_neox=False):
    """Rotary step + cache write of the absorbed-MLA decode (models/deepseek_v2.py:1127-1193) in one launch: q_pe
    [B, H, rope] (any strides with a contiguous last dimension) rotated into q_out[..., kv_lora_rank:]; latent [B,
    kv_lora_rank + rope] = [k_nope | unrotated k_pe] written to kv_buffer rows loc with the rope part rotated."""
    _need_gpu(q_pe, q_out, latent, kv_buffer, loc, cos_sin_cache, positions)
    B, H, rope_dim = q_pe.shape
    kb = kv_buffer.view(kv_buffer.shape[0], -1)
    if q_pe.stride(2) != 1 or q_out.stride(2) != 1 or latent.stride(1) != 1 or positions.dtype != torch.int64:
        raise SglkError("mla_rope_and_store: last dimensions must be contiguous and positions int64")
    if (q_out.shape[0] != B or q_out.shape[1] != H or q_out.shape[2] < kv_lora_rank + rope_dim or latent.shape[0] != B
            or latent.shape[1] != kv_lora_rank + rope_dim or kb.shape[1] < kv_lora_rank + rope_dim or loc.numel() != B):
        raise SglkError("mla_rope_and_store: shapes do not fit")
    if not (q_pe.dtype == q_out.dtype == latent.dtype == kv_buffer.dtype) or cos_sin_cache.dtype != torch.float32:
        raise SglkError("mla_rope_and_store: one element type for q, the latent row and the pool; fp32 cos / sin")
    rd = rope_dim if rotary_dim is None else rotary_dim
    check(lib().sglk_mla_rope_and_store(_ptr(q_pe), _ptr(q_out), _ptr(latent), _ptr(kb), _ptr(loc), _is64(loc),
                                        _ptr(cos_sin_cache), _ptr(positions), B, H, kv_lora_rank, rope_dim, rd,
                                        q_pe.stride(0), q_pe.stride(1), q_out.stride(0), q_out.stride(1), latent.stride(0),
                                        kb.stride(0), cos_sin_cache.stride(0), 1 if is_neox else 0,
                                        dtype_code(q_pe.dtype), _stream(q_pe)), "mla_rope_and_store")


def _b_as_nk(mat_b, name):
    """The reference passes B as a column-major [K, N] view (mat_b.stride(0) == 1): the same memory as a
    row-major [N, K] matrix."""
    if mat_b.dim() != 2 or mat_b.stride(0) != 1:
        raise SglkError(f"{name}: mat_b must be a column-major [K, N] tensor")
    return mat_b.t()


def rmsnorm_quant_fp8(x, residual, weight, out_q, out_s, eps, gemma=False):
    """fused_add_rmsnorm (residual += x when given) + per-token fp8 quantisation of the normalised rows."""
    _need_gpu(x, weight, out_q, out_s)
    rows, hidden = x.shape
    if not (x.is_contiguous() and out_q.is_contiguous() and (residual is None or residual.is_contiguous())):
        raise SglkError("rmsnorm_quant_fp8: tensors must be contiguous")
    if out_q.dtype != torch.float8_e4m3fn or out_s.dtype != torch.float32 or out_s.numel() < rows:
        raise SglkError("rmsnorm_quant_fp8: out_q must be float8_e4m3fn and out_s float32 [rows]")
    check(lib().sglk_rmsnorm_quant_fp8(_ptr(x), _ptr(residual), _ptr(weight), _ptr(out_q), _ptr(out_s), rows, hidden,
                                       float(eps), 1 if gemma else 0, dtype_code(x.dtype), _stream(x)), "rmsnorm_quant_fp8")


def silu_and_mul_quant_fp8(x, out_q, out_s):
    """silu_and_mul over x = [gate | up] + per-token fp8 quantisation of the product rows."""
    _need_gpu(x, out_q, out_s)
    rows, two_d = x.shape
    if not (x.is_contiguous() and out_q.is_contiguous()):
        raise SglkError("silu_and_mul_quant_fp8: tensors must be contiguous")
    if out_q.dtype != torch.float8_e4m3fn or out_s.dtype != torch.float32 or out_s.numel() < rows:
        raise SglkError("silu_and_mul_quant_fp8: out_q must be float8_e4m3fn and out_s float32 [rows]")
    check(lib().sglk_silu_and_mul_quant_fp8(_ptr(x), _ptr(out_q), _ptr(out_s), rows, two_d // 2, dtype_code(x.dtype),
                                            _stream(x)), "silu_and_mul_quant_fp8")


def _group_quant_outputs(name, rows, cols, out_q, out_s):
    if cols % 128 != 0 or tuple(out_q.shape) != (rows, cols) or not out_q.is_contiguous() or \
            out_q.dtype != torch.float8_e4m3fn:
        raise SglkError(f"{name}: out_q must be a contiguous float8_e4m3fn [{rows},{cols}] with {cols} a multiple of 128")
    if tuple(out_s.shape) != (rows, cols // 128) or not out_s.is_contiguous() or out_s.dtype != torch.float32:
        raise SglkError(f"{name}: out_s must be a contiguous float32 [{rows},{cols // 128}]")


def rmsnorm_group_quant_fp8(x, residual, weight, out_q, out_s, eps, gemma=False):
    """fused_add_rmsnorm (residual += x when given; rmsnorm otherwise) + sgl_per_token_group_quant_fp8 (group 128,
    eps 1e-10, +-448) of the normalised rows, which are not written: out_q / out_s hold what the two reference calls
    would have produced.  x rows may lie a pitch apart (a column slice)."""
    _need_gpu(x, weight, out_q, out_s)
    rows, hidden = x.shape
    if x.stride(1) != 1 or (residual is not None and not residual.is_contiguous()) or not weight.is_contiguous():
        raise SglkError("rmsnorm_group_quant_fp8: rows must be contiguous")
    _group_quant_outputs("rmsnorm_group_quant_fp8", rows, hidden, out_q, out_s)
    check(lib().sglk_rmsnorm_group_quant_fp8(_ptr(x), _ptr(residual), _ptr(weight), _ptr(out_q), _ptr(out_s), rows, hidden,
                                             x.stride(0) if rows > 1 else hidden, float(eps), 1 if gemma else 0,
                                             dtype_code(x.dtype), _stream(x)), "rmsnorm_group_quant_fp8")


def silu_and_mul_group_quant_fp8(x, out_q, out_s):
    """silu_and_mul over x = [gate | up] + sgl_per_token_group_quant_fp8 (group 128) of the product rows."""
    _need_gpu(x, out_q, out_s)
    rows, two_d = x.shape
    if not x.is_contiguous():
        raise SglkError("silu_and_mul_group_quant_fp8: x must be contiguous")
    _group_quant_outputs("silu_and_mul_group_quant_fp8", rows, two_d // 2, out_q, out_s)
    check(lib().sglk_silu_and_mul_group_quant_fp8(_ptr(x), _ptr(out_q), _ptr(out_s), rows, two_d // 2, dtype_code(x.dtype),
                                                  _stream(x)), "silu_and_mul_group_quant_fp8")


def fp8_mm_workspace_bytes(M, N, K) -> int:
    return int(lib().sglk_fp8_mm_workspace_bytes(M, N, K))


def fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias=None, out=None, workspace=None):
    """fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias) (sgl-kernel/python/sgl_kernel/gemm.py:48-57);
    `out` (row-major [M, N] of out_dtype) and `workspace` (fp8_mm_workspace_bytes: narrow outputs are split along
    K) are additions for callers that own their buffers."""
    _need_gpu(mat_a, mat_b, scales_a, scales_b)
    if mat_a.dim() != 2 or mat_a.stride(1) != 1:
        raise SglkError("fp8_scaled_mm: mat_a must be a row-major 2-D tensor")
    b = _b_as_nk(mat_b, "fp8_scaled_mm")
    M, K = mat_a.shape
    N = b.shape[0]
    if b.shape[1] != K:
        raise SglkError(f"fp8_scaled_mm: inner dimensions differ ({K} vs {b.shape[1]})")
    if scales_a.numel() != M or scales_b.numel() != N:
        raise SglkError("fp8_scaled_mm: scales_a must have M and scales_b N elements")
    if bias is not None and (bias.numel() != N or bias.dtype != out_dtype):
        raise SglkError("fp8_scaled_mm: bias must have N elements of the output dtype")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=mat_a.device)
    elif tuple(out.shape) != (M, N) or out.dtype != out_dtype or out.stride(1) != 1:
        raise SglkError(f"fp8_scaled_mm: out must be a row-major [{M}, {N}] tensor of the output dtype")
    if workspace is not None:
        _need_gpu(workspace)
        check(lib().sglk_fp8_scaled_mm_ws(_ptr(mat_a), _ptr(b), _ptr(scales_a.contiguous()), _ptr(scales_b.contiguous()),
                                          _ptr(bias), _ptr(out), M, N, K, mat_a.stride(0), b.stride(0), out.stride(0),
                                          _ptr(workspace), workspace.numel() * workspace.element_size(),
                                          dtype_code(out_dtype), _stream(mat_a)), "fp8_scaled_mm")
        return out
    check(lib().sglk_fp8_scaled_mm(_ptr(mat_a), _ptr(b), _ptr(scales_a.contiguous()), _ptr(scales_b.contiguous()),
                                   _ptr(bias), _ptr(out), M, N, K, mat_a.stride(0), b.stride(0), out.stride(0),
                                   dtype_code(out_dtype), _stream(mat_a)), "fp8_scaled_mm")
    return out


def fp8_repack(w_nk):
    """An e4m3 weight [N, K] (row-major) in the sequential-streaming layout of fp8_scaled_mm_packed (same shape and
    byte count, not the same meaning)."""
    _need_gpu(w_nk)
    if w_nk.dim() != 2 or w_nk.stride(1) != 1 or w_nk.element_size() != 1:
        raise SglkError("fp8_repack: a row-major [N, K] fp8 tensor")
    N, K = w_nk.shape
    out = torch.empty((N, K), dtype=w_nk.dtype, device=w_nk.device)
    check(lib().sglk_fp8_repack(_ptr(w_nk), _ptr(out), N, K, w_nk.stride(0), _stream(w_nk)), "fp8_repack")
    return out


def fp8_mm_packed_workspace_bytes(M, N, K) -> int:
    return int(lib().sglk_fp8_mm_packed_workspace_bytes(M, N, K))


def fp8_scaled_mm_packed(mat_a, packed, scales_a, scales_b, out_dtype, bias=None, out=None, workspace=None):
    """fp8_scaled_mm on a weight from fp8_repack: mat_a [M, K] e4m3 row-major, packed [N, K]."""
    _need_gpu(mat_a, packed, scales_a, scales_b)
    M, K = mat_a.shape
    N = packed.shape[0]
    if packed.shape[1] != K or not packed.is_contiguous() or mat_a.stride(1) != 1:
        raise SglkError("fp8_scaled_mm_packed: mat_a row-major [M, K], packed a contiguous [N, K] tensor from fp8_repack")
    if scales_a.numel() != M or scales_b.numel() != N:
        raise SglkError("fp8_scaled_mm_packed: scales_a must have M and scales_b N elements")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=mat_a.device)
    elif tuple(out.shape) != (M, N) or out.dtype != out_dtype or out.stride(1) != 1:
        raise SglkError(f"fp8_scaled_mm_packed: out must be a row-major [{M}, {N}] tensor of the output dtype")
    check(lib().sglk_fp8_scaled_mm_packed(_ptr(mat_a), _ptr(packed), _ptr(scales_a.contiguous()), _ptr(scales_b.contiguous()),
                                          _ptr(bias), _ptr(out), M, N, K, mat_a.stride(0), out.stride(0), _ptr(workspace),
                                          0 if workspace is None else workspace.numel() * workspace.element_size(),
                                          dtype_code(out_dtype), _stream(mat_a)), "fp8_scaled_mm_packed")
    return out


def fp8_blockwise_scaled_mm_packed(mat_a, packed, scales_a, scales_b, out_dtype, workspace=None, out=None):
    """fp8_blockwise_scaled_mm on a weight from fp8_repack: mat_a [M, K] contiguous, packed [N, K], scales_a [M, K/128]
    (any strides), scales_b [K/128, N/128] as the reference passes it."""
    _need_gpu(mat_a, packed, scales_a, scales_b)
    M, K = mat_a.shape
    N = packed.shape[0]
    kb, nb = K // 128, (N + 127) // 128
    if not mat_a.is_contiguous() or packed.shape[1] != K or not packed.is_contiguous():
        raise SglkError("fp8_blockwise_scaled_mm_packed: mat_a contiguous [M, K], packed a contiguous [N, K] from fp8_repack")
    if tuple(scales_a.shape) != (M, kb) or tuple(scales_b.shape) != (kb, nb):
        raise SglkError(f"fp8_blockwise_scaled_mm_packed: scales_a must be [{M},{kb}] and scales_b [{kb},{nb}]")
    out = _mm_out(out, M, N, out_dtype, mat_a.device, "fp8_blockwise_scaled_mm_packed")
    check(lib().sglk_fp8_blockwise_scaled_mm_packed(
        _ptr(mat_a), _ptr(packed), _ptr(scales_a), _ptr(scales_b), _ptr(out), M, N, K, scales_a.stride(0), scales_a.stride(1),
        scales_b.stride(1), scales_b.stride(0), _ptr(workspace),
        0 if workspace is None else workspace.numel() * workspace.element_size(), dtype_code(out_dtype), _stream(mat_a)),
        "fp8_blockwise_scaled_mm_packed")
    return out


def fp8_mm_packed_partial_splits(M, N, K) -> int:
    return int(lib().sglk_fp8_mm_packed_partial_splits(M, N, K))


def fp8_mm_packed_partial(mat_a, packed, partial):
    """Raw split sums partial[split][M][N] (fp32) of mat_a @ W^T for a weight from fp8_repack; returns the slab count."""
    _need_gpu(mat_a, packed, partial)
    M, K = mat_a.shape
    N = packed.shape[0]
    splits = fp8_mm_packed_partial_splits(M, N, K)
    if (packed.shape[1] != K or not packed.is_contiguous() or mat_a.stride(1) != 1 or splits <= 0
            or partial.dtype != torch.float32 or partial.numel() < splits * M * N):
        raise SglkError("fp8_mm_packed_partial: shape not covered or partial buffer too small")
    check(lib().sglk_fp8_mm_packed_partial(_ptr(mat_a), _ptr(packed), _ptr(partial), M, N, K, mat_a.stride(0),
                                           _stream(mat_a)), "fp8_mm_packed_partial")
    return splits


def rmsnorm_quant_fp8_partial(partials, splits, row_scale, col_scale, residual, weight, out_q, out_s, eps, gemma=False,
                              bias=None):
    """rmsnorm_quant_fp8 on rows that arrive as the raw split sums of an fp8 product (partials [splits, rows, hidden],
    scaled by row_scale[row] * col_scale[col]); residual (updated in place) as there."""
    _need_gpu(partials, weight, out_q, out_s)
    rows, hidden = out_q.shape
    check(lib().sglk_rmsnorm_quant_fp8_partial(_ptr(partials), int(splits), rows * hidden, _ptr(row_scale), _ptr(col_scale),
                                               _ptr(bias), _ptr(residual), _ptr(weight), _ptr(out_q), _ptr(out_s), rows,
                                               hidden, float(eps), 1 if gemma else 0, dtype_code(weight.dtype),
                                               _stream(out_q)), "rmsnorm_quant_fp8_partial")


def silu_and_mul_quant_fp8_partial(partials, splits, row_scale, col_scale, out_q, out_s, dtype):
    """silu_and_mul_quant_fp8 on gate_up rows that arrive as the raw split sums of an fp8 product (partials
    [splits, rows, 2 d]); dtype: the 16-bit type that product would have stored its rows in."""
    _need_gpu(partials, out_q, out_s)
    rows, d = out_q.shape
    check(lib().sglk_silu_and_mul_quant_fp8_partial(_ptr(partials), int(splits), rows * 2 * d, _ptr(row_scale),
                                                    _ptr(col_scale), _ptr(out_q), _ptr(out_s), rows, d, dtype_code(dtype),
                                                    _stream(out_q)), "silu_and_mul_quant_fp8_partial")


def bmm_fp8(A, B, A_scale, B_scale, dtype, out=None):
    """bmm_fp8 (sgl-kernel/python/sgl_kernel/gemm.py:66-82): A [b, M, K] row-major, B [b, K, N] column-major (a
    transposed view of [b, N, K]), one fp32 scale per operand; e4m3 or e5m2 operands (not both e5m2)."""
    _need_gpu(A, B, A_scale, B_scale)
    if A.dim() != 3 or B.dim() != 3 or A.shape[0] != B.shape[0] or A.shape[2] != B.shape[1]:
        raise SglkError("bmm_fp8: A must be [b, M, K] and B [b, K, N]")
    if A.stride(2) != 1 or B.stride(1) != 1:
        raise SglkError("bmm_fp8: A must be row-major and B column-major")
    f8 = (torch.float8_e4m3fn, torch.float8_e5m2)
    if A.dtype not in f8 or B.dtype not in f8:
        raise SglkError("bmm_fp8: operands must be float8_e4m3fn / float8_e5m2")
    b, M, K = A.shape
    N = B.shape[2]
    if out is None:
        out = torch.empty((b, M, N), dtype=dtype, device=A.device)
    if out.stride(2) != 1:
        raise SglkError("bmm_fp8: out must be row-major")
    check(lib().sglk_bmm_fp8(_ptr(A), _ptr(B), _ptr(A_scale), _ptr(B_scale), _ptr(out), b, M, N, K, A.stride(0),
                             A.stride(1), B.stride(0), B.stride(2), out.stride(0), out.stride(1),
                             1 if A.dtype == torch.float8_e5m2 else 0, 1 if B.dtype == torch.float8_e5m2 else 0,
                             dtype_code(out.dtype), _stream(A)), "bmm_fp8")
    return out


def moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad,
                         token_cnts_buffer=None, cumsum_buffer=None):
    """moe_align_block_size (sgl-kernel/python/sgl_kernel/moe.py:4-23); token_cnts_buffer is not needed."""
    _need_gpu(topk_ids, sorted_token_ids, experts_ids, num_tokens_post_pad)
    if topk_ids.dtype not in (torch.int32, torch.int64) or not topk_ids.is_contiguous():
        raise SglkError("moe_align_block_size: topk_ids must be contiguous int32 / int64")
    for t in (sorted_token_ids, experts_ids, num_tokens_post_pad):
        if t.dtype != torch.int32:
            raise SglkError("moe_align_block_size: outputs must be int32")
    if cumsum_buffer is not None and (cumsum_buffer.dtype != torch.int32 or cumsum_buffer.numel() < num_experts + 1):
        raise SglkError("moe_align_block_size: cumsum_buffer must be int32 [num_experts + 1]")
    numel = topk_ids.numel()
    if sorted_token_ids.numel() < numel + num_experts * (block_size - 1):  # the bound fused_moe allocates
        raise SglkError("moe_align_block_size: sorted_token_ids smaller than numel + num_experts * (block_size - 1)")
    if experts_ids.numel() < (numel + num_experts * (block_size - 1) + block_size - 1) // block_size:
        raise SglkError("moe_align_block_size: experts_ids too small for the padded layout")
    check(lib().sglk_moe_align_block_size(_ptr(topk_ids), 1 if topk_ids.dtype == torch.int64 else 0, numel, num_experts,
                                          block_size, _ptr(sorted_token_ids), _ptr(experts_ids),
                                          _ptr(num_tokens_post_pad), _ptr(cumsum_buffer), _stream(topk_ids)),
          "moe_align_block_size")


def topk_softmax(topk_weights, topk_ids, token_expert_indices, gating_output):
    """topk_softmax (sgl-kernel/python/sgl_kernel/moe.py:26-34)."""
    _need_gpu(topk_weights, topk_ids, token_expert_indices, gating_output)
    if gating_output.dim() != 2 or gating_output.stride(1) != 1:
        raise SglkError("topk_softmax: gating_output must be [num_tokens, num_experts] with unit inner stride")
    T, E = gating_output.shape
    k = topk_weights.shape[-1]
    if topk_weights.dtype != torch.float32 or topk_ids.dtype != torch.int32 or token_expert_indices.dtype != torch.int32:
        raise SglkError("topk_softmax: topk_weights must be float32, the index outputs int32")
    for t in (topk_weights, topk_ids, token_expert_indices):
        if tuple(t.shape) != (T, k) or not t.is_contiguous():
            raise SglkError(f"topk_softmax: outputs must be contiguous [{T}, {k}]")
    check(lib().sglk_topk_softmax(_ptr(gating_output), _ptr(topk_weights), _ptr(topk_ids), _ptr(token_expert_indices), T,
                                  E, k, gating_output.stride(0), dtype_code(gating_output.dtype),
                                  _stream(gating_output)), "topk_softmax")


def _mm_out(out, M, N, out_dtype, device, name):
    """The [M, N] result tensor of a product: a fresh one, or the caller's (an addition: contiguous, of the type)."""
    if out is None:
        return torch.empty((M, N), dtype=out_dtype, device=device)
    if tuple(out.shape) != (M, N) or out.dtype != out_dtype or not out.is_contiguous() or out.device != device:
        raise SglkError(f"{name}: out must be a contiguous [{M},{N}] {out_dtype} tensor on the operands' device")
    return out


def fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, workspace=None, out=None):
    """fp8_blockwise_scaled_mm (sgl-kernel/python/sgl_kernel/gemm.py:36-45): scales_a [M, K/128] (any strides),
    scales_b [K/128, N/128] as the reference passes it (the transpose of the [N/128, K/128] checkpoint tensor)."""
    _need_gpu(mat_a, mat_b, scales_a, scales_b)
    if mat_a.dim() != 2 or not mat_a.is_contiguous():
        raise SglkError("fp8_blockwise_scaled_mm: mat_a must be a contiguous 2-D tensor")
    b = _b_as_nk(mat_b, "fp8_blockwise_scaled_mm")
    if not b.is_contiguous():
        raise SglkError("fp8_blockwise_scaled_mm: mat_b must be dense")
    M, K = mat_a.shape
    N = b.shape[0]
    kb = (K + 127) // 128
    nb = (N + 127) // 128
    if tuple(scales_a.shape) != (M, kb) or tuple(scales_b.shape) != (kb, nb):
        raise SglkError(f"fp8_blockwise_scaled_mm: scales_a must be [{M},{kb}] and scales_b [{kb},{nb}]")
    out = _mm_out(out, M, N, out_dtype, mat_a.device, "fp8_blockwise_scaled_mm")
    if workspace is not None:  # an addition: narrow outputs are split along K (fp8_mm_workspace_bytes)
        _need_gpu(workspace)
        check(lib().sglk_fp8_blockwise_scaled_mm_ws(_ptr(mat_a), _ptr(b), _ptr(scales_a), _ptr(scales_b), _ptr(out), M, N,
                                                    K, scales_a.stride(0), scales_a.stride(1), scales_b.stride(1),
                                                    scales_b.stride(0), _ptr(workspace),
                                                    workspace.numel() * workspace.element_size(), dtype_code(out_dtype),
                                                    _stream(mat_a)), "fp8_blockwise_scaled_mm")
        return out
    check(lib().sglk_fp8_blockwise_scaled_mm(_ptr(mat_a), _ptr(b), _ptr(scales_a), _ptr(scales_b), _ptr(out), M, N, K,
                                             scales_a.stride(0), scales_a.stride(1), scales_b.stride(1),
                                             scales_b.stride(0), dtype_code(out_dtype), _stream(mat_a)),
          "fp8_blockwise_scaled_mm")
    return out


# ---- fusion forms of the decode layer ---------------------------------------------------------------------
def awq_gemm_partial_splits(M, N, K, group=128) -> int:
    return int(lib().sglk_awq_gemm_partial_splits(M, N, K, group))


def awq_gemm_partial(x, packed, scales, qzeros, partial, xsum=None):
    """fp32 split-K partials [splits, M, N] of x @ W into `partial`; returns the number of splits written."""
    _need_gpu(x, packed, scales, qzeros, partial)
    M, K = x.shape
    N = scales.shape[1]
    group = K // scales.shape[0]
    splits = awq_gemm_partial_splits(M, N, K, group)
    if splits == 0:
        raise SglkError(f"awq_gemm_partial: shape M={M} N={N} K={K} group={group} is not covered")
    if partial.dtype != torch.float32 or partial.numel() < splits * M * N:
        raise SglkError(f"awq_gemm_partial: partial must hold {splits} x {M} x {N} fp32")
    check(lib().sglk_awq_gemm_partial(_ptr(x), _ptr(packed), _ptr(scales), _ptr(qzeros), _ptr(xsum),
                                      0 if xsum is None else xsum.stride(0), _ptr(partial), M, N, K, group, x.stride(0),
                                      dtype_code(x.dtype), _stream(x)), "awq_gemm_partial")
    return splits


def awq_gemm_xs(x, packed, scales, qzeros, xsum, out, workspace, bias=None):
    """awq_gemm with the block sums of x ([K/128, >= M] fp32) supplied by the producer of x."""
    _need_gpu(x, packed, scales, qzeros, out, workspace)
    M, K = x.shape
    N = scales.shape[1]
    check(lib().sglk_awq_gemm_xs(_ptr(x), _ptr(packed), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(out), _ptr(workspace),
                                 M, N, K, K // scales.shape[0], x.stride(0), out.stride(0), dtype_code(x.dtype), _ptr(xsum),
                                 0 if xsum is None else xsum.stride(0), _stream(x)), "awq_gemm_xs")
    return out


def awq_interleave_gate_up(qweight, scales, qzeros):
    """Column order sglk_awq_gemm_silu_mul expects: blocks of 64 gate columns alternate with the matching 64 up
    columns.  Inputs in the reference layout ([K, N/8] int32, [K/G, N], [K/G, N/8]) with N = 2 * intermediate size."""
    N = scales.shape[1]
    inter = N // 2
    if inter % 64:
        raise SglkError("awq_interleave_gate_up: intermediate size must be a multiple of 64")
    t = torch.arange(inter // 64, device=scales.device)
    blocks = torch.stack([t, t + inter // 64], dim=1).reshape(-1)  # 64-column block order: g0, u0, g1, u1, ...
    cols = (blocks[:, None] * 64 + torch.arange(64, device=scales.device)[None, :]).reshape(-1)
    packs = (blocks[:, None] * 8 + torch.arange(8, device=scales.device)[None, :]).reshape(-1)
    return (qweight.index_select(1, packs).contiguous(), scales.index_select(1, cols).contiguous(),
            qzeros.index_select(1, packs).contiguous())


def awq_gemm_silu_mul(x, packed, scales, qzeros, out, xsum=None, xsum_out=None):
    """out[M, N/2] = silu(x @ W_gate) * (x @ W_up) on a weight interleaved by awq_interleave_gate_up."""
    _need_gpu(x, packed, scales, qzeros, out)
    M, K = x.shape
    N = scales.shape[1]
    check(lib().sglk_awq_gemm_silu_mul(_ptr(x), _ptr(packed), _ptr(scales), _ptr(qzeros), _ptr(out), M, N, K,
                                       K // scales.shape[0], x.stride(0), out.stride(0), dtype_code(x.dtype), _ptr(xsum),
                                       0 if xsum is None else xsum.stride(0), _ptr(xsum_out),
                                       0 if xsum_out is None else xsum_out.stride(0), _stream(x)), "awq_gemm_silu_mul")
    return out


class _ChainJob(C.Structure):
    # sglk_chain_job (include/sglk.h)
    _fields_ = [("kind", C.c_int), ("N", C.c_int), ("K", C.c_int), ("splits_expected", C.c_int), ("x", C.c_void_p),
                ("x_stride", C.c_int64), ("packed", C.c_void_p), ("scales", C.c_void_p), ("qzeros", C.c_void_p),
                ("xsum_in", C.c_void_p), ("xs_stride", C.c_int64), ("out", C.c_void_p), ("out_stride", C.c_int64),
                ("partial", C.c_void_p), ("xsum_out", C.c_void_p), ("xs_out_stride", C.c_int64)]


def awq_chain_sync(device) -> torch.Tensor:
    """The counters of awq_chain (zero-filled; every launch leaves them zero)."""
    return torch.zeros(int(lib().sglk_awq_chain_sync_bytes()) // 4, dtype=torch.int32, device=device)


def awq_chain_status(sync) -> int:
    """1 if a block of a chain launched over `sync` gave up waiting for its producers (synchronises)."""
    return int(lib().sglk_awq_chain_status(_ptr(sync)))


def chain_partial(x, packed, scales, qzeros, xsum, partial):
    """A job of awq_chain: awq_gemm_partial(x, packed, scales, qzeros, partial, xsum)."""
    return dict(kind=0, x=x, packed=packed, scales=scales, qzeros=qzeros, xsum=xsum, partial=partial)


def chain_silu_mul(x, packed, scales, qzeros, xsum, out, xsum_out=None):
    """A job of awq_chain: awq_gemm_silu_mul(x, packed, scales, qzeros, out, xsum, xsum_out)."""
    return dict(kind=1, x=x, packed=packed, scales=scales, qzeros=qzeros, xsum=xsum, out=out, xsum_out=xsum_out)


def awq_chain(jobs, sync):
    """Several projections in one launch of persistent blocks, each multiplying what the one before it wrote (jobs:
    chain_partial / chain_silu_mul records, in order).  Same bits as the separate launches."""
    n = len(jobs)
    arr = (_ChainJob * n)()
    M = jobs[0]["x"].shape[0]
    group = None
    for i, j in enumerate(jobs):
        x, sc = j["x"], j["scales"]
        _need_gpu(x, j["packed"], sc, j["qzeros"], j["xsum"])
        if x.shape[0] != M:
            raise SglkError("awq_chain: every job multiplies the same rows")
        K, N = x.shape[1], sc.shape[1]
        g = K // sc.shape[0]
        if group is not None and g != group:
            raise SglkError("awq_chain: one group size per chain")
        group = g
        a = arr[i]
        a.kind, a.N, a.K, a.splits_expected = j["kind"], N, K, 0
        a.x, a.x_stride = _ptr(x), x.stride(0)
        a.packed, a.scales, a.qzeros = _ptr(j["packed"]), _ptr(sc), _ptr(j["qzeros"])
        a.xsum_in, a.xs_stride = _ptr(j["xsum"]), j["xsum"].stride(0)
        if j["kind"] == 0:
            part = j["partial"]
            splits = awq_gemm_partial_splits(M, N, K, g)
            if part.dtype != torch.float32 or part.numel() < splits * M * N:
                raise SglkError(f"awq_chain: job {i}: partial must hold {splits} x {M} x {N} fp32")
            a.partial, a.splits_expected = _ptr(part), splits
        else:
            out, xo = j["out"], j.get("xsum_out")
            _need_gpu(out)
            a.out, a.out_stride = _ptr(out), out.stride(0)
            a.xsum_out, a.xs_out_stride = _ptr(xo), 0 if xo is None else xo.stride(0)
    x0 = jobs[0]["x"]
    check(lib().sglk_awq_chain(arr, n, _ptr(sync), M, group, dtype_code(x0.dtype), _stream(x0)), "awq_chain")


def fused_add_rmsnorm_ex(x, residual, weight, eps, gemma=False, partials=None, splits=0, bias=None, xsum_out=None,
                         zero_out=None):
    """x (and residual) updated in place; the input row may come from `partials` [splits, rows, hidden] fp32."""
    _need_gpu(x, weight)
    hidden = x.shape[-1]
    rows = x.numel() // hidden
    check(lib().sglk_fused_add_rmsnorm_ex(_ptr(x), _ptr(residual), _ptr(weight), rows, hidden, float(eps),
                                          1 if gemma else 0, dtype_code(x.dtype), _ptr(partials), int(splits),
                                          rows * hidden, _ptr(bias), _ptr(xsum_out),
                                          0 if xsum_out is None else xsum_out.stride(0), _ptr(zero_out),
                                          0 if zero_out is None else zero_out.numel(), _stream(x)),
          "fused_add_rmsnorm_ex")


def act_and_mul_xs(out, x, act, xsum_out):
    _need_gpu(out, x, xsum_out)
    d = x.shape[-1] // 2
    check(lib().sglk_act_and_mul_xs(_ptr(out), _ptr(x), x.numel() // (2 * d), d, act, dtype_code(x.dtype), _ptr(xsum_out),
                                    xsum_out.stride(0), _stream(x)), "act_and_mul_xs")


def rope_and_store_kv_partial(qkv, partials, splits, cos_sin_cache, positions, k_buffer, v_buffer, loc, q_heads, kv_heads,
                              head_size, interleave=False, bias=None, row_scale=None, col_scale=None):
    """rope_and_store_kv on a qkv row that arrives as split-K partial sums; with row_scale / col_scale the sums are the
    raw ones of an fp8 product and are scaled as that product's own split sum scales them."""
    _need_gpu(qkv, partials, cos_sin_cache, positions, k_buffer, v_buffer, loc)
    _same_type_pool("rope_and_store_kv_partial", qkv, k_buffer, v_buffer)
    kb = k_buffer.view(k_buffer.shape[0], -1)
    vb = v_buffer.view(v_buffer.shape[0], -1)
    if row_scale is not None or col_scale is not None:
        _need_gpu(row_scale, col_scale)
        check(lib().sglk_rope_and_store_kv_partial_scaled(
            _ptr(qkv), _ptr(partials), int(splits), qkv.numel(), _ptr(bias), _ptr(row_scale), _ptr(col_scale),
            _ptr(cos_sin_cache), _ptr(positions), _ptr(kb), _ptr(vb), _ptr(loc), _is64(loc), qkv.shape[0], q_heads, kv_heads,
            head_size, cos_sin_cache.shape[1], kb.stride(0), vb.stride(0), 1 if interleave else 0, dtype_code(qkv.dtype),
            _stream(qkv)), "rope_and_store_kv_partial_scaled")
        return
    check(lib().sglk_rope_and_store_kv_partial(_ptr(qkv), _ptr(partials), int(splits), qkv.numel(), _ptr(bias),
                                               _ptr(cos_sin_cache), _ptr(positions), _ptr(kb), _ptr(vb), _ptr(loc),
                                               _is64(loc), qkv.shape[0], q_heads, kv_heads, head_size,
                                               cos_sin_cache.shape[1], kb.stride(0), vb.stride(0), 1 if interleave else 0,
                                               dtype_code(qkv.dtype), _stream(qkv)), "rope_and_store_kv_partial")


def decode_attention_fwd_xs(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits,
                            max_kv_splits, sm_scale, logit_cap, xsum_out):
    _need_gpu(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits)
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    check(lib().sglk_decode_attention_xs(
        _ptr(q), _ptr(k_buffer), _ptr(v_buffer), _ptr(o), _ptr(kv_indptr), _ptr(kv_indices), _ptr(attn_logits),
        _ptr(attn_lse), _ptr(num_kv_splits), max_kv_splits, B, Hq, Hkv, D, Dv, q.stride(0), q.stride(1),
        k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1), o.stride(0), o.stride(1),
        float(sm_scale), float(logit_cap), _attn_dtype(q, k_buffer, v_buffer), _ptr(xsum_out),
        0 if xsum_out is None else xsum_out.stride(0), _stream(q)), "decode_attention_xs")


def decode_attention_shared_prefix_xs(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, group_indptr, prefix_lens,
                                      workspace, max_kv_splits, sm_scale, logit_cap, xsum_out):
    _need_gpu(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, group_indptr, prefix_lens, workspace)
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    G = group_indptr.shape[0] - 1
    check(lib().sglk_decode_attention_shared_prefix_xs(
        _ptr(q), _ptr(k_buffer), _ptr(v_buffer), _ptr(o), _ptr(kv_indptr), _ptr(kv_indices), _ptr(group_indptr),
        _ptr(prefix_lens), G, _ptr(workspace), max_kv_splits, B, Hq, Hkv, D, Dv, q.stride(0), q.stride(1),
        k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1), o.stride(0), o.stride(1),
        float(sm_scale), float(logit_cap), _attn_dtype(q, k_buffer, v_buffer), _ptr(xsum_out),
        0 if xsum_out is None else xsum_out.stride(0), _stream(q)), "decode_attention_shared_prefix_xs")


def decode_attention_shared_prefix_rope(partials, splits, part_stride, bias, cos_sin_cache, positions, loc, k_buffer,
                                        v_buffer, o, kv_indptr, kv_indices, group_indptr, prefix_lens, workspace,
                                        max_kv_splits, sm_scale, logit_cap, xsum_out, interleave=False):
    """rope_and_store_kv_partial + decode_attention_shared_prefix_xs in one launch: q and the new K / V rows come from
    the qkv projection's split-K partial sums [splits][B][(Hq + 2 Hkv) * 128]; o is [B, Hq, 128]."""
    _need_gpu(partials, cos_sin_cache, positions, loc, k_buffer, v_buffer, o, kv_indptr, kv_indices, group_indptr,
              prefix_lens, workspace)
    _same_type_pool("decode_attention_shared_prefix_rope", o, k_buffer, v_buffer)
    B, Hq, D = o.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    G = group_indptr.shape[0] - 1
    check(lib().sglk_decode_attention_shared_prefix_rope(
        _ptr(partials), int(splits), int(part_stride), _ptr(bias), _ptr(cos_sin_cache), _ptr(positions),
        cos_sin_cache.shape[1], 1 if interleave else 0, _ptr(loc), _is64(loc), _ptr(k_buffer), _ptr(v_buffer), _ptr(o),
        _ptr(kv_indptr), _ptr(kv_indices), _ptr(group_indptr), _ptr(prefix_lens), G, _ptr(workspace), max_kv_splits, B, Hq,
        Hkv, D, Dv, k_buffer.stride(0), k_buffer.stride(1), v_buffer.stride(0), v_buffer.stride(1), o.stride(0),
        o.stride(1), float(sm_scale), float(logit_cap), dtype_code(o.dtype), _ptr(xsum_out),
        0 if xsum_out is None else xsum_out.stride(0), _stream(o)), "decode_attention_shared_prefix_rope")


def prefill_attention_paged(q, o, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, max_q_len, sm_scale,
                            logit_cap=0.0, causal=True):
    """Prefill over a paged pool that already holds the new tokens (kv_indices lists every token)."""
    _need_gpu(q, o, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices)
    T, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[-2], v_buffer.shape[-1]
    kb = k_buffer.reshape(-1, Hkv, D)
    vb = v_buffer.reshape(-1, Hkv, Dv)
    B = qo_indptr.shape[0] - 1
    check(lib().sglk_prefill_attention_paged(
        _ptr(q), _ptr(o), _ptr(kb), _ptr(vb), _ptr(qo_indptr), _ptr(kv_indptr), _ptr(kv_indices), B, Hq, Hkv, D, Dv,
        q.stride(0), q.stride(1), o.stride(0), o.stride(1), kb.stride(0), kb.stride(1), vb.stride(0), vb.stride(1),
        int(max_q_len), float(sm_scale), float(logit_cap), 1 if causal else 0, _attn_dtype(q, kb, vb), _stream(q)),
        "prefill_attention_paged")


# ---- sampling filters ---------------------------------------------------------------------------------------
def _probs2d(probs, name):
    _need_gpu(probs)
    if probs.dim() != 2 or probs.dtype != torch.float32 or probs.stride(1) != 1:
        raise SglkError(f"{name}: probs must be float32 [batch, vocab] with unit inner stride")
    return probs.shape


def _per_row(v, batch, dtype, name):
    """(tensor or None, scalar): a per-row tensor of the right dtype on the device, or the scalar for every row."""
    if isinstance(v, torch.Tensor):
        _need_gpu(v)
        t = v.to(dtype).contiguous()
        if t.numel() != batch:
            raise SglkError(f"{name}: per-row values must have batch elements")
        return t, 0
    return None, v


def top_k_renorm_probs(probs, top_k):
    B, V = _probs2d(probs, "top_k_renorm_probs")
    arr, val = _per_row(top_k, B, torch.int32, "top_k_renorm_probs")
    out = torch.empty_like(probs)
    check(lib().sglk_top_k_renorm_probs(_ptr(probs), _ptr(out), _ptr(arr), int(val), B, V, probs.stride(0), out.stride(0),
                                        _stream(probs)), "top_k_renorm_probs")
    return out


def top_p_renorm_probs(probs, top_p):
    B, V = _probs2d(probs, "top_p_renorm_probs")
    arr, val = _per_row(top_p, B, torch.float32, "top_p_renorm_probs")
    out = torch.empty_like(probs)
    check(lib().sglk_top_p_renorm_probs(_ptr(probs), _ptr(out), _ptr(arr), float(val), B, V, probs.stride(0),
                                        out.stride(0), _stream(probs)), "top_p_renorm_probs")
    return out


def sampling_from_probs(probs, uniform, top_k=0, top_p=1.0, min_p=0.0, joint=True):
    """int32 [batch] samples; uniform float32 [batch] in [0, 1)."""
    B, V = _probs2d(probs, "sampling_from_probs")
    _need_gpu(uniform)
    if uniform.dtype != torch.float32 or uniform.numel() != B or not uniform.is_contiguous():
        raise SglkError("sampling_from_probs: uniform must be contiguous float32 [batch]")
    ka, kv = _per_row(top_k, B, torch.int32, "sampling_from_probs")
    pa, pv = _per_row(top_p, B, torch.float32, "sampling_from_probs")
    ma, mv = _per_row(min_p, B, torch.float32, "sampling_from_probs")
    out = torch.empty(B, dtype=torch.int32, device=probs.device)
    check(lib().sglk_sampling_from_probs(_ptr(probs), _ptr(uniform), _ptr(out), _ptr(ka), int(kv), _ptr(pa), float(pv),
                                         _ptr(ma), float(mv), 1 if joint else 0, B, V, probs.stride(0), _stream(probs)),
          "sampling_from_probs")
    return out


# ------------------------------------------------------------------------------------------ draft-tree helpers
def _typed(t, dtype, shape_len, name, what):
    _need_gpu(t)
    if t.dtype != dtype or t.dim() != shape_len or not t.is_contiguous():
        raise SglkError(f"{name}: {what} must be a contiguous {dtype} tensor of {shape_len} dimension(s)")


def build_tree_kernel_efficient(parent_list, selected_index, verified_seq_len, tree_mask, positions, retrive_index,
                                retrive_next_token, retrive_next_sibling, topk, depth, draft_token_num):
    """In place, arguments as sgl_kernel.build_tree_kernel_efficient (sgl-kernel/python/sgl_kernel/speculative.py:61-86)."""
    name = "build_tree_kernel_efficient"
    bs = selected_index.shape[0]
    n = int(draft_token_num)
    _typed(selected_index, torch.int64, 2, name, "selected_index")
    _typed(verified_seq_len, torch.int32, 1, name, "verified_seq_len")
    _typed(tree_mask, torch.bool, 1, name, "tree_mask")
    _typed(positions, torch.int64, 1, name, "positions")
    for t, what in ((retrive_index, "retrive_index"), (retrive_next_token, "retrive_next_token"),
                    (retrive_next_sibling, "retrive_next_sibling")):
        _typed(t, torch.int64, 2, name, what)
        if tuple(t.shape) != (bs, n):
            raise SglkError(f"{name}: {what} must be [bs, draft_token_num]")
    if selected_index.shape[1] != n - 1 or verified_seq_len.numel() != bs or positions.numel() != bs * n:
        raise SglkError(f"{name}: selected_index [bs, n - 1], verified_seq_len [bs], positions [bs * n] expected")
    pitch = int(topk) * (int(depth) - 1) + 1
    if int(depth) > 1:
        _typed(parent_list, torch.int64, 2, name, "parent_list")
        if tuple(parent_list.shape) != (bs, pitch):
            raise SglkError(f"{name}: parent_list must be [bs, topk * (depth - 1) + 1]")
    if tree_mask.numel() < bs * n * n:
        raise SglkError(f"{name}: tree_mask holds {tree_mask.numel()} entries, at least bs * n * n needed")
    check(lib().sglk_build_tree_efficient(_ptr(parent_list) if int(depth) > 1 else None, _ptr(selected_index),
                                          _ptr(verified_seq_len), _ptr(tree_mask), _ptr(positions), _ptr(retrive_index),
                                          _ptr(retrive_next_token), _ptr(retrive_next_sibling), int(topk), int(depth), n,
                                          bs, tree_mask.numel(), _stream(selected_index)), name)


def _tree_walk_args(name, predicts, accept_index, accept_token_num, candidates, retrive_index, retrive_next_token,
                    retrive_next_sibling):
    _typed(predicts, torch.int32, 1, name, "predicts")
    _typed(accept_index, torch.int32, 2, name, "accept_index")
    _typed(accept_token_num, torch.int32, 1, name, "accept_token_num")
    _typed(candidates, torch.int32, 2, name, "candidates")
    bs, n = candidates.shape
    for t, what in ((retrive_index, "retrive_index"), (retrive_next_token, "retrive_next_token"),
                    (retrive_next_sibling, "retrive_next_sibling")):
        _typed(t, torch.int32, 2, name, what)
        if tuple(t.shape) != (bs, n):
            raise SglkError(f"{name}: {what} must match candidates [bs, num_draft_tokens]")
    if accept_index.shape[0] != bs or accept_token_num.shape[0] != bs:
        raise SglkError(f"{name}: accept_index / accept_token_num must have bs rows")
    return bs, n, accept_index.shape[1]


def verify_tree_greedy(predicts, accept_index, accept_token_num, candidates, retrive_index, retrive_next_token,
                       retrive_next_sibling, target_predict):
    """In place, arguments as sgl_kernel.verify_tree_greedy (speculative.py:38-58)."""
    name = "verify_tree_greedy"
    bs, n, steps = _tree_walk_args(name, predicts, accept_index, accept_token_num, candidates, retrive_index,
                                   retrive_next_token, retrive_next_sibling)
    _typed(target_predict, torch.int32, 2, name, "target_predict")
    if tuple(target_predict.shape) != (bs, n) or predicts.numel() != bs * n:
        raise SglkError(f"{name}: target_predict must be [bs, num_draft_tokens] and predicts hold bs * num_draft_tokens")
    check(lib().sglk_verify_tree_greedy(_ptr(predicts), _ptr(accept_index), _ptr(accept_token_num), _ptr(candidates),
                                        _ptr(retrive_index), _ptr(retrive_next_token), _ptr(retrive_next_sibling),
                                        _ptr(target_predict), bs, steps, n, predicts.numel(), _stream(candidates)), name)


def tree_speculative_sampling_target_only(predicts, accept_index, accept_token_num, candidates, retrive_index,
                                          retrive_next_token, retrive_next_sibling, uniform_samples, target_probs,
                                          draft_probs, threshold_single=1.0, threshold_acc=1.0, deterministic=True):
    """In place, arguments as sgl_kernel.tree_speculative_sampling_target_only (speculative.py:5-35); the kernel is
    deterministic in its inputs either way."""
    name = "tree_speculative_sampling_target_only"
    bs, n, steps = _tree_walk_args(name, predicts, accept_index, accept_token_num, candidates, retrive_index,
                                   retrive_next_token, retrive_next_sibling)
    _typed(uniform_samples, torch.float32, 2, name, "uniform_samples")
    _typed(target_probs, torch.float32, 3, name, "target_probs")
    _typed(draft_probs, torch.float32, 3, name, "draft_probs")
    if tuple(uniform_samples.shape) != (bs, n) or target_probs.shape[:2] != (bs, n) or draft_probs.shape != target_probs.shape:
        raise SglkError(f"{name}: uniform_samples [bs, n], target_probs / draft_probs [bs, n, vocab] expected")
    vocab = target_probs.shape[2]
    if predicts.numel() != bs * n:
        raise SglkError(f"{name}: predicts must hold bs * num_draft_tokens entries")
    check(lib().sglk_tree_speculative_sampling_target_only(
        _ptr(predicts), _ptr(accept_index), _ptr(accept_token_num), _ptr(candidates), _ptr(retrive_index),
        _ptr(retrive_next_token), _ptr(retrive_next_sibling), _ptr(uniform_samples), _ptr(target_probs), _ptr(draft_probs),
        bs, steps, n, predicts.numel(), vocab, float(threshold_single), float(threshold_acc), _stream(candidates)), name)


def segment_packbits(x, input_indptr, output_indptr, y):
    """In place, arguments as sgl_kernel.segment_packbits (speculative.py:89-101)."""
    name = "segment_packbits"
    _need_gpu(x, input_indptr, output_indptr, y)
    xb = x.view(torch.uint8) if x.dtype == torch.bool else x
    if xb.dtype != torch.uint8 or y.dtype != torch.uint8 or not (xb.is_contiguous() and y.is_contiguous()):
        raise SglkError(f"{name}: x must be contiguous bool / uint8 and y contiguous uint8")
    for t in (input_indptr, output_indptr):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise SglkError(f"{name}: indptr arrays must be contiguous int32")
    batch = input_indptr.numel() - 1
    if output_indptr.numel() != batch + 1:
        raise SglkError(f"{name}: output_indptr must have batch + 1 entries")
    check(lib().sglk_segment_packbits(_ptr(xb), _ptr(input_indptr), _ptr(output_indptr), _ptr(y), batch, _stream(x)), name)


def moe_fused_gate(input_tensor, bias, num_expert_group, topk_group, topk, n_share_experts_fusion=0,
                   routed_scaling_factor=0.0):
    """-> (weights fp32 [rows, topk], expert ids int32 [rows, topk]); sgl_kernel.moe_fused_gate (moe.py:37-62)."""
    name = "moe_fused_gate"
    _need_gpu(input_tensor, bias)
    if input_tensor.dim() != 2 or input_tensor.stride(1) != 1 or bias.dtype != input_tensor.dtype or not bias.is_contiguous():
        raise SglkError(f"{name}: input must be [rows, experts] with unit inner stride, bias [experts] of the same dtype")
    rows, E = input_tensor.shape
    if bias.numel() != E:
        raise SglkError(f"{name}: bias must have one entry per expert")
    out = torch.empty((rows, topk), dtype=torch.float32, device=input_tensor.device)
    idx = torch.empty((rows, topk), dtype=torch.int32, device=input_tensor.device)
    check(lib().sglk_moe_fused_gate(_ptr(input_tensor), _ptr(bias), _ptr(out), _ptr(idx), rows, E, int(num_expert_group),
                                    int(topk_group), int(topk), int(n_share_experts_fusion), float(routed_scaling_factor),
                                    input_tensor.stride(0), dtype_code(input_tensor.dtype), _stream(input_tensor)), name)
    return out, idx


_mla_scratch = {}


def decode_attention_fwd_grouped_rope(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, k_pe_tokens, kv_lora_rank,
                                      rotary_dim, cos_sin_cache, positions, attn_logits, num_kv_splits, sm_scale,
                                      logit_cap=0.0, use_rope=False, is_neox_style=False):
    """Same signature as triton_ops/rocm_mla_decode_rope.py:402-440.  With use_rope the rope part of q and of every
    request's newest pool row is rotated first (the pool row in place; see sglk_mla_rope_prepare), then the absorbed-MLA
    decode runs.  attn_logits only sizes the split count (dim 2); the kernels keep their own fp32 scratch."""
    _need_gpu(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, num_kv_splits)
    B, H, D = q.shape
    rope_dim = D - kv_lora_rank
    max_kv_splits = attn_logits.shape[2]
    Dv = v_buffer.shape[-1]
    key = (q.device, B, H, max_kv_splits, Dv, D, q.dtype)
    sc = _mla_scratch.get(key)
    if sc is None:
        sc = (torch.empty(B, H, max_kv_splits, Dv, dtype=torch.float32, device=q.device),
              torch.empty(B, H, max_kv_splits, dtype=torch.float32, device=q.device), torch.empty_like(q))
        _mla_scratch[key] = sc
    logits, lse, q_rot = sc
    if use_rope:
        _need_gpu(k_pe_tokens, cos_sin_cache, positions)
        cs = cos_sin_cache if cos_sin_cache.dtype == torch.float32 else cos_sin_cache.float()
        pos = positions if positions.dtype == torch.int64 else positions.long()
        kb = k_buffer.view(k_buffer.shape[0], -1)
        kpe = k_pe_tokens.view(B, -1)
        if kb.stride(1) != 1 or kpe.stride(1) != 1 or cs.stride(1) != 1 or q.stride(2) != 1:
            raise SglkError("decode_attention_fwd_grouped_rope: rows must be contiguous")
        check(lib().sglk_mla_rope_prepare(_ptr(q), _ptr(q_rot), _ptr(kb), _ptr(kpe), _ptr(kv_indptr), _ptr(kv_indices),
                                          _ptr(cs), _ptr(pos), B, H, int(kv_lora_rank), rope_dim, int(rotary_dim),
                                          q.stride(0), q.stride(1), q_rot.stride(0), q_rot.stride(1), kb.stride(0),
                                          kpe.stride(0), cs.stride(0), 1 if is_neox_style else 0, dtype_code(q.dtype),
                                          _stream(q)), "mla_rope_prepare")
        q = q_rot
    decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, logits, lse, num_kv_splits, max_kv_splits,
                         sm_scale, logit_cap)
