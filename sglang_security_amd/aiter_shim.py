"""The two `aiter` entry points the reference's default ROCm attention backend calls
(python/sglang/srt/layers/attention/aiter_backend.py:353-367 and :393-417), on the HIP kernels of libsglk.

    import sglang_security_amd.aiter_shim as shim; shim.install()     # sys.modules["aiter"] = this module

so that an unmodified `AiterAttnBackend` runs on these kernels.  The aiter library itself (v0.1.2 in the
reference's Dockerfile.rocm) is not part of the reference tree; the argument lists below are the reference's
call sites, the arithmetic is the same attention the Triton backend computes (parity tests use that oracle).
"""
from __future__ import annotations

import sys
from typing import Dict, Optional, Tuple

import torch

from . import ops
from ._lib import SglkError, lib

_MAX_KV_SPLITS = 8
_scratch: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}


def _decode_scratch(B: int, Hq: int, Dv: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    key = (B, Hq, Dv, str(device))
    if key not in _scratch:
        _scratch[key] = (torch.empty(B, Hq, _MAX_KV_SPLITS, Dv, dtype=torch.float32, device=device),
                         torch.empty(B, Hq, _MAX_KV_SPLITS, dtype=torch.float32, device=device),
                         torch.empty(B, dtype=torch.int32, device=device))
    return _scratch[key]


def paged_attention_ragged(o, workspace, q, k_cache, v_cache, scale, kv_indptr, kv_indices, kv_last_page_lens,
                           block_size, max_num_partitions, alibi_slopes=None, kv_cache_dtype="auto", layout="NHD",
                           logits_soft_cap=0.0, k_scale=None, v_scale=None, fp8_out_scale=None, partition_size=256):
    """Decode over a paged pool with page size 1: k_cache [N, 1, Hkv, D], v_cache [N, 1, Hkv, Dv]."""
    if block_size != 1:
        raise SglkError("paged_attention_ragged: only page size 1 (the reference's setting) is implemented")
    if alibi_slopes is not None or kv_cache_dtype != "auto" or layout != "NHD" or fp8_out_scale is not None:
        raise SglkError("paged_attention_ragged: alibi, fp8 KV cache and layouts other than NHD are not implemented")
    B, Hq, D = q.shape
    Hkv, Dv = v_cache.shape[-2], v_cache.shape[-1]
    kb = k_cache.view(-1, Hkv, D)
    vb = v_cache.view(-1, Hkv, Dv)
    logits, lse, splits = _decode_scratch(B, Hq, Dv, q.device)
    seq_lens = (kv_indptr[1:B + 1] - kv_indptr[:B]).to(torch.int32)
    ops.get_num_kv_splits(splits, seq_lens, Hq, Hkv, _MAX_KV_SPLITS, int(lib().sglk_device_cu_count()))
    ops.decode_attention_fwd(q, kb, vb, o, kv_indptr, kv_indices, logits, lse, splits, _MAX_KV_SPLITS, float(scale),
                             float(logits_soft_cap or 0.0))
    return o


def mha_batch_prefill_func(q, k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, max_q_len, max_kv_len, causal=True,
                           logits_soft_cap=0.0, alibi_slopes=None, return_lse=False, return_attn_probs=False,
                           softmax_scale: Optional[float] = None):
    """Prefill over the paged pool (the new tokens have been written to it already)."""
    if alibi_slopes is not None or return_lse or return_attn_probs:
        raise SglkError("mha_batch_prefill_func: alibi / returned statistics are not implemented")
    T, Hq, D = q.shape
    Dv = v_cache.shape[-1]
    o = torch.empty((T, Hq, Dv), dtype=q.dtype, device=q.device)
    scale = softmax_scale if softmax_scale is not None else D ** -0.5
    ops.prefill_attention_paged(q, o, k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, max_q_len, scale,
                                float(logits_soft_cap or 0.0), bool(causal))
    return o


def install(name: str = "aiter") -> None:
    sys.modules[name] = sys.modules[__name__]
