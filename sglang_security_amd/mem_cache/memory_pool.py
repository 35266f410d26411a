"""KV pools: host-side mirror of python/sglang/srt/mem_cache/memory_pool.py for the decode path.

  ReqToTokenPool            (:55-96)    request -> token-slot table, int32 [size, max_context_len]
  TokenToKVPoolAllocator    (:150-214)  free list of slot indices; slot 0 is a dummy for padded tokens
  MHATokenToKVPool          (:240-412)  per layer K [size + 1, Hkv, D] and V [size + 1, Hkv, Dv]; set_kv_buffer
                                        scatters new rows with the HIP kernel (sglk_set_kv_buffer)
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import ops


class ReqToTokenPool:
    def __init__(self, size: int, max_context_len: int, device: str, enable_memory_saver: bool = False):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)
        self.free_slots = list(range(size))

    def write(self, indices, values):
        self.req_to_token[indices] = values

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self.free_slots):
            return None
        select = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return select

    def free(self, free_index):
        if isinstance(free_index, int):
            self.free_slots.append(free_index)
        else:
            self.free_slots.extend(free_index)

    def clear(self):
        self.free_slots = list(range(self.size))


class TokenToKVPoolAllocator:
    def __init__(self, size: int, dtype: torch.dtype, device: str, kvcache: "MHATokenToKVPool"):
        self.size = size
        self.dtype = dtype
        self.device = device
        self.page_size = 1
        self._kvcache = kvcache
        self.free_group: List[torch.Tensor] = []
        self.is_not_in_free_group = True
        self.clear()

    def available_size(self) -> int:
        return len(self.free_slots)

    def get_kvcache(self):
        return self._kvcache

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if need_size > len(self.free_slots):
            return None
        select = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return select

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            self.free_slots = torch.cat((self.free_slots, free_index.to(self.free_slots.dtype)))
        else:
            self.free_group.append(free_index)

    def free_group_begin(self):
        self.is_not_in_free_group = False
        self.free_group = []

    def backup_state(self):
        return self.free_slots

    def restore_state(self, free_slots):
        self.free_slots = free_slots

    def free_group_end(self):
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(torch.cat(self.free_group))

    def clear(self):
        # slot 0 is reserved for writes of padded tokens (memory_pool.py:207-211)
        self.free_slots = torch.arange(1, self.size + 1, dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group = []


class MHATokenToKVPool:
    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int, layer_num: int,
                 device: str, enable_memory_saver: bool = False, start_layer: Optional[int] = None,
                 end_layer: Optional[int] = None, v_head_dim: Optional[int] = None):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        # an fp8 cache is kept as bytes and handed out as float8 views (memory_pool.py:119-123, 363-374)
        self.store_dtype = torch.uint8 if ops.kv_format_code(dtype) else dtype
        self.head_num = head_num
        self.head_dim = head_dim
        self.v_head_dim = v_head_dim or head_dim
        self.layer_num = layer_num
        self.device = device
        self.start_layer = start_layer or 0
        self.end_layer = end_layer if end_layer is not None else layer_num - 1
        # [size + page_size, head_num, head_dim] per layer; the padded slot 0 absorbs dummy writes
        self.k_buffer = [torch.zeros((size + page_size, head_num, head_dim), dtype=self.store_dtype, device=device)
                         for _ in range(layer_num)]
        self.v_buffer = [torch.zeros((size + page_size, head_num, self.v_head_dim), dtype=self.store_dtype, device=device)
                         for _ in range(layer_num)]

    def get_kv_size_bytes(self):
        k = sum(t.numel() * t.element_size() for t in self.k_buffer)
        v = sum(t.numel() * t.element_size() for t in self.v_buffer)
        return k, v

    def get_key_buffer(self, layer_id: int):
        b = self.k_buffer[layer_id - self.start_layer]
        return b.view(self.dtype) if self.store_dtype != self.dtype else b

    def get_value_buffer(self, layer_id: int):
        b = self.v_buffer[layer_id - self.start_layer]
        return b.view(self.dtype) if self.store_dtype != self.dtype else b

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor,
                      k_scale: Optional[float] = None, v_scale: Optional[float] = None):
        """memory_pool.py:378-412.  An fp8 pool divides the rows by the scales (if given) and rounds them to its
        format in one kernel; the reference applies the scales only when the rows' type differs from the pool's
        (:389-396), and so does this."""
        layer_id = layer.layer_id if hasattr(layer, "layer_id") else int(layer)
        kb, vb = self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)
        if cache_k.dtype != self.dtype:
            if self.store_dtype != self.dtype:
                ops.set_kv_buffer(kb, vb, loc, cache_k, cache_v, k_scale, v_scale)
                return
            if k_scale is not None:
                cache_k = cache_k / k_scale
            if v_scale is not None:
                cache_v = cache_v / v_scale
            cache_k = cache_k.to(self.dtype)
            cache_v = cache_v.to(self.dtype)
        ops.set_kv_buffer(kb, vb, loc, cache_k, cache_v)


class MLATokenToKVPool:
    """memory_pool.py:511-647: one latent row per token, kv_buffer [size + page, 1, kv_lora_rank + rope_dim]; keys are
    the whole row, values its first kv_lora_rank elements (a strided view: the decode kernel takes strides)."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, kv_lora_rank: int, qk_rope_head_dim: int,
                 layer_num: int, device: str, enable_memory_saver: bool = False, start_layer: Optional[int] = None,
                 end_layer: Optional[int] = None):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        # an fp8 cache is kept as bytes and handed out as float8 views (memory_pool.py:528-533, 576-593)
        self.store_dtype = torch.uint8 if ops.kv_format_code(dtype) else dtype
        self.kv_lora_rank = kv_lora_rank
        self.qk_rope_head_dim = qk_rope_head_dim
        self.layer_num = layer_num
        self.device = device
        self.start_layer = start_layer or 0
        self.end_layer = end_layer if end_layer is not None else layer_num - 1
        self.kv_buffer = [torch.zeros((size + page_size, 1, kv_lora_rank + qk_rope_head_dim), dtype=self.store_dtype,
                                      device=device) for _ in range(layer_num)]

    def get_kv_size_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.kv_buffer)

    def get_key_buffer(self, layer_id: int):
        b = self.kv_buffer[layer_id - self.start_layer]
        return b.view(self.dtype) if self.store_dtype != self.dtype else b

    def get_value_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id)[..., : self.kv_lora_rank]

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor):
        """memory_pool.py:595-611: cache_k is the whole latent row; cache_v is ignored (it is a view of the same data)."""
        layer_id = layer.layer_id if hasattr(layer, "layer_id") else int(layer)
        buf = self.get_key_buffer(layer_id)
        row = cache_k.reshape(loc.numel(), 1, -1)
        if self.store_dtype == self.dtype:
            row = row.to(self.dtype)
        ops.set_kv_buffer(buf, buf, loc, row, row)   # an fp8 pool rounds the rows as Tensor.to(float8) does

    def set_mla_kv_buffer(self, layer, loc: torch.Tensor, cache_k_nope: torch.Tensor, cache_k_rope: torch.Tensor):
        """memory_pool.py:613-625 (set_mla_kv_buffer_triton :446-508): the row is [nope | rope]."""
        row = torch.cat([cache_k_nope.reshape(loc.numel(), -1), cache_k_rope.reshape(loc.numel(), -1)], dim=-1)
        self.set_kv_buffer(layer, loc, row, row)
