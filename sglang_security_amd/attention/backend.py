"""Attention backend on the HIP kernels: mirror of TritonAttnBackend
(python/sglang/srt/layers/attention/triton_backend.py:151-836) for the paths RadixAttention takes with a radix
cache: decode over paged KV, extend over a cached prefix, metadata at fixed addresses for graph replay.

Speculative batches are served as the reference serves them: a decode batch whose `spec_info` brings its own
kv_indptr / kv_indices (draft decode), target-verify batches (the draft tree's `custom_mask`, one row of
seq_len + num_draft_tokens mask bytes per draft token, `mask_indptr` built here) and draft-extend batches
(`spec_info.generate_attn_arg_prefill`), all through the masked extend kernel.

It reads the same fields of the objects the reference passes around:
  model_runner : .device, .req_to_token_pool.req_to_token, .token_to_kv_pool, .model_config.num_attention_heads,
                 .model_config.get_num_kv_heads(tp), .model_config.head_dim (v_head_dim optional), .tp_size,
                 .server_args.triton_attention_num_kv_splits
  forward_batch: .forward_mode.is_decode_or_idle() / is_target_verify() / is_draft_extend(), .batch_size,
                 .req_pool_indices, .seq_lens, .seq_lens_sum, .out_cache_loc, .extend_prefix_lens, .extend_seq_lens,
                 .token_to_kv_pool, .spec_info (.kv_indptr, .kv_indices, .custom_mask, .accept_length,
                 .generate_attn_arg_prefill)
  layer        : .layer_id, .tp_q_head_num, .tp_k_head_num, .qk_head_dim, .v_head_dim, .scaling, .logit_cap

Decode batches take the shared-prefix kernel without being told anything beyond what the reference backend is told:
the metadata call (eager, graph capture and graph replay alike) finds, on the device and at fixed addresses, the groups
of requests whose req_to_token rows start with the same pool slots -- the radix-tree hit RadixCache.match_prefix handed
them -- and lays the step's work out for the kernel, which reads a group's shared part of the KV pool once.  Requests
that share nothing walk their whole context in the same launch (ops.decode_attention_shared_planned).
The kernel covers head size 128 over 16-bit and fp8 pools, head size 64 over 16-bit pools and (as two launches)
absorbed MLA; left to itself the backend uses it where it measured faster (ops.shared_prefix_pays: a token's K and V
rows of at least 1 KiB per layer, not MLA -- elsewhere the per-request kernels, which find the shared rows in L2, win).  `share_prefix=True` (or SGLK_SHARE_PREFIX=1) uses
it wherever it can run, `share_prefix=False` (or SGLK_SHARE_PREFIX=0) keeps every batch on the per-request kernels.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from .. import ops
from .._lib import SglkError, lib


@dataclass
class ForwardMetadata:
    """triton_backend.py:134-148."""
    attn_logits: Optional[torch.Tensor]
    attn_lse: Optional[torch.Tensor]
    max_extend_len: Optional[int]
    num_kv_splits: Optional[torch.Tensor]
    kv_indptr: torch.Tensor
    kv_indices: torch.Tensor
    qo_indptr: Optional[torch.Tensor]
    custom_mask: Optional[torch.Tensor] = None
    mask_indptr: Optional[torch.Tensor] = None
    shared_plan: Optional[torch.Tensor] = None       # the step's plan of the shared-prefix kernel (device resident)
    shared_workspace: Optional[torch.Tensor] = None
    total_prefix: Optional[int] = None               # extend: entries of kv_indices in use
    # sliding-window layers (triton_backend.py:146-148)
    window_kv_indptr: Optional[torch.Tensor] = None
    window_kv_indices: Optional[torch.Tensor] = None
    window_num_kv_splits: Optional[torch.Tensor] = None


class SglkAttnBackend:
    def __init__(self, model_runner, skip_prefill: bool = False, kv_indptr_buf: Optional[torch.Tensor] = None,
                 share_prefix: Optional[bool] = None):
        self.device = model_runner.device
        # None: share where it pays (ops.shared_prefix_pays); True / SGLK_SHARE_PREFIX=1: wherever the kernel can;
        # False / SGLK_SHARE_PREFIX=0: never
        env = os.environ.get("SGLK_SHARE_PREFIX")
        if share_prefix is None and env is not None:
            share_prefix = env != "0"
        self.share_auto = share_prefix is None
        self.share_prefix = True if share_prefix is None else bool(share_prefix)
        self.kv_pool = getattr(model_runner, "token_to_kv_pool", None)
        self._pool_kind = None  # found on first use: "16" (fp16 / bf16 pool), "8" (an fp8 format) or "other"
        self.req_to_token = model_runner.req_to_token_pool.req_to_token
        max_bs = self.req_to_token.shape[0]
        self.kv_indptr = kv_indptr_buf if kv_indptr_buf is not None else torch.zeros(
            (max_bs + 1,), dtype=torch.int32, device=self.device)
        self.qo_indptr = torch.zeros((max_bs + 1,), dtype=torch.int32, device=self.device)
        self.mask_indptr = torch.zeros((max_bs + 1,), dtype=torch.int64, device=self.device)  # triton_backend.py:208-210
        self.num_draft_tokens = getattr(model_runner.server_args, "speculative_num_draft_tokens", None)
        cfg = model_runner.model_config
        tp = getattr(model_runner, "tp_size", 1)
        self.num_head = cfg.num_attention_heads // tp
        self.num_kv_head = cfg.get_num_kv_heads(tp)
        self.head_dim = cfg.head_dim
        self.v_head_dim = getattr(cfg, "v_head_dim", None) or cfg.head_dim
        self.max_kv_splits = getattr(model_runner.server_args, "triton_attention_num_kv_splits", 8)
        self.static_kv_splits = False
        self.device_core_count = int(lib().sglk_device_cu_count())
        self.max_context_len = self.req_to_token.shape[1]
        self.forward_metadata: Optional[ForwardMetadata] = None
        # triton_backend.py:175-199: models with sliding-window layers list, beside the full sequences, the last
        # window + 1 tokens of every request
        sw = getattr(model_runner, "sliding_window_size", None)
        self.sliding_window_size = int(sw) if sw is not None and sw > 0 else None
        self.window_kv_indptr = torch.zeros_like(self.kv_indptr) if self.sliding_window_size else None

    # ---- metadata ------------------------------------------------------------------------------------------
    def get_num_kv_splits(self, num_kv_splits: torch.Tensor, seq_lens: torch.Tensor):
        """triton_backend.py:235-269."""
        if self.static_kv_splits or self.device_core_count <= 0:
            num_kv_splits.fill_(self.max_kv_splits)
            return
        ops.get_num_kv_splits(num_kv_splits, seq_lens, self.num_head, self.num_kv_head, self.max_kv_splits,
                              self.device_core_count)

    def _decode_buffers(self, bs: int):
        attn_logits = torch.empty((bs, self.num_head, self.max_kv_splits, self.v_head_dim), dtype=torch.float32,
                                  device=self.device)
        attn_lse = torch.empty((bs, self.num_head, self.max_kv_splits), dtype=torch.float32, device=self.device)
        num_kv_splits = torch.empty((bs,), dtype=torch.int32, device=self.device)
        return attn_logits, attn_lse, num_kv_splits

    def _window_lists(self, lens: torch.Tensor, req_pool_indices: torch.Tensor, bs: int, indices_out=None):
        """update_sliding_window_buffer (triton_backend.py:75-131): the last min(len, window + 1) slots of every
        request.  Returns (window_kv_indptr[: bs + 1], window_kv_indices, window_kv_lens)."""
        lens = lens[:bs]
        wlens = torch.clamp(lens, max=self.sliding_window_size + 1)
        ops.kv_indptr_cumsum(wlens, self.window_kv_indptr)
        indptr = self.window_kv_indptr[: bs + 1]
        if indices_out is None:
            indices_out = torch.empty(max(bs * (self.sliding_window_size + 1), 1), dtype=torch.int32, device=self.device)
        ops.create_kv_indices(self.req_to_token, req_pool_indices[:bs], wlens, indptr, lens - wlens, indices_out)
        return indptr, indices_out, wlens

    def _shared_workspace(self, bs: int):
        return ops.decode_shared_workspace(bs, self.num_head, self.v_head_dim, self.max_kv_splits, self.device)

    def _shares(self, bs: int) -> bool:
        """Whether decode batches of bs requests go through the plan + shared-prefix kernel.  The one-launch kernel
        takes head size 128 over 16-bit and fp8 pools, head size 64 (an even number of kv heads, 16-bit pools) as
        pairs of kv heads, and absorbed MLA (576 / 512, 16-bit pool) in a two-launch form; a plan covers up to 1024
        requests; 16 splits at most."""
        if self._pool_kind is None:
            try:
                dt = self.kv_pool.get_key_buffer(0).dtype
                self._pool_kind = ("16" if dt in (torch.float16, torch.bfloat16)
                                   else "8" if ops.kv_format_code(dt) else "other")
            except Exception:
                self._pool_kind = "other"
        shape_ok = ((self.head_dim == 128 and self.v_head_dim == 128 and self._pool_kind in ("16", "8"))
                    or (self.head_dim == 64 and self.v_head_dim == 64 and self.num_kv_head % 2 == 0
                        and self._pool_kind == "16")
                    or (self.head_dim == 576 and self.v_head_dim == 512 and self.num_kv_head == 1
                        and self._pool_kind == "16"))
        if shape_ok and self.share_auto:
            shape_ok = ops.shared_prefix_pays(self.num_kv_head, self.head_dim, self.v_head_dim,
                                              1 if self._pool_kind == "8" else 2)
        return (self.share_prefix and shape_ok and 0 < bs <= 1024 and self.max_kv_splits <= 16
                and not self.sliding_window_size)

    def _plan_buffer(self, max_bs: int):
        """Room for the plan of any batch of up to max_bs requests (the size is not monotonic in the batch size)."""
        hkv = ops.plan_kv_heads(self.num_kv_head, self.head_dim)
        need = max(int(lib().sglk_decode_shared_plan_bytes(b, self.num_head, hkv, self.max_kv_splits))
                   for b in range(1, max_bs + 1))
        return torch.zeros(need // 4, dtype=torch.int32, device=self.device)

    def init_forward_metadata(self, forward_batch):
        """triton_backend.py:271-446."""
        bs = forward_batch.batch_size
        kv_indptr = self.kv_indptr
        spec_info = getattr(forward_batch, "spec_info", None)
        mode = forward_batch.forward_mode
        is_mode = lambda name: bool(getattr(mode, name, lambda: False)())
        if mode.is_decode_or_idle():
            plan = None
            if spec_info is None:
                ops.kv_indptr_cumsum(forward_batch.seq_lens, kv_indptr)
                kv_indptr = kv_indptr[: bs + 1]
                kv_indices = torch.empty(forward_batch.seq_lens_sum, dtype=torch.int32, device=self.device)
                ops.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens, kv_indptr,
                                      None, kv_indices)
            else:  # draft decode: topk rows per request, indices prepared by the draft worker (:317-319)
                kv_indptr, kv_indices = spec_info.kv_indptr, spec_info.kv_indices
                bs = kv_indptr.shape[0] - 1
            attn_logits, attn_lse, num_kv_splits = self._decode_buffers(bs)
            self.get_num_kv_splits(num_kv_splits, forward_batch.seq_lens)
            if spec_info is None and self._shares(bs):
                plan = ops.decode_shared_plan_buffer(bs, self.num_head, self.num_kv_head, self.max_kv_splits, self.device,
                                                     head_dim=self.head_dim)
                ops.decode_shared_plan(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens, kv_indptr,
                                       kv_indices, self.num_head, self.num_kv_head, self.max_kv_splits, plan,
                                       head_dim=self.head_dim)
            self.forward_metadata = ForwardMetadata(attn_logits, attn_lse, None, num_kv_splits, kv_indptr, kv_indices,
                                                    None, shared_plan=plan,
                                                    shared_workspace=self._shared_workspace(bs) if plan is not None else None)
            if self.sliding_window_size and spec_info is None:  # :297-316
                wp, wi, wl = self._window_lists(forward_batch.seq_lens, forward_batch.req_pool_indices, bs)
                wsplits = torch.empty((bs,), dtype=torch.int32, device=self.device)
                self.get_num_kv_splits(wsplits, wl)
                md = self.forward_metadata
                md.window_kv_indptr, md.window_kv_indices, md.window_num_kv_splits = wp, wi, wsplits
            return
        if is_mode("is_target_verify"):
            # :338-374: every request scores num_draft_tokens tree nodes against its whole cached sequence; the tree's
            # ancestor mask holds, per draft token, one byte per cached token and per draft token
            if spec_info is None or self.num_draft_tokens is None:
                raise SglkError("target-verify batch without spec_info / speculative_num_draft_tokens")
            bs = len(forward_batch.req_pool_indices)
            nd = int(self.num_draft_tokens)
            qo_indptr = torch.arange(0, (1 + bs) * nd, step=nd, dtype=torch.int32, device=self.device)
            ops.kv_indptr_cumsum(forward_batch.seq_lens, kv_indptr)
            kv_indptr = kv_indptr[: bs + 1]
            kv_indices = torch.empty(max(int(forward_batch.seq_lens_sum), 1), dtype=torch.int32, device=self.device)
            ops.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens, kv_indptr,
                                  None, kv_indices)
            seq_mask_len = nd * (forward_batch.seq_lens.to(torch.int64) + nd)
            mask_indptr = self.mask_indptr
            mask_indptr[1: bs + 1] = torch.cumsum(seq_mask_len[:bs], dim=0)
            self.forward_metadata = ForwardMetadata(None, None, nd, None, kv_indptr, kv_indices, qo_indptr,
                                                    custom_mask=spec_info.custom_mask, mask_indptr=mask_indptr[: bs + 1])
            return
        if is_mode("is_draft_extend"):
            # :375-391
            kv_indices, kv_indptr, qo_indptr, custom_mask = spec_info.generate_attn_arg_prefill(
                forward_batch.req_pool_indices, forward_batch.seq_lens, None, self.req_to_token)
            max_extend_len = int(torch.max(spec_info.accept_length).item())
            mask_indptr = None
            if custom_mask is not None:
                # without mask offsets the kernel cannot address the mask: rows of (prefix + extend) bytes per new token
                ext = (qo_indptr[1:] - qo_indptr[:-1]).to(torch.int64)
                pre = (kv_indptr[1:] - kv_indptr[:-1]).to(torch.int64)
                mask_indptr = torch.zeros(ext.numel() + 1, dtype=torch.int64, device=self.device)
                mask_indptr[1:] = torch.cumsum(ext * (pre + ext), dim=0)
            self.forward_metadata = ForwardMetadata(None, None, max_extend_len, None, kv_indptr, kv_indices, qo_indptr,
                                                    custom_mask=custom_mask, mask_indptr=mask_indptr)
            return
        # extend: the prefix part comes from the pool through kv_indices, the new part is contiguous
        ops.kv_indptr_cumsum(forward_batch.extend_prefix_lens, kv_indptr)
        kv_indptr = kv_indptr[: bs + 1]
        total_prefix = int(forward_batch.extend_prefix_lens.sum().item())
        kv_indices = torch.empty(max(total_prefix, 1), dtype=torch.int32, device=self.device)
        ops.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.extend_prefix_lens,
                              kv_indptr, None, kv_indices)
        qo_indptr = self.qo_indptr
        ops.kv_indptr_cumsum(forward_batch.extend_seq_lens, qo_indptr)
        qo_indptr = qo_indptr[: bs + 1]
        max_extend_len = int(forward_batch.extend_seq_lens.max().item())
        self.forward_metadata = ForwardMetadata(None, None, max_extend_len, None, kv_indptr, kv_indices, qo_indptr,
                                                total_prefix=total_prefix)
        if self.sliding_window_size:  # :412-421: the window over the cached prefix
            wp, wi, _ = self._window_lists(forward_batch.extend_prefix_lens, forward_batch.req_pool_indices, bs)
            self.forward_metadata.window_kv_indptr, self.forward_metadata.window_kv_indices = wp, wi

    # ---- graph capture: metadata lives at fixed addresses, refreshed by one launch per replay -----------------
    def init_cuda_graph_state(self, max_bs: int, kv_indices_buf: Optional[torch.Tensor] = None):
        """triton_backend.py:448-492."""
        self.cuda_graph_attn_logits, self.cuda_graph_attn_lse, self.cuda_graph_num_kv_splits = self._decode_buffers(max_bs)
        self.cuda_graph_num_kv_splits.fill_(self.max_kv_splits)
        self.cuda_graph_kv_indices = kv_indices_buf if kv_indices_buf is not None else torch.zeros(
            (max_bs * self.max_context_len,), dtype=torch.int32, device=self.device)
        self.cuda_graph_shared_workspace = self._shared_workspace(max_bs)
        self.cuda_graph_shared_plan = self._plan_buffer(min(max_bs, 1024)) if self._shares(1) else None
        if self.sliding_window_size:  # :480-490 (window + 1 slots per request: what update_sliding_window_buffer lists)
            self.cuda_graph_window_kv_indices = torch.zeros((max_bs * (self.sliding_window_size + 1),), dtype=torch.int32,
                                                            device=self.device)
            self.cuda_graph_window_num_kv_splits = torch.full((max_bs,), self.max_kv_splits, dtype=torch.int32,
                                                              device=self.device)

    def init_forward_metadata_capture_cuda_graph(self, bs: int, num_tokens: int, req_pool_indices: torch.Tensor,
                                                 seq_lens: torch.Tensor, encoder_lens=None, forward_mode=None,
                                                 spec_info=None):
        """triton_backend.py:494-625 (decode only)."""
        if forward_mode is not None and not forward_mode.is_decode_or_idle():
            raise SglkError("graph capture is implemented for decode batches")
        plan = self.cuda_graph_shared_plan if (spec_info is None and self._shares(bs)) else None
        self._graph_metadata(bs, req_pool_indices, seq_lens, plan)
        self.forward_metadata = ForwardMetadata(self.cuda_graph_attn_logits, self.cuda_graph_attn_lse, None,
                                                self.cuda_graph_num_kv_splits, self.kv_indptr, self.cuda_graph_kv_indices,
                                                None, shared_plan=plan,
                                                shared_workspace=self.cuda_graph_shared_workspace if plan is not None else None)
        self._graph_plan_bs = bs if plan is not None else None
        if self.sliding_window_size:  # :523-541
            wp, wi, wl = self._window_lists(seq_lens, req_pool_indices, bs, self.cuda_graph_window_kv_indices)
            self.get_num_kv_splits(self.cuda_graph_window_num_kv_splits[:bs], wl)
            md = self.forward_metadata
            md.window_kv_indptr, md.window_kv_indices = self.window_kv_indptr, wi
            md.window_num_kv_splits = self.cuda_graph_window_num_kv_splits

    def init_forward_metadata_replay_cuda_graph(self, bs: int, req_pool_indices: torch.Tensor, seq_lens: torch.Tensor,
                                                seq_lens_sum: int, encoder_lens=None, forward_mode=None, spec_info=None,
                                                seq_lens_cpu=None):
        """triton_backend.py:627-730: rewrite kv_indptr / kv_indices / num_kv_splits -- and the shared-prefix plan: the
        groups of THIS step's batch -- in place, in one call."""
        plan = self.cuda_graph_shared_plan if (spec_info is None and self._shares(bs)) else None
        self._graph_metadata(bs, req_pool_indices, seq_lens, plan)
        if self.sliding_window_size:  # :656-675
            _, _, wl = self._window_lists(seq_lens, req_pool_indices, bs, self.cuda_graph_window_kv_indices)
            self.get_num_kv_splits(self.cuda_graph_window_num_kv_splits[:bs], wl)

    def _graph_metadata(self, bs, req_pool_indices, seq_lens, plan):
        if plan is not None:
            ops.decode_metadata_shared(self.req_to_token, req_pool_indices[:bs], seq_lens[:bs], self.kv_indptr,
                                       self.cuda_graph_kv_indices, self.cuda_graph_num_kv_splits, self.num_head,
                                       self.num_kv_head, self.max_kv_splits, self.device_core_count, plan,
                                       head_dim=self.head_dim)
        else:
            ops.decode_metadata(self.req_to_token, req_pool_indices[:bs], seq_lens[:bs], self.kv_indptr,
                                self.cuda_graph_kv_indices, self.cuda_graph_num_kv_splits, self.num_head, self.num_kv_head,
                                self.max_kv_splits, self.device_core_count)

    def get_cuda_graph_seq_len_fill_value(self):
        return 1

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        if forward_batch.forward_mode.is_decode_or_idle():
            return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache)
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache)

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        """triton_backend.py:735-788."""
        md = self.forward_metadata
        o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim)) if layer.qk_head_dim != layer.v_head_dim \
            else torch.empty_like(q)
        if save_kv_cache:
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        kv_indptr, kv_indices, window = md.kv_indptr, md.kv_indices, -1
        if self._windowed(layer):  # :759-768
            if md.window_kv_indptr is None:
                raise SglkError("sliding-window layer, but model_runner.sliding_window_size was not set")
            kv_indptr, kv_indices, window = md.window_kv_indptr, md.window_kv_indices, layer.sliding_window_size
        ops.extend_attention_fwd(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), k.contiguous().view(-1, layer.tp_k_head_num, layer.qk_head_dim),
            v.contiguous().view(-1, layer.tp_k_head_num, layer.v_head_dim), o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id), md.qo_indptr, kv_indptr, kv_indices,
            md.custom_mask, True, md.mask_indptr, md.max_extend_len, layer.scaling, layer.logit_cap,
            sliding_window_size=window, total_prefix=md.total_prefix if window < 0 else None)
        return o

    @staticmethod
    def _windowed(layer) -> bool:
        sw = getattr(layer, "sliding_window_size", None)
        return sw is not None and sw > -1

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        """triton_backend.py:790-835."""
        md = self.forward_metadata
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim)) if layer.qk_head_dim != layer.v_head_dim \
            else torch.empty_like(q)
        if save_kv_cache:
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        kb = forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id)
        vb = forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id)
        q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
        o3 = o.view(-1, layer.tp_q_head_num, layer.v_head_dim)
        if self._windowed(layer):  # :814-816: the window's slots instead of the whole sequence
            if md.window_kv_indptr is None:
                raise SglkError("sliding-window layer, but model_runner.sliding_window_size was not set")
            ops.decode_attention_fwd(q3, kb, vb, o3, md.window_kv_indptr, md.window_kv_indices, md.attn_logits, md.attn_lse,
                                     md.window_num_kv_splits, self.max_kv_splits, layer.scaling, layer.logit_cap)
        elif md.shared_plan is not None and q3.dtype in (torch.float16, torch.bfloat16):
            # one launch for the whole batch: the plan's groups, and whole contexts for requests that share nothing
            ops.decode_attention_shared_planned(q3, kb, vb, o3, md.kv_indptr[: q3.shape[0] + 1], md.kv_indices,
                                                md.shared_plan, md.shared_workspace, self.max_kv_splits, layer.scaling,
                                                layer.logit_cap, num_kv_splits=md.num_kv_splits)
        else:
            ops.decode_attention_fwd(q3, kb, vb, o3, md.kv_indptr, md.kv_indices, md.attn_logits, md.attn_lse,
                                     md.num_kv_splits, self.max_kv_splits, layer.scaling, layer.logit_cap)
        return o
