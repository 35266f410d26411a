"""Decode-step driver for a Llama-family model on the sglk kernels.

This is the host-side sequence the reference runs per decode step (python/sglang/srt/models/llama.py:245-267
per layer; model_executor/model_runner.py:1139-1149 forward_decode; managers/schedule_batch.py:1464-1527
prepare_for_decode), reduced to what the hot path needs: fixed-address buffers, every launch on the current
stream, no allocation and no host synchronisation inside step(), so a step can be captured into a HIP graph
(the reference's CudaGraphRunner, model_executor/cuda_graph_runner.py).

Weights are synthetic (random int4 / scales in the AWQ layout of layers/quantization/awq.py:98-151).
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import List, Optional

import torch

from . import ops


@dataclasses.dataclass
class LlamaConfig:
    hidden: int = 4096
    layers: int = 32
    q_heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    inter: int = 14336
    vocab: int = 128256
    rope_theta: float = 500000.0
    max_pos: int = 8192
    eps: float = 1e-5
    group_size: int = 128
    quant: str = "awq"  # "awq" (int4), "bf16" or "fp8" (w8a8: per-channel weight scales, per-token activation scales)
    qkv_bias: bool = False  # Qwen2: the fused q/k/v projection carries a bias (models/qwen2.py:133-139)

    @staticmethod
    def llama3_8b(quant="awq"):
        return LlamaConfig(quant=quant)

    @staticmethod
    def llama3_70b_tp8_rank(quant="fp8"):
        """One rank's shard of Llama-3-70B under TP=8 (llama.py:72-86,145-160: q/k/v and gate/up column-sharded,
        o/down row-sharded, vocab-parallel lm_head): 8 of 64 query heads, 1 of 8 kv heads, 28672/8 MLP columns."""
        return LlamaConfig(hidden=8192, layers=80, q_heads=8, kv_heads=1, head_dim=128, inter=3584, vocab=128256 // 8,
                           quant=quant)

    @staticmethod
    def qwen2_0_5b(quant="bf16"):
        """Qwen2-0.5B (models/qwen2.py; config.json of Qwen/Qwen2-0.5B): 24 layers, hidden 896, 14 query / 2 kv heads of
        64, MLP 4864, vocabulary 151936, rope base 1e6, rms eps 1e-6, bias on q/k/v only."""
        return LlamaConfig(hidden=896, layers=24, q_heads=14, kv_heads=2, head_dim=64, inter=4864, vocab=151936,
                           rope_theta=1000000.0, max_pos=32768, eps=1e-6, quant=quant, qkv_bias=True)

    @staticmethod
    def tiny_qwen(quant="bf16"):
        return LlamaConfig(hidden=384, layers=2, q_heads=6, kv_heads=2, head_dim=64, inter=512, vocab=512, max_pos=2048,
                           eps=1e-6, quant=quant, qkv_bias=True)

    @staticmethod
    def tiny(quant="awq"):
        return LlamaConfig(hidden=512, layers=2, q_heads=8, kv_heads=2, head_dim=128, inter=1024, vocab=1024,
                           max_pos=2048, quant=quant)


class _AwqLinear:
    """AWQ weight in the reference layout plus the repacked copy the fused GEMM streams."""

    def __init__(self, K, N, group, dtype, device, gen, keep_reference_layout=False):
        self.K, self.N = K, N
        qweight = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=device, generator=gen)
        self.qzeros = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // group, N // 8), dtype=torch.int32, device=device,
                                    generator=gen)
        # U(0,1) * 0.01 scales as in the reference's kernel tests, rescaled so activations stay O(1)
        self.scales = (torch.rand(K // group, N, device=device, generator=gen) * (2.0 / math.sqrt(K) / 4.6)).to(dtype)
        self.packed = ops.awq_repack(qweight)
        self.qweight = qweight if keep_reference_layout else None
        self.ws_bytes = 0

    def bytes(self):
        return self.packed.numel() * 4 + self.qzeros.numel() * 4 + self.scales.numel() * 2


class _Fp8Linear:
    """w8a8 fp8 linear (layers/quantization/fp8.py / w8a8_fp8.py): e4m3 weight [N, K] with one scale per output
    channel; activations are quantised per token right before the product."""

    def __init__(self, K, N, dtype, device, gen):
        self.K, self.N = K, N
        w = torch.randn(N, K, device=device, generator=gen) / math.sqrt(K)
        amax = w.abs().amax(dim=1, keepdim=True).clamp(min=1e-12)
        self.scale = (amax / 448.0).float().reshape(N).contiguous()
        self.weight = (w / (amax / 448.0)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
        self.wpacked = None  # the weight in the sequential-streaming layout, once pack() has run

    def pack(self):
        """A second copy in the streaming layout of fp8_scaled_mm_packed (the decode step reads it; the row-major one
        stays for callers of the reference's fp8_scaled_mm).  Short inner dimensions stay on the row-major kernel, whose
        waves split K inside the block (measured on the 70B TP = 8 shard: K = 1024 and 3584 faster there, K >= 4096
        faster here)."""
        if self.wpacked is None and self.N % 16 == 0 and self.K % 128 == 0 and self.K >= 4096:
            self.wpacked = ops.fp8_repack(self.weight)
        return self

    def bytes(self):
        return self.weight.numel() + self.scale.numel() * 4


class _DenseLinear:
    def __init__(self, K, N, dtype, device, gen):
        self.K, self.N = K, N
        self.weight = (torch.randn(N, K, device=device, generator=gen) / math.sqrt(K)).to(dtype)
        self.wpacked = None  # the weight in the sequential-streaming layout, once pack() has run

    def pack(self):
        """Re-lay the weight out for sequential streaming (the lm_head: read whole at every step) and drop the
        row-major copy."""
        if self.wpacked is None and self.N % 16 == 0 and self.K % 128 == 0:
            self.wpacked = ops.dense_repack(self.weight)
            self.weight = None
        return self

    def bytes(self):
        return (self.weight if self.weight is not None else self.wpacked).numel() * 2


class LlamaDecoder:
    def __init__(self, cfg: LlamaConfig, batch: int, pool_slots: int, max_ctx: int, dtype=torch.bfloat16,
                 device="cuda", seed=0, max_kv_splits=8, keep_reference_layout=False):
        self.cfg, self.B, self.dtype, self.device = cfg, batch, dtype, device
        self._keep_ref = keep_reference_layout
        gen = torch.Generator(device=device).manual_seed(seed)
        c = cfg
        self.q_size, self.kv_size = c.q_heads * c.head_dim, c.kv_heads * c.head_dim

        def lin(K, N, keep=False):
            if c.quant == "awq":
                return _AwqLinear(K, N, c.group_size, dtype, device, gen, keep_reference_layout or keep)
            if c.quant == "fp8":
                return _Fp8Linear(K, N, dtype, device, gen)
            return _DenseLinear(K, N, dtype, device, gen)

        self.embed = (torch.randn(c.vocab, c.hidden, device=device, generator=gen) * 0.5).to(dtype)
        self.lm_head = _DenseLinear(c.hidden, c.vocab, dtype, device, gen).pack()
        self.final_norm = torch.ones(c.hidden, dtype=dtype, device=device)
        self.layers = []
        for _ in range(c.layers):
            self.layers.append(dict(
                ln1=(1.0 + 0.1 * torch.randn(c.hidden, device=device, generator=gen)).to(dtype),
                ln2=(1.0 + 0.1 * torch.randn(c.hidden, device=device, generator=gen)).to(dtype),
                qkv=lin(c.hidden, self.q_size + 2 * self.kv_size),
                o=lin(self.q_size, c.hidden),
                gate_up=lin(c.hidden, 2 * c.inter, keep=True),  # kept until the fused layout has been derived
                down=lin(c.inter, c.hidden),
                qkv_bias=((0.1 * torch.randn(self.q_size + 2 * self.kv_size, device=device, generator=gen)).to(dtype)
                          if c.qkv_bias else None),
            ))
        # rotary cache, layout of RotaryEmbedding._compute_cos_sin_cache (layers/rotary_embedding.py)
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32, device=device) / c.head_dim))
        t = torch.arange(c.max_pos, dtype=torch.float32, device=device)
        fr = torch.einsum("i,j->ij", t, inv)
        self.cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous()
        # KV pool: per layer [slots, Hkv, D], slot 0 a dummy (mem_cache/memory_pool.py:277-296)
        self.k_pool = [torch.zeros(pool_slots, c.kv_heads, c.head_dim, dtype=dtype, device=device) for _ in range(c.layers)]
        self.v_pool = [torch.zeros(pool_slots, c.kv_heads, c.head_dim, dtype=dtype, device=device) for _ in range(c.layers)]
        self.req_to_token = torch.zeros(batch + 1, max_ctx + 4, dtype=torch.int32, device=device)
        # attention metadata at fixed addresses (TritonAttnBackend.init_cuda_graph_state, triton_backend.py:448-487)
        self.max_kv_splits = max_kv_splits
        self.kv_indptr = torch.zeros(batch + 1, dtype=torch.int32, device=device)
        self.kv_indices = torch.zeros(batch * (max_ctx + 4), dtype=torch.int32, device=device)
        self.num_kv_splits = torch.zeros(batch, dtype=torch.int32, device=device)
        self.attn_logits = torch.empty(batch, c.q_heads, max_kv_splits, c.head_dim, dtype=torch.float32, device=device)
        self.attn_lse = torch.empty(batch, c.q_heads, max_kv_splits, dtype=torch.float32, device=device)
        self.cu_count = 256
        # activations
        B = batch
        self.hidden = torch.empty(B, c.hidden, dtype=dtype, device=device)
        self.residual = torch.empty(B, c.hidden, dtype=dtype, device=device)
        self.qkv = torch.empty(B, self.q_size + 2 * self.kv_size, dtype=dtype, device=device)
        self.attn_out = torch.empty(B, self.q_size, dtype=dtype, device=device)
        self.gate_up = torch.empty(B, 2 * c.inter, dtype=dtype, device=device)
        self.act = torch.empty(B, c.inter, dtype=dtype, device=device)
        self.logits = torch.empty(B, c.vocab, dtype=dtype, device=device)
        ws = 0
        if c.quant == "awq":
            for K, N in ((c.hidden, self.q_size + 2 * self.kv_size), (self.q_size, c.hidden), (c.hidden, 2 * c.inter),
                         (c.inter, c.hidden)):
                ws = max(ws, ops.awq_gemm_workspace_bytes(B, N, K))
        else:
            need = ops.fp8_mm_workspace_bytes if c.quant == "fp8" else ops.dense_gemm_workspace_bytes
            for K, N in ((c.hidden, self.q_size + 2 * self.kv_size), (self.q_size, c.hidden), (c.hidden, 2 * c.inter),
                         (c.inter, c.hidden)):
                ws = max(ws, need(B, N, K))
                if c.quant == "fp8":
                    ws = max(ws, ops.fp8_mm_packed_workspace_bytes(B, N, K))
        self.workspace = torch.zeros(max(ws, 256), dtype=torch.uint8, device=device)  # ticket words start at zero
        if c.quant == "fp8":
            kmax = max(c.hidden, self.q_size, c.inter)
            self.x_q = torch.empty(B, kmax, dtype=torch.float8_e4m3fn, device=device)
            self.x_s = torch.empty(B, dtype=torch.float32, device=device)
        # layer fusion: block sums of every GEMM input come from its producer, split-K partials go to the consumer
        self.fused = False
        self.fold_rope = False
        # ... and as blocks of their own at the head of the attention launch (decode_attention_shared_planned_rope):
        # the attention blocks' first trips to memory run beside the rotation instead of behind a kernel boundary
        # (SGLK_ROPE_ROLE=1; off: the hand-over from the rope blocks to the attention blocks -- stores acknowledged,
        # count, poll, query rows fetched -- costs the launch what the separate kernel cost the step: measured equal)
        self.rope_role = self.fused and c.head_dim == 128 and os.environ.get("SGLK_ROPE_ROLE", "0") == "1"
        self.xs_hidden = torch.zeros(c.hidden // 128, B, dtype=torch.float32, device=device)
        self.xs_attn = torch.zeros(self.q_size // 128, B, dtype=torch.float32, device=device)
        self.xs_act = torch.zeros(c.inter // 128, B, dtype=torch.float32, device=device)
        self.partial = None
        # batch state (ScheduleBatch fields, device resident)
        self.input_ids = torch.zeros(B, dtype=torch.int64, device=device)
        self.positions = torch.zeros(B, dtype=torch.int64, device=device)
        self.seq_lens = torch.zeros(B, dtype=torch.int64, device=device)
        self.out_cache_loc = torch.zeros(B, dtype=torch.int64, device=device)
        self.req_pool_indices = torch.arange(B, dtype=torch.int64, device=device)
        self.next_ids = torch.zeros(B, dtype=torch.int64, device=device)
        self.gemm_timer = None  # optional callable(name, fn) used by bench.py to time launches with events

    # ---- weights / bytes ------------------------------------------------------------------------
    def layer_weight_bytes(self) -> int:
        L = self.layers[0]
        return sum(L[k].bytes() for k in ("qkv", "o", "gate_up", "down"))

    # ---- batch set-up: shared prefix + private suffix (bench_in_batch_prefix shape) -------------------
    def setup_shared_prefix_batch(self, prefix_len: int, suffix_len: int, seed: int = 0, share: bool = True):
        """req_to_token rows share slots 1..prefix_len, then private ranges; pools filled with N(0,1)."""
        B = self.B
        gen = torch.Generator(device=self.device).manual_seed(seed + 1)
        r2t = torch.zeros_like(self.req_to_token)
        per_req = (self.k_pool[0].shape[0] - 1 - (prefix_len if share else 0)) // B
        base0 = 1 + (prefix_len if share else 0)
        self.private_base = torch.zeros(B, dtype=torch.int64, device=self.device)
        width = r2t.shape[1]
        for b in range(B):
            base = base0 + b * per_req
            first = prefix_len if share else 0
            if share:
                r2t[b, :prefix_len] = torch.arange(1, 1 + prefix_len, dtype=torch.int32, device=self.device)
            n = min(per_req, width - first)
            r2t[b, first:first + n] = torch.arange(base, base + n, dtype=torch.int32, device=self.device)
        self.req_to_token.copy_(r2t)
        for l in range(self.cfg.layers):
            self.k_pool[l].normal_(generator=gen)
            self.v_pool[l].normal_(generator=gen)
        L = prefix_len + suffix_len
        # state BEFORE the first decode step: L tokens cached, the next token goes to position L
        self.seq_lens.fill_(L)
        self.input_ids.copy_(torch.randint(0, self.cfg.vocab, (B,), device=self.device, generator=gen))
        self.prefix_len = prefix_len if share else 0

    # ---- one decode step ------------------------------------------------------------------------------
    def _linear(self, name, w, x, out, bias=None):
        def run():
            if isinstance(w, _AwqLinear):
                ops.awq_gemm(x, w.packed, w.scales, w.qzeros, bias, out, self.workspace)
            elif isinstance(w, _Fp8Linear):
                xq = self.x_q.view(-1)[:x.shape[0] * w.K].view(x.shape[0], w.K)
                ops.per_token_quant_fp8(x, xq, self.x_s)
                ops.fp8_scaled_mm(xq, w.weight.t(), self.x_s, w.scale, self.dtype, bias, out, self.workspace)
            else:
                ops.dense_gemm_nt(x, w.weight, bias, out, self.workspace)

        if self.gemm_timer is not None:
            self.gemm_timer(name, run)
        else:
            run()
        return out

    def prepare_for_decode(self):
        """schedule_batch.py:1464-1527: seq_lens += 1; out_cache_loc = the request's next free slot;
        req_to_token[req, seq_len - 1] = out_cache_loc.  All on device."""
        # the new token's position = tokens cached so far; private slots are pre-assigned per request
        ops.prepare_for_decode(self.req_to_token, self.req_pool_indices, self.seq_lens, self.positions, self.out_cache_loc)

    def _enable_dense_fusion(self) -> bool:
        """16-bit weights: every layer projection in the sequential-streaming layout (a second copy: the extend path
        keeps multiplying by the row-major one), K split over the chip, and the split sums of qkv / o / down handed to
        the kernels that consume them (rotation + pool write, fused add + norm) instead of a reduction launch each."""
        c = self.cfg
        ok = (c.quant == "bf16" and c.hidden % 128 == 0 and c.inter % 128 == 0 and self.q_size % 128 == 0
              and (self.q_size + 2 * self.kv_size) % 16 == 0 and getattr(self, "tp_group", None) is None)
        if ok:
            shapes = {"qkv": (c.hidden, self.q_size + 2 * self.kv_size), "o": (self.q_size, c.hidden),
                      "down": (c.inter, c.hidden)}
            self.part_splits = {k: ops.dense_gemm_partial_splits(self.B, n, kk) for k, (kk, n) in shapes.items()}
            ok = all(v > 0 for v in self.part_splits.values())
        if ok:
            need = max(self.part_splits[k] * self.B * n for k, (kk, n) in shapes.items())
            self.partial = torch.empty(need, dtype=torch.float32, device=self.device)
            # the gate_up slabs live in the workspace until the launch that sums them applies the activation
            gu = max(ops.dense_gemm_partial_splits(self.B, 2 * c.inter, c.hidden), 1) * self.B * 2 * c.inter * 4
            if self.workspace.numel() * self.workspace.element_size() < gu:
                self.workspace = torch.zeros(gu, dtype=torch.uint8, device=self.device)
            for L in self.layers:
                for name in ("qkv", "o", "gate_up", "down"):
                    w = L[name]
                    if w.wpacked is None:
                        w.wpacked = ops.dense_repack(w.weight)
        self.fused_dense = bool(ok)
        return self.fused_dense

    def enable_fusion(self, on: bool = True) -> bool:
        """Decoder-layer fusion (AWQ, 16-bit, head size 128, shapes the partial GEMM covers)."""
        c = self.cfg
        self.fused_dense = False
        if c.quant == "fp8":
            # no fusion forms for the fp8 step; `on` selects the streaming weight layout for its products
            need = 0
            for L in self.layers:
                for name in ("qkv", "o", "gate_up", "down"):
                    w = L[name]
                    if on:
                        w.pack()
                        if w.wpacked is not None:
                            need = max(need, ops.fp8_mm_packed_partial_splits(self.B, w.N, w.K) * self.B * w.N)
                    else:
                        w.wpacked = None
            # raw split sums of qkv / o / down on their way to the consumer kernels
            self.fp8_partial = torch.empty(need, dtype=torch.float32, device=self.device) if (on and need) else None
            self.fused = False
            return False
        if on and c.quant == "bf16":
            self.fused = False
            return self._enable_dense_fusion()
        ok = (on and c.quant == "awq" and c.head_dim == 128 and c.hidden % 128 == 0 and c.inter % 128 == 0
              and not c.qkv_bias)
        if ok:
            shapes = {"qkv": (c.hidden, self.q_size + 2 * self.kv_size), "o": (self.q_size, c.hidden),
                      "down": (c.inter, c.hidden)}
            self.part_splits = {k: ops.awq_gemm_partial_splits(self.B, n, kk, c.group_size) for k, (kk, n) in shapes.items()}
            ok = all(v > 0 for v in self.part_splits.values())
            if ok:
                need = max(self.part_splits[k] * self.B * n for k, (kk, n) in shapes.items())
                self.partial = torch.empty(need, dtype=torch.float32, device=self.device)
        self.fused = bool(ok)
        # rotary embedding and the new K / V rows folded into the shared-prefix attention kernel
        # (decode_attention_shared_prefix_rope; needs at most 16 query heads per kv head and full rotation of 128)
        # (off: four splits of fp32 partial sums per query row cost the attention kernel more than the separate
        # rotation launch they replace; worth it only where the projection runs unsplit)
        self.fold_rope = False
        # gate_up with the activation in its epilogue: columns interleaved in blocks of 64 (gate, up, gate, up, ...)
        self.fused_act = self.fused and c.inter % 128 == 0 and c.hidden % 512 == 0
        # gate_up and down in one launch of persistent blocks (ops.awq_chain), SGLK_CHAIN=1: off by default -- the
        # hand-over between the two jobs (stores acknowledged, count, poll, first activation stage fetched) takes as
        # long as the kernel boundary and the prologue it replaces, and the step measured 4 % slower with it
        self.chain = self.fused and self.B <= 64 and os.environ.get("SGLK_CHAIN", "0") == "1"
        if self.chain:
            self.chain_sync = ops.awq_chain_sync(self.device)
        if self.fused_act:
            for L in self.layers:
                w = L["gate_up"]
                if "gate_up_il" not in L and w.qweight is not None:
                    qw, sc, qz = ops.awq_interleave_gate_up(w.qweight, w.scales, w.qzeros)
                    L["gate_up_il"] = (ops.awq_repack(qw), sc, qz)
                    if not self._keep_ref:
                        w.qweight = None
            self.fused_act = all("gate_up_il" in L for L in self.layers)
        return self.fused

    def _timed(self, name, fn):
        if self.gemm_timer is not None:
            self.gemm_timer(name, fn)
        else:
            fn()

    def gemm_sweep(self, name: str):
        """Launch projection `name` of every layer back to back, in the form the decode step uses (timing aid)."""
        c = self.cfg
        tp = getattr(self, "tp_group", None) is not None
        src = {"qkv": self.hidden, "o": self.attn_out, "gate_up": self.hidden, "down": self.act}[name]
        row_out = self.tp_buf if tp else self.hidden
        dst = {"qkv": self.qkv, "o": row_out, "gate_up": self.gate_up, "down": row_out}[name]
        xs = {"qkv": self.xs_hidden, "o": self.xs_attn, "gate_up": self.xs_hidden, "down": self.xs_act}[name]
        for L in self.layers:
            w = L[name]
            if getattr(self, "fused_dense", False):
                if name == "gate_up":
                    ops.dense_gemm_packed_act(src, w.wpacked, self.act, 0, self.workspace)
                else:
                    ops.dense_gemm_packed_partial(src, w.wpacked, self.partial)
            elif not self.fused:
                if c.quant == "fp8" and getattr(self, "fp8_fold_quant", True):
                    self._fp8_mm(name, w, dst)
                else:
                    self._linear(name, w, src, dst)
            elif name == "gate_up" and getattr(self, "fused_act", False):
                il = L["gate_up_il"]
                ops.awq_gemm_silu_mul(self.hidden, il[0], il[1], il[2], self.act, self.xs_hidden, self.xs_act)
            elif name == "gate_up" or (tp and name != "qkv"):
                ops.awq_gemm_xs(src, w.packed, w.scales, w.qzeros, xs, dst, self.workspace)
            else:
                ops.awq_gemm_partial(src, w.packed, w.scales, w.qzeros, self.partial, xs)

    def step_fused(self):
        c = self.cfg
        self.prepare_for_decode()
        self._metadata()
        torch.index_select(self.embed, 0, self.input_ids, out=self.hidden)
        sm_scale = 1.0 / math.sqrt(c.head_dim)
        q_view = self.qkv[:, : self.q_size].view(self.B, c.q_heads, c.head_dim)
        o_view = self.attn_out.view(self.B, c.q_heads, c.head_dim)
        P, S = self.partial, self.part_splits
        tp = getattr(self, "tp_group", None)
        for li, L in enumerate(self.layers):
            if li == 0:
                self.residual.copy_(self.hidden)
                ops.fused_add_rmsnorm_ex(self.hidden, None, L["ln1"], c.eps, xsum_out=self.xs_hidden)
            elif tp is None:  # consumes the previous layer's down-projection partials
                ops.fused_add_rmsnorm_ex(self.hidden, self.residual, L["ln1"], c.eps, partials=P, splits=S["down"],
                                         xsum_out=self.xs_hidden)
            else:
                ops.fused_add_rmsnorm_ex(self.hidden, self.residual, L["ln1"], c.eps, xsum_out=self.xs_hidden)
            w = L["qkv"]
            self._timed("qkv", lambda: ops.awq_gemm_partial(self.hidden, w.packed, w.scales, w.qzeros, P, self.xs_hidden))
            if self.fold_rope and getattr(self, "shared", False):
                # the attention kernel assembles q and the new K / V rows from the projection's partial sums itself
                ops.decode_attention_shared_prefix_rope(P, S["qkv"], self.qkv.numel(), L["qkv_bias"], self.cos_sin,
                                                        self.positions, self.out_cache_loc, self.k_pool[li],
                                                        self.v_pool[li], o_view, self.kv_indptr, self.kv_indices,
                                                        self.group_indptr, self.group_prefix, self.shared_ws,
                                                        self.max_kv_splits, sm_scale, 0.0, self.xs_attn)
            elif self.rope_role and getattr(self, "shared", False) and getattr(self, "shared_plan", None) is not None:
                # rotation + pool write as the first blocks of the attention launch
                ops.decode_attention_shared_planned_rope(P, S["qkv"], self.qkv.numel(), L["qkv_bias"], self.cos_sin,
                                                         self.positions, self.out_cache_loc, q_view, self.k_pool[li],
                                                         self.v_pool[li], o_view, self.kv_indptr, self.kv_indices,
                                                         self.shared_plan, li + 1, self.shared_ws, self.max_kv_splits,
                                                         sm_scale, 0.0, self.xs_attn)
            else:
                ops.rope_and_store_kv_partial(self.qkv, P, S["qkv"], self.cos_sin, self.positions, self.k_pool[li],
                                              self.v_pool[li], self.out_cache_loc, c.q_heads, c.kv_heads, c.head_dim,
                                              bias=L["qkv_bias"])
                self.attention(li, q_view, o_view, sm_scale, xsum_out=self.xs_attn)
            w = L["o"]
            if tp is None:
                self._timed("o", lambda: ops.awq_gemm_partial(self.attn_out, w.packed, w.scales, w.qzeros, P, self.xs_attn))
                ops.fused_add_rmsnorm_ex(self.hidden, self.residual, L["ln2"], c.eps, partials=P, splits=S["o"],
                                         xsum_out=self.xs_hidden, zero_out=self.xs_act if self.fused_act else None)
            else:  # row-parallel: this rank's product (split-K closed inside the GEMM), the sum over ranks, the consumer
                self._timed("o", lambda: ops.awq_gemm_xs(self.attn_out, w.packed, w.scales, w.qzeros, self.xs_attn,
                                                         self.tp_buf, self.workspace))
                self._tp_sum()
                ops.fused_add_rmsnorm_ex(self.hidden, self.residual, L["ln2"], c.eps, xsum_out=self.xs_hidden,
                                         zero_out=self.xs_act if self.fused_act else None)
            if self.fused_act and tp is None and self.chain:
                # gate_up (+ activation) and down in one launch of persistent blocks: the down projection's first
                # weights are on their way while gate_up's blocks still reduce and store
                il, w = L["gate_up_il"], L["down"]
                self._timed("gate_up+down", lambda: ops.awq_chain(
                    [ops.chain_silu_mul(self.hidden, il[0], il[1], il[2], self.xs_hidden, self.act, self.xs_act),
                     ops.chain_partial(self.act, w.packed, w.scales, w.qzeros, self.xs_act, P)], self.chain_sync))
                continue
            if self.fused_act:
                il = L["gate_up_il"]
                self._timed("gate_up", lambda: ops.awq_gemm_silu_mul(self.hidden, il[0], il[1], il[2], self.act,
                                                                     self.xs_hidden, self.xs_act))
            else:
                w = L["gate_up"]
                self._timed("gate_up", lambda: ops.awq_gemm_xs(self.hidden, w.packed, w.scales, w.qzeros, self.xs_hidden,
                                                               self.gate_up, self.workspace))
                ops.act_and_mul_xs(self.act, self.gate_up, 0, self.xs_act)
            w = L["down"]
            if tp is None:
                self._timed("down", lambda: ops.awq_gemm_partial(self.act, w.packed, w.scales, w.qzeros, P, self.xs_act))
            else:
                self._timed("down", lambda: ops.awq_gemm_xs(self.act, w.packed, w.scales, w.qzeros, self.xs_act,
                                                            self.tp_buf, self.workspace))
                self._tp_sum()
        if tp is None:
            ops.fused_add_rmsnorm_ex(self.hidden, self.residual, self.final_norm, c.eps, partials=P, splits=S["down"])
        else:
            ops.fused_add_rmsnorm_ex(self.hidden, self.residual, self.final_norm, c.eps)
        self._lm_head()
        self._sample()

    def step_fused_dense(self):
        """The decode step over 16-bit weights in the streaming layout: 9 launches per layer (norm, qkv, rotation + pool
        write, attention, o, norm, gate_up, its split sum + activation, down), the split sums of qkv / o / down folded
        into their consumers.  Same arithmetic as the unfused sequence: the consumers add the splits in split order
        and round once, where the reduction launch would have."""
        c = self.cfg
        self.prepare_for_decode()
        self._metadata()
        torch.index_select(self.embed, 0, self.input_ids, out=self.hidden)
        sm_scale = 1.0 / math.sqrt(c.head_dim)
        q_view = self.qkv[:, : self.q_size].view(self.B, c.q_heads, c.head_dim)
        o_view = self.attn_out.view(self.B, c.q_heads, c.head_dim)
        P, S = self.partial, self.part_splits
        for li, L in enumerate(self.layers):
            if li == 0:
                self.residual.copy_(self.hidden)
                ops.fused_add_rmsnorm_ex(self.hidden, None, L["ln1"], c.eps)
            else:
                ops.fused_add_rmsnorm_ex(self.hidden, self.residual, L["ln1"], c.eps, partials=P, splits=S["down"])
            self._timed("qkv", lambda: ops.dense_gemm_packed_partial(self.hidden, L["qkv"].wpacked, P))
            ops.rope_and_store_kv_partial(self.qkv, P, S["qkv"], self.cos_sin, self.positions, self.k_pool[li],
                                          self.v_pool[li], self.out_cache_loc, c.q_heads, c.kv_heads, c.head_dim,
                                          bias=L["qkv_bias"])
            self.attention(li, q_view, o_view, sm_scale)
            self._timed("o", lambda: ops.dense_gemm_packed_partial(self.attn_out, L["o"].wpacked, P))
            ops.fused_add_rmsnorm_ex(self.hidden, self.residual, L["ln2"], c.eps, partials=P, splits=S["o"])
            self._timed("gate_up", lambda: ops.dense_gemm_packed_act(self.hidden, L["gate_up"].wpacked, self.act, 0,
                                                                     self.workspace))
            self._timed("down", lambda: ops.dense_gemm_packed_partial(self.act, L["down"].wpacked, P))
        ops.fused_add_rmsnorm_ex(self.hidden, self.residual, self.final_norm, c.eps, partials=P, splits=S["down"])
        self._lm_head()
        self._sample()

    def _lm_head(self):
        if self.lm_head.wpacked is not None:
            ops.dense_gemm_packed(self.hidden, self.lm_head.wpacked, None, self.logits)
        else:
            ops.dense_gemm_nt(self.hidden, self.lm_head.weight, None, self.logits)

    def _xq(self, K):
        return self.x_q.view(-1)[: self.B * K].view(self.B, K)

    def _fp8_mm(self, name, w, out):
        """The product on activations a producer has already quantised into x_q / x_s."""
        if w.wpacked is not None:
            run = lambda: ops.fp8_scaled_mm_packed(self._xq(w.K), w.wpacked, self.x_s, w.scale, self.dtype, None, out,
                                                   self.workspace)
        else:
            run = lambda: ops.fp8_scaled_mm(self._xq(w.K), w.weight.t(), self.x_s, w.scale, self.dtype, None, out,
                                            self.workspace)
        if self.gemm_timer is not None:
            self.gemm_timer(name, run)
        else:
            run()

    def step_fp8(self):
        """w8a8 layer with the activation quantisation folded into its producers (norm, gated activation), and, where a
        projection runs on the streaming weight layout, its raw split sums handed to the kernel that consumes them
        (rotation + pool write after qkv, fused add + norm + quantisation after o and down) instead of a reduction
        launch: 11 launches per layer at most, 8 with every hand-over."""
        c = self.cfg
        self.prepare_for_decode()
        self._metadata()
        torch.index_select(self.embed, 0, self.input_ids, out=self.hidden)
        sm_scale = 1.0 / math.sqrt(c.head_dim)
        q_view = self.qkv[:, : self.q_size].view(self.B, c.q_heads, c.head_dim)
        o_view = self.attn_out.view(self.B, c.q_heads, c.head_dim)
        tp = getattr(self, "tp_group", None)
        row_out = self.hidden if tp is None else self.tp_buf
        P = getattr(self, "fp8_partial", None)

        def product(name, w, out, hand_over):
            """Returns the number of split sums left in P for the consumer, 0 when `out` holds the finished rows."""
            if hand_over and P is not None and w.wpacked is not None:
                self._timed(name, lambda: ops.fp8_mm_packed_partial(self._xq(w.K), w.wpacked, P))
                return ops.fp8_mm_packed_partial_splits(self.B, w.N, w.K)
            self._fp8_mm(name, w, out)
            return 0

        pending, pending_w = 0, None   # the previous layer's down projection, when it left split sums
        last = len(self.layers) - 1
        for li, L in enumerate(self.layers):
            if li == 0:
                self.residual.copy_(self.hidden)
                ops.rmsnorm_quant_fp8(self.hidden, None, L["ln1"], self._xq(c.hidden), self.x_s, c.eps)
            elif pending:
                ops.rmsnorm_quant_fp8_partial(P, pending, self.x_s, pending_w.scale, self.residual, L["ln1"],
                                              self._xq(c.hidden), self.x_s, c.eps)
            else:
                ops.rmsnorm_quant_fp8(self.hidden, self.residual, L["ln1"], self._xq(c.hidden), self.x_s, c.eps)
            sq = product("qkv", L["qkv"], self.qkv, True)
            if sq:
                ops.rope_and_store_kv_partial(self.qkv, P, sq, self.cos_sin, self.positions, self.k_pool[li],
                                              self.v_pool[li], self.out_cache_loc, c.q_heads, c.kv_heads, c.head_dim,
                                              row_scale=self.x_s, col_scale=L["qkv"].scale)
            else:
                ops.rope_and_store_kv(self.qkv, self.cos_sin, self.positions, self.k_pool[li], self.v_pool[li],
                                      self.out_cache_loc, c.q_heads, c.kv_heads, c.head_dim)
            self.attention(li, q_view, o_view, sm_scale)
            ops.per_token_quant_fp8(self.attn_out, self._xq(self.q_size), self.x_s)
            so = product("o", L["o"], row_out, tp is None)
            if tp is not None:
                self._tp_sum()
            if so:
                ops.rmsnorm_quant_fp8_partial(P, so, self.x_s, L["o"].scale, self.residual, L["ln2"], self._xq(c.hidden),
                                              self.x_s, c.eps)
            else:
                ops.rmsnorm_quant_fp8(self.hidden, self.residual, L["ln2"], self._xq(c.hidden), self.x_s, c.eps)
            w = L["gate_up"]
            sg = ops.fp8_mm_packed_partial_splits(self.B, w.N, w.K) if (P is not None and w.wpacked is not None) else 0
            if sg > 1:  # an unsplit product gains nothing from the hand-over: it stores finished rows itself
                product("gate_up", w, self.gate_up, True)
                ops.silu_and_mul_quant_fp8_partial(P, sg, self.x_s, w.scale, self._xq(c.inter), self.x_s, self.dtype)
            else:
                self._fp8_mm("gate_up", w, self.gate_up)
                ops.silu_and_mul_quant_fp8(self.gate_up, self._xq(c.inter), self.x_s)
            pending = product("down", L["down"], row_out, tp is None and li != last)  # the last rows go to the final norm
            pending_w = L["down"]
            if tp is not None:
                self._tp_sum()
        ops.fused_add_rmsnorm(self.hidden, self.residual, self.final_norm, c.eps)
        self._lm_head()
        self._sample()

    def step(self):
        if self.fused:
            return self.step_fused()
        if getattr(self, "fused_dense", False):
            return self.step_fused_dense()
        if self.cfg.quant == "fp8" and getattr(self, "fp8_fold_quant", True) and not self.cfg.qkv_bias:
            return self.step_fp8()
        c = self.cfg
        self.prepare_for_decode()
        self._metadata()
        torch.index_select(self.embed, 0, self.input_ids, out=self.hidden)
        sm_scale = 1.0 / math.sqrt(c.head_dim)
        q_view = self.qkv[:, : self.q_size].view(self.B, c.q_heads, c.head_dim)
        o_view = self.attn_out.view(self.B, c.q_heads, c.head_dim)
        for li, L in enumerate(self.layers):
            if li == 0:
                self.residual.copy_(self.hidden)
                ops.rmsnorm(self.hidden, self.hidden, L["ln1"], c.eps)
            else:
                ops.fused_add_rmsnorm(self.hidden, self.residual, L["ln1"], c.eps)
            self._linear("qkv", L["qkv"], self.hidden, self.qkv, L["qkv_bias"])
            ops.rope_and_store_kv(self.qkv, self.cos_sin, self.positions, self.k_pool[li], self.v_pool[li],
                                  self.out_cache_loc, c.q_heads, c.kv_heads, c.head_dim)
            self.attention(li, q_view, o_view, sm_scale)
            self._row_parallel("o", L["o"], self.attn_out)
            ops.fused_add_rmsnorm(self.hidden, self.residual, L["ln2"], c.eps)
            self._linear("gate_up", L["gate_up"], self.hidden, self.gate_up)
            ops.act_and_mul(self.act, self.gate_up, 0)
            self._row_parallel("down", L["down"], self.act)
        ops.fused_add_rmsnorm(self.hidden, self.residual, self.final_norm, c.eps)
        self._lm_head()
        self._sample()

    def enable_tp(self, group) -> None:
        """Tensor parallelism over `group` (a distributed.parallel_state.GroupCoordinator): this decoder holds one
        rank's shard (llama.py:72-86,145-160).  The row-parallel projections (o, down) write this rank's product into a
        buffer registered with the group's peer-to-peer communicator and the all-reduce leaves the sum in `hidden`
        (linear.py RowParallelLinear -> tensor_model_parallel_all_reduce); the vocab-parallel lm_head is followed by an
        all-gather of every rank's logits (logits_processor.py -> tensor_model_parallel_all_gather), so every rank
        samples from the full row.  When the communicator declines a tensor or has been disabled, the group's RCCL
        collectives take over (parallel_state.py:436-461).  Works with the plain op sequence, the fp8 sequence and
        the fused layer path (there the split-K reduction closes inside the GEMM before the collective)."""
        self.tp_group = group
        self.fused_dense = False  # the 16-bit step's consumer-side split sums have no row-parallel form: plain sequence
        self.tp_rank, self.tp_world = group.rank_in_group, group.world_size
        self.tp_buf = torch.empty_like(self.hidden)
        self.tp_buf_registered = group.register_buffer(self.tp_buf)
        self.tp_logits_registered = group.register_buffer(self.logits)
        self.logits_all = torch.empty(self.tp_world, self.B, self.cfg.vocab, dtype=self.dtype, device=self.device)
        self.logits_full = torch.empty(self.B, self.tp_world * self.cfg.vocab, dtype=self.dtype, device=self.device)

    def _tp_sum(self):
        self.tp_group.all_reduce_into(self.tp_buf, self.hidden, registered=self.tp_buf_registered)

    def check_tp_health(self) -> None:
        """After the step's synchronise: raises if a peer never reached an all-reduce barrier (outputs are NaN then)."""
        tp = getattr(self, "tp_group", None)
        if tp is not None:
            tp.check_health()

    def _row_parallel(self, name, w, x):
        """o / down: the local product, then the sum over the TP ranks (linear.py RowParallelLinear)."""
        if getattr(self, "tp_group", None) is None:
            self._linear(name, w, x, self.hidden)
            return
        self._linear(name, w, x, self.tp_buf)
        self._tp_sum()

    def _sample(self):
        """Greedy token over the (vocab-parallel) logits."""
        tp = getattr(self, "tp_group", None)
        if tp is None:
            ops.argmax(self.logits, self.next_ids)
        else:
            tp.all_gather_into_tensor(self.logits_all, self.logits, registered=self.tp_logits_registered)
            self.logits_full.view(self.B, self.tp_world, self.cfg.vocab).copy_(self.logits_all.permute(1, 0, 2))
            ops.argmax(self.logits_full, self.next_ids)
            # this rank's embedding shard stands in for the whole table (synthetic weights): fold the id into it
            torch.remainder(self.next_ids, self.cfg.vocab, out=self.input_ids)
            return
        self.input_ids.copy_(self.next_ids)

    def enable_shared_prefix(self, on: bool = True) -> bool:
        """Route decode attention through the shared-prefix kernel: the batch is one group whose first
        prefix_len slots are the radix-tree hit every request shares (what RadixCache.match_prefix returned)."""
        self.shared = bool(on) and getattr(self, "prefix_len", 0) > 0
        if self.shared:
            c = self.cfg
            self.group_indptr = torch.tensor([0, self.B], dtype=torch.int32, device=self.device)
            self.group_prefix = torch.tensor([self.prefix_len], dtype=torch.int32, device=self.device)
            self.shared_ws = ops.decode_shared_workspace(self.B, c.q_heads, c.head_dim, self.max_kv_splits, self.device)
            # "derived": the groups are found on the device from the request table, once per step, as the attention
            # backend does (nothing about the batch is told to the kernel); "given": the plan is laid out from
            # group_indptr / group_prefix; "legacy": the kernel reads group_indptr / group_prefix itself, per layer
            self.plan_mode = getattr(self, "plan_mode", None) or os.environ.get("SGLK_SHARED_PLAN", "derived")
            self.shared_plan = None
            pair64 = c.head_dim == 64 and c.kv_heads % 2 == 0   # pairs of kv heads as one 128-wide head
            if self.plan_mode != "legacy" and (c.head_dim == 128 or pair64) and self.B <= 1024:
                self.shared_plan = ops.decode_shared_plan_buffer(self.B, c.q_heads, c.kv_heads, self.max_kv_splits,
                                                                 self.device, head_dim=c.head_dim)
        return self.shared

    def _metadata(self):
        """kv_indptr / kv_indices / num_kv_splits and, with prefix sharing on, the step's plan (one call)."""
        c = self.cfg
        plan = getattr(self, "shared_plan", None) if getattr(self, "shared", False) else None
        if plan is not None and self.plan_mode == "derived":
            ops.decode_metadata_shared(self.req_to_token, self.req_pool_indices, self.seq_lens, self.kv_indptr,
                                       self.kv_indices, self.num_kv_splits, c.q_heads, c.kv_heads, self.max_kv_splits,
                                       self.cu_count, plan, head_dim=c.head_dim)
            return
        ops.decode_metadata(self.req_to_token, self.req_pool_indices, self.seq_lens, self.kv_indptr, self.kv_indices,
                            self.num_kv_splits, c.q_heads, c.kv_heads, self.max_kv_splits, self.cu_count)
        if plan is not None:
            ops.decode_shared_plan_from_groups(self.kv_indptr, self.kv_indices, self.group_indptr, self.group_prefix,
                                               self.B, c.q_heads, c.kv_heads, self.max_kv_splits, plan,
                                               head_dim=c.head_dim)

    def attention(self, li, q_view, o_view, sm_scale, xsum_out=None):
        if getattr(self, "shared", False) and getattr(self, "shared_plan", None) is not None:
            ops.decode_attention_shared_planned(q_view, self.k_pool[li], self.v_pool[li], o_view, self.kv_indptr,
                                                self.kv_indices, self.shared_plan, self.shared_ws, self.max_kv_splits,
                                                sm_scale, 0.0, xsum_out)
            return
        if getattr(self, "shared", False):
            ops.decode_attention_shared_prefix_xs(q_view, self.k_pool[li], self.v_pool[li], o_view, self.kv_indptr,
                                                  self.kv_indices, self.group_indptr, self.group_prefix, self.shared_ws,
                                                  self.max_kv_splits, sm_scale, 0.0, xsum_out)
            return
        if xsum_out is not None:
            ops.decode_attention_fwd_xs(q_view, self.k_pool[li], self.v_pool[li], o_view, self.kv_indptr, self.kv_indices,
                                        self.attn_logits, self.attn_lse, self.num_kv_splits, self.max_kv_splits, sm_scale,
                                        0.0, xsum_out)
            return
        ops.decode_attention_fwd(q_view, self.k_pool[li], self.v_pool[li], o_view, self.kv_indptr, self.kv_indices,
                                 self.attn_logits, self.attn_lse, self.num_kv_splits, self.max_kv_splits, sm_scale, 0.0)
