"""Decode-step driver for DeepSeek-V3's dense decoder layers on the sglk kernels: absorbed multi-head latent
attention over a paged latent KV pool, block-quantised fp8 projections, dense gated MLP.

Per layer this is the op sequence of the reference's absorbed decode path
(python/sglang/srt/models/deepseek_v2.py:904-1034 forward_absorb_prepare / forward_absorb_core, 1351-1470
DeepseekV2DecoderLayer, 140-200 DeepseekV2MLP):

    input norm -> fused q_a / kv_a projection (block fp8) -> q_a_layernorm, kv_a_layernorm
    -> q_b projection (block fp8) -> q_nope . w_kc (per-tensor fp8 bmm) -> rope on q_pe / k_pe (interleaved)
    -> latent row [k_nope | k_pe] into the MLA pool (mem_cache/memory_pool.py:511-646) -> decode attention with
       576-wide keys and values = the first 512 elements of the same rows -> attn . w_vc (per-tensor fp8 bmm)
    -> o_proj (block fp8) -> post-attention norm -> gate_up (block fp8) -> silu * up -> down (block fp8)

Block fp8 is the checkpoint format of layers/quantization/fp8.py with weight_block_size [128, 128]: e4m3 weights
[N, K] with one fp32 scale per 128 x 128 block, activations quantised per token and per 128-element group right
before each product (fp8_kernel.py:190 per_token_group_quant_fp8, called from the block-fp8 linear of fp8_utils.py).  Where the reference's ROCm branch falls back to a bf16 torch.bmm for
the two absorbed products (deepseek_v2.py:959-964, "TODO add bmm_fp8 to ROCm"), this runs its CUDA branch
(per-tensor quantisation + bmm_fp8, :965-972) on the HIP bmm_fp8 kernel.

DeepSeek-V3 has three such dense layers (first_k_dense_replace = 3); the 58 routed-expert layers share the attention
block and differ in the MLP, whose grouped expert GEMMs are outside this package's path.  Weights are synthetic.
"""
from __future__ import annotations

import dataclasses
import math

import torch

from . import ops

F8 = torch.float8_e4m3fn


@dataclasses.dataclass
class DeepseekV3Config:
    hidden: int = 7168
    layers: int = 3                # the dense layers
    heads: int = 128
    q_lora_rank: int = 1536
    kv_lora_rank: int = 512        # the attention kernel is built for 512 + 64
    qk_nope_head_dim: int = 128
    qk_rope_head_dim: int = 64
    v_head_dim: int = 128
    inter: int = 18432
    rope_theta: float = 10000.0
    max_pos: int = 8192
    eps: float = 1e-6

    @staticmethod
    def deepseek_v3():
        return DeepseekV3Config()

    @staticmethod
    def tiny():
        return DeepseekV3Config(hidden=512, layers=2, heads=16, q_lora_rank=256, inter=1024, max_pos=2048)


class BlockFp8Linear:
    """e4m3 weight [N, K] + fp32 scale [ceil(N/128), K/128] (layers/quantization/fp8.py Fp8LinearMethod, block form)."""

    def __init__(self, K, N, device, gen):
        self.K, self.N = K, N
        w = torch.randn(N, K, device=device, generator=gen) / math.sqrt(K)
        nb, kb = (N + 127) // 128, K // 128
        pad = torch.zeros(nb * 128, K, device=device)
        pad[:N] = w
        amax = pad.view(nb, 128, kb, 128).abs().amax(dim=(1, 3)).clamp(min=1e-12)
        self.scale = (amax / 448.0).float().contiguous()                       # [nb, kb]
        s_full = self.scale.repeat_interleave(128, 0)[:N].repeat_interleave(128, 1)
        self.weight = (w / s_full).clamp(-448.0, 448.0).to(F8).contiguous()
        self.wpacked = None

    def pack(self):
        """A second copy in the streaming layout of fp8_blockwise_scaled_mm_packed for the decode step (inner
        dimensions of 4096 and more, or 1536 and more under a wide output (q_b): the rest is faster on the row-major
        kernel)."""
        if self.wpacked is None and self.N % 16 == 0 and self.K % 128 == 0 and (self.K >= 4096 or (self.K >= 1536 and self.N >= 16384)):
            self.wpacked = ops.fp8_repack(self.weight)
        return self

    def bytes(self):
        return self.weight.numel() + self.scale.numel() * 4

    def dequant(self):
        s_full = self.scale.repeat_interleave(128, 0)[: self.N].repeat_interleave(128, 1)
        return self.weight.float() * s_full


class DeepseekV3DenseDecoder:
    def __init__(self, cfg: DeepseekV3Config, batch: int, pool_slots: int, max_ctx: int, device="cuda", seed=0,
                 max_kv_splits=8, streaming_weights=True):
        self.cfg, self.B, self.device, self.dtype = cfg, batch, device, torch.bfloat16
        c, B, dt = cfg, batch, torch.bfloat16
        assert c.kv_lora_rank == 512 and c.qk_rope_head_dim == 64, "the latent attention kernel is built for 512 + 64"
        gen = torch.Generator(device=device).manual_seed(seed)
        self.qk_head_dim = c.qk_nope_head_dim + c.qk_rope_head_dim
        self.lat = c.kv_lora_rank + c.qk_rope_head_dim                          # 576
        self.sm_scale = 1.0 / math.sqrt(self.qk_head_dim)                        # deepseek_v2.py:586

        def lin(K, N):
            return BlockFp8Linear(K, N, device, gen).pack() if streaming_weights else BlockFp8Linear(K, N, device, gen)

        def nw(n):
            return (1.0 + 0.1 * torch.randn(n, device=device, generator=gen)).to(dt)

        self.layers = []
        for _ in range(c.layers):
            w_kc = torch.randn(c.heads, c.kv_lora_rank, c.qk_nope_head_dim, device=device, generator=gen) / math.sqrt(c.qk_nope_head_dim)
            w_vc = torch.randn(c.heads, c.v_head_dim, c.kv_lora_rank, device=device, generator=gen) / math.sqrt(c.kv_lora_rank)
            kv_scale = torch.tensor([max(w_kc.abs().max().item(), w_vc.abs().max().item()) / 448.0], device=device)
            self.layers.append(dict(
                ln1=nw(c.hidden), ln2=nw(c.hidden), q_a_ln=nw(c.q_lora_rank), kv_a_ln=nw(c.kv_lora_rank),
                qkv_a=lin(c.hidden, c.q_lora_rank + self.lat),
                q_b=lin(c.q_lora_rank, c.heads * self.qk_head_dim),
                # absorbed halves of kv_b_proj (deepseek_v2.py:1717 post_load_weights): one scale for both
                w_kc=(w_kc / kv_scale).clamp(-448, 448).to(F8),                   # [H, 512, 128] = column-major [H, 128, 512]
                w_vc=(w_vc / kv_scale).clamp(-448, 448).to(F8),                   # [H, 128, 512] = column-major [H, 512, 128]
                w_scale=kv_scale.float(),
                o=lin(c.heads * c.v_head_dim, c.hidden),
                gate_up=lin(c.hidden, 2 * c.inter),
                down=lin(c.inter, c.hidden),
            ))
        self.final_norm = nw(c.hidden)
        # rotary cache over the rope dims (layout of RotaryEmbedding._compute_cos_sin_cache), interleaved pairs
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.qk_rope_head_dim, 2, dtype=torch.float32, device=device) / c.qk_rope_head_dim))
        fr = torch.einsum("i,j->ij", torch.arange(c.max_pos, dtype=torch.float32, device=device), inv)
        self.cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).contiguous()
        # latent pool: per layer [slots, 1, 576], slot 0 a dummy (memory_pool.py:511-560)
        self.kv_pool = [torch.zeros(pool_slots, 1, self.lat, dtype=dt, device=device) for _ in range(c.layers)]
        self.req_to_token = torch.zeros(B + 1, max_ctx + 4, dtype=torch.int32, device=device)
        self.max_kv_splits = max_kv_splits
        self.kv_indptr = torch.zeros(B + 1, dtype=torch.int32, device=device)
        self.kv_indices = torch.zeros(B * (max_ctx + 4), dtype=torch.int32, device=device)
        self.num_kv_splits = torch.zeros(B, dtype=torch.int32, device=device)
        self.attn_logits = torch.empty(B, c.heads, max_kv_splits, c.kv_lora_rank, dtype=torch.float32, device=device)
        self.attn_lse = torch.empty(B, c.heads, max_kv_splits, dtype=torch.float32, device=device)
        self.cu_count = 256
        # activations at fixed addresses
        self.hidden = torch.zeros(B, c.hidden, dtype=dt, device=device)
        self.residual = torch.zeros(B, c.hidden, dtype=dt, device=device)
        self.q_input = torch.zeros(B, c.heads, self.lat, dtype=dt, device=device)
        self.qkv_a = torch.zeros(B, c.q_lora_rank + self.lat, dtype=dt, device=device)
        self.q_a = torch.zeros(B, c.q_lora_rank, dtype=dt, device=device)
        self.q_b = torch.zeros(B, c.heads * self.qk_head_dim, dtype=dt, device=device)
        self.attn_flat = torch.zeros(B, c.heads * c.v_head_dim, dtype=dt, device=device)
        self.gate_up = torch.zeros(B, 2 * c.inter, dtype=dt, device=device)
        self.attn_out = torch.zeros(B, c.heads, c.kv_lora_rank, dtype=dt, device=device)
        self.act = torch.zeros(B, c.inter, dtype=dt, device=device)
        kmax = max(c.hidden, c.q_lora_rank, c.heads * c.v_head_dim, c.inter)
        self.x_q = torch.empty(B * kmax, dtype=F8, device=device)
        self.x_s = torch.empty(B * (kmax // 128), dtype=torch.float32, device=device)
        self.bq = torch.empty(c.heads * B * max(c.qk_nope_head_dim, c.kv_lora_rank), dtype=F8, device=device)
        self.bs = torch.zeros(2 * c.layers, dtype=torch.float32, device=device)   # one scale slot per bmm of the step
        ws = 256
        for K, N in ((c.hidden, c.q_lora_rank + self.lat), (c.q_lora_rank, c.heads * self.qk_head_dim),
                     (c.heads * c.v_head_dim, c.hidden), (c.hidden, 2 * c.inter), (c.inter, c.hidden)):
            ws = max(ws, ops.fp8_mm_workspace_bytes(B, N, K), ops.fp8_mm_packed_workspace_bytes(B, N, K))
        self.workspace = torch.zeros(ws, dtype=torch.uint8, device=device)
        self.positions = torch.zeros(B, dtype=torch.int64, device=device)
        self.seq_lens = torch.zeros(B, dtype=torch.int64, device=device)
        self.out_cache_loc = torch.zeros(B, dtype=torch.int64, device=device)
        self.req_pool_indices = torch.arange(B, dtype=torch.int64, device=device)

    def layer_weight_bytes(self) -> int:
        L = self.layers[0]
        return sum(L[k].bytes() for k in ("qkv_a", "q_b", "o", "gate_up", "down")) + L["w_kc"].numel() + L["w_vc"].numel()

    # ---- batch set-up: shared prefix + private suffix, as decode_runner.LlamaDecoder ---------------------------
    def setup_shared_prefix_batch(self, prefix_len: int, suffix_len: int, seed: int = 0):
        B = self.B
        gen = torch.Generator(device=self.device).manual_seed(seed + 1)
        r2t = torch.zeros_like(self.req_to_token)
        per_req = (self.kv_pool[0].shape[0] - 1 - prefix_len) // B
        width = r2t.shape[1]
        for b in range(B):
            base = 1 + prefix_len + b * per_req
            r2t[b, :prefix_len] = torch.arange(1, 1 + prefix_len, dtype=torch.int32, device=self.device)
            n = min(per_req, width - prefix_len)
            r2t[b, prefix_len:prefix_len + n] = torch.arange(base, base + n, dtype=torch.int32, device=self.device)
        self.req_to_token.copy_(r2t)
        for l in range(self.cfg.layers):
            self.kv_pool[l].normal_(generator=gen)
        self.seq_lens.fill_(prefix_len + suffix_len)
        self.hidden.normal_(generator=gen)

    # ---- pieces ---------------------------------------------------------------------------------------------------
    def _quant(self, x2d):
        """Per token and 128-element group (fp8_kernel.py:190 -> sgl_per_token_group_quant_fp8)."""
        M, K = x2d.shape
        q = self.x_q[: M * K].view(M, K)
        s = self.x_s[: M * (K // 128)].view(M, K // 128)
        ops.per_token_group_quant_fp8(x2d, q, s, 128, 1e-10, -448.0, 448.0)
        return q, s

    def _qbuf(self, M, K):
        return self.x_q[: M * K].view(M, K), self.x_s[: M * (K // 128)].view(M, K // 128)

    def _mm(self, x2d, w: BlockFp8Linear, out=None, quantised=None):
        """quantised: (q, s) when the producer of x2d already left its group-quantised form behind."""
        q, s = self._quant(x2d) if quantised is None else quantised
        if w.wpacked is not None:
            return ops.fp8_blockwise_scaled_mm_packed(q, w.wpacked, s, w.scale.t(), self.dtype, self.workspace, out)
        return ops.fp8_blockwise_scaled_mm(q, w.weight.t(), s, w.scale.t(), self.dtype, self.workspace, out)

    def _bmm(self, a_bhk, w_hnk, w_scale, scale, out_hbn):
        """a [B, H, K] bf16 (rows one pitch apart) x w [H, N, K] e4m3 -> out [H, B, N] (a view with any row pitch):
        per-tensor dynamic quantisation of a (per_tensor_quant_mla_fp8, deepseek_v2.py:965-972) in the layout it
        comes in, then bmm_fp8 over the transposed view; `scale` is this product's own zeroed scale slot."""
        B, H, K = a_bhk.shape
        q = self.bq[: B * H * K].view(B, H, K)
        ops.per_tensor_quant_fp8_rows(a_bhk, q, scale, False)
        return ops.bmm_fp8(q.transpose(0, 1), w_hnk.transpose(1, 2), scale, w_scale, self.dtype, out=out_hbn)

    def step(self, input_hidden=None):
        """One decode step of the layer stack: `hidden` holds the layers' input (the embedding rows of the new tokens)
        on entry and the normalised output of the last layer on exit.  Every product writes where its consumer
        reads: no copies between the launches."""
        c, B = self.cfg, self.B
        ops.prepare_for_decode(self.req_to_token, self.req_pool_indices, self.seq_lens, self.positions, self.out_cache_loc)
        ops.decode_metadata(self.req_to_token, self.req_pool_indices, self.seq_lens, self.kv_indptr, self.kv_indices,
                            self.num_kv_splits, c.heads, 1, self.max_kv_splits, self.cu_count)
        if input_hidden is not None:
            self.hidden.copy_(input_hidden)
        self.bs.zero_()                                                              # the running maxima of every bmm
        nope, lat = c.qk_nope_head_dim, c.kv_lora_rank
        attn_hbn = self.attn_flat.view(B, c.heads, c.v_head_dim).transpose(0, 1)    # [H, B, 128] over [B, H * 128]
        for li, L in enumerate(self.layers):
            # the norms and the activation leave the group-quantised input of the product behind them (the 16-bit
            # rows in between are formed and rounded in registers only)
            xq = self._qbuf(B, c.hidden)
            if li == 0:
                self.residual.copy_(self.hidden)
                ops.rmsnorm_group_quant_fp8(self.hidden, None, L["ln1"], xq[0], xq[1], c.eps)
            else:
                ops.rmsnorm_group_quant_fp8(self.hidden, self.residual, L["ln1"], xq[0], xq[1], c.eps)
            qkv_a = self._mm(None, L["qkv_a"], self.qkv_a, quantised=xq)               # [B, 1536 + 576]
            xq = self._qbuf(B, c.q_lora_rank)
            ops.rmsnorm_group_quant_fp8(qkv_a[:, : c.q_lora_rank], None, L["q_a_ln"], xq[0], xq[1], c.eps)
            k_nope = qkv_a[:, c.q_lora_rank: c.q_lora_rank + lat]
            ops.rmsnorm(k_nope, k_nope, L["kv_a_ln"], c.eps)
            q = self._mm(None, L["q_b"], self.q_b, quantised=xq).view(B, c.heads, self.qk_head_dim)
            # absorbed key projection: q_nope [B, H, 128] . w_kc^T -> q_input[..., :512]
            self._bmm(q[..., :nope], L["w_kc"], L["w_scale"], self.bs[2 * li: 2 * li + 1],
                      self.q_input[..., :lat].transpose(0, 1))
            # rope on q_pe / k_pe, q_input[..., 512:] = q_pe, the latent row [k_nope | k_pe] into the pool
            pool = self.kv_pool[li]
            ops.mla_rope_and_store(q[..., nope:], self.q_input, qkv_a[:, c.q_lora_rank:], pool, self.out_cache_loc,
                                   self.cos_sin, self.positions, lat, is_neox=False)
            ops.decode_attention_fwd(self.q_input, pool, pool[..., :lat], self.attn_out, self.kv_indptr, self.kv_indices,
                                     self.attn_logits, self.attn_lse, self.num_kv_splits, self.max_kv_splits, self.sm_scale)
            # absorbed value projection: attn [B, H, 512] . w_vc^T -> [B, H * 128]
            self._bmm(self.attn_out, L["w_vc"], L["w_scale"], self.bs[2 * li + 1: 2 * li + 2], attn_hbn)
            self._mm(self.attn_flat, L["o"], self.hidden)
            xq = self._qbuf(B, c.hidden)
            ops.rmsnorm_group_quant_fp8(self.hidden, self.residual, L["ln2"], xq[0], xq[1], c.eps)
            gu = self._mm(None, L["gate_up"], self.gate_up, quantised=xq)
            xq = self._qbuf(B, c.inter)
            ops.silu_and_mul_group_quant_fp8(gu, xq[0], xq[1])
            self._mm(None, L["down"], self.hidden, quantised=xq)
        ops.fused_add_rmsnorm(self.hidden, self.residual, self.final_norm, c.eps)
