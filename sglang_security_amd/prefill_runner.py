"""Extend (prefill-with-prefix) step of a Llama-family model on the sglk kernels, and the prefix-cache harness that
times it the way the fork's timing experiment does.

Per layer this is the reference's extend forward (python/sglang/srt/models/llama.py:245-267;
model_executor/model_runner.py:1151-1164 forward_extend): norm, qkv projection, rotary + KV-pool write, extend
attention over the cached prefix (through kv_indices) and the new tokens (causal), o projection, norm, gated MLP.
The reference always dequantises an AWQ weight and multiplies with a plain GEMM (layers/quantization/awq.py:197-215).
Here fewer than DEQUANT_GEMM_MIN_TOKENS rows go through the fused int4 kernel and only larger batches take the
dequantise-then-GEMM route; the threshold is this package's own choice, not the reference's.

`PrefixCacheHarness` is the scheduler's part around it, reduced to one request at a time: match_prefix with the
sender's user id, slot allocation, req_to_token row, extend of the unmatched tail, cache_finished_req
(managers/schedule_batch.py:1072-1200 prepare_for_extend; mem_cache/radix_cache.py:160-197).  The time to the first
token of a request whose prefix is cached by another user is what the fork's experiment measures
(ndss_scripts/timing_side_channel_attacker.py): short when the nodes are public, as long as a cold request when
SafeKV has marked them private.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import ops
from .decode_runner import LlamaDecoder, _AwqLinear, _DenseLinear, _Fp8Linear
from .mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool, TokenToKVPoolAllocator
from .mem_cache.radix_cache import RadixCache

DEQUANT_GEMM_MIN_TOKENS = 256  # this package's heuristic: where re-reading int4 weights per 64-row tile stops paying


class LlamaPrefill:
    """Extend forward over the weights of a LlamaDecoder and the buffers of an MHATokenToKVPool."""

    def __init__(self, dec: LlamaDecoder, kv_pool: MHATokenToKVPool, max_tokens: int, use_graph: bool = True,
                 resident_bf16: bool = False, prefill_gemm: str = "library", overlap_dequant: bool = False):
        """prefill_gemm: what multiplies batches of DEQUANT_GEMM_MIN_TOKENS rows or more.  "library": dequantise the
        int4 weight (sglk_awq_dequantize_packed), then torch.matmul = hipBLASLt -- the faster of the two today;
        "fused": the hand-written tile kernels of this package (sglk_awq_gemm_prefill dequantises inside the GEMM and
        writes no dequantised copy; sglk_dense_gemm_prefill for 16-bit weights).

        overlap_dequant: on the "library" path, dequantise projection i + 1 on a second stream while the library
        multiplies by projection i; three weight-sized buffers go round.  Same values either way, and off by default:
        measured on Llama-3-8B at 1152 tokens the two kernels slow each other down by more than the overlap hides
        (cold request 27.5 ms overlapped against 23.7 ms in sequence on the same GPU)."""
        if prefill_gemm not in ("library", "fused"):
            raise ValueError('prefill_gemm is "library" or "fused"')
        self.prefill_gemm = prefill_gemm
        self.overlap_dequant = overlap_dequant
        self._pipe = None                                # state of the running extend's dequantisation pipeline
        self._ring = None
        c = dec.cfg
        self.dec, self.cfg, self.kv = dec, c, kv_pool
        self.max_tokens = max_tokens
        dev, dt = dec.device, dec.dtype
        self.q_size, self.kv_size = dec.q_size, dec.kv_size
        T = max_tokens
        self.hidden = torch.empty(T, c.hidden, dtype=dt, device=dev)
        self.residual = torch.empty(T, c.hidden, dtype=dt, device=dev)
        self.qkv = torch.empty(T, self.q_size + 2 * self.kv_size, dtype=dt, device=dev)
        self.attn_out = torch.empty(T, self.q_size, dtype=dt, device=dev)
        self.gate_up = torch.empty(T, 2 * c.inter, dtype=dt, device=dev)
        self.act = torch.empty(T, c.inter, dtype=dt, device=dev)
        ws = 256
        if c.quant == "awq":
            m = min(T, DEQUANT_GEMM_MIN_TOKENS)
            for K, N in ((c.hidden, self.q_size + 2 * self.kv_size), (self.q_size, c.hidden), (c.hidden, 2 * c.inter),
                         (c.inter, c.hidden)):
                ws = max(ws, ops.awq_gemm_workspace_bytes(m, N, K))
        self.workspace = torch.zeros(ws, dtype=torch.uint8, device=dev)
        self.logits = torch.empty(1, c.vocab, dtype=dt, device=dev)
        self.next_id = torch.zeros(1, dtype=torch.int64, device=dev)
        # resident_bf16: keep a dequantised copy of every AWQ weight for the large-batch path (2 bytes per weight on top
        # of the int4 tensors: 14 GB for Llama-3-8B, affordable next to 288 GB of HBM) instead of dequantising per call
        self.resident_bf16 = resident_bf16
        self._dense = {}
        # fixed-address inputs of the captured single-request extends
        self.use_graph = use_graph
        self.graphs = {}
        g = min(T, DEQUANT_GEMM_MIN_TOKENS)
        self.g_ids = torch.zeros(g, dtype=torch.int64, device=dev)
        self.g_pos = torch.zeros(g, dtype=torch.int64, device=dev)
        self.g_loc = torch.zeros(g, dtype=torch.int64, device=dev)
        self.g_qo = torch.zeros(2, dtype=torch.int32, device=dev)
        self.g_kp = torch.zeros(2, dtype=torch.int32, device=dev)
        self.g_ki = torch.zeros(max(T, 1), dtype=torch.int32, device=dev)
        # Short single-request extends (the tail after a prefix hit) take the decode step's layer fusion when the
        # decoder has it: split-K partials handed to the consumer kernels, block sums from the producers, the gated
        # activation in the gate_up epilogue -- 8 launches per layer instead of 11 and no in-kernel split-K reductions.
        self.fused_rows = 0
        if c.quant == "awq" and dec.enable_fusion(True) and getattr(dec, "fused_act", False):
            g = min(T, DEQUANT_GEMM_MIN_TOKENS - 1)
            shapes = {"qkv": (c.hidden, self.q_size + 2 * self.kv_size), "o": (self.q_size, c.hidden),
                      "down": (c.inter, c.hidden)}
            need = 0
            for m in sorted({min(g, r) for r in (64, 128, 192, 255)}):
                sp = {k: ops.awq_gemm_partial_splits(m, n, kk, c.group_size) for k, (kk, n) in shapes.items()}
                if min(sp.values()) <= 0:
                    need = 0
                    break
                need = max(need, max(sp[k] * g * n for k, (kk, n) in shapes.items()))
            if need:
                self.fused_rows = g
                self._shapes = shapes
                self.partial = torch.empty(need, dtype=torch.float32, device=dev)
                self.xs_hidden = torch.zeros(c.hidden // 128, 256, dtype=torch.float32, device=dev)
                self.xs_act = torch.zeros(c.inter // 128, 256, dtype=torch.float32, device=dev)

    def _linear(self, w, x, out):
        T = x.shape[0]
        fused = self.prefill_gemm == "fused" and T >= DEQUANT_GEMM_MIN_TOKENS and w.N % 128 == 0 and w.K % 128 == 0
        if isinstance(w, _AwqLinear):
            if fused:
                ops.awq_gemm_prefill(x, w.packed, w.scales, w.qzeros, None, out)
            elif T >= DEQUANT_GEMM_MIN_TOKENS:
                if self._pipe is not None:
                    return self._pipe_matmul(w, x, out)
                if self.resident_bf16:
                    wd = self._dense.get(id(w))
                    if wd is None:
                        wd = self._dense[id(w)] = ops.awq_dequantize_packed(w.packed, w.scales, w.qzeros)
                else:
                    wd = ops.awq_dequantize_packed(w.packed, w.scales, w.qzeros)
                torch.matmul(x, wd, out=out)
            else:
                ops.awq_gemm(x, w.packed, w.scales, w.qzeros, None, out, self.workspace)
        elif isinstance(w, _DenseLinear):
            if fused:
                ops.dense_gemm_prefill(x, w.weight, None, out)
            else:
                torch.matmul(x, w.weight.t(), out=out)
        else:
            raise NotImplementedError("prefill runner: awq and bf16 weights")
        return out

    # -- dequantisation one projection ahead of the library product, on a second stream --------------------------------
    _RING = 3

    def _pipe_begin(self, T):
        c = self.cfg
        if not (self.overlap_dequant and self.prefill_gemm == "library" and not self.resident_bf16
                and c.quant == "awq" and T >= DEQUANT_GEMM_MIN_TOKENS):
            return
        if self._ring is None:
            seq = [L[k] for L in self.dec.layers for k in ("qkv", "o", "gate_up", "down")]
            if not all(isinstance(w, _AwqLinear) for w in seq):
                self.overlap_dequant = False
                return
            n = max(w.K * w.N for w in seq)
            self._seq = seq
            self._ring = [torch.empty(n, dtype=self.dec.dtype, device=self.dec.device) for _ in range(self._RING)]
            self._side = torch.cuda.Stream(device=self.dec.device)
            self._ready = [torch.cuda.Event() for _ in range(self._RING)]
            self._free = [torch.cuda.Event() for _ in range(self._RING)]
        self._side.wait_stream(torch.cuda.current_stream(self.dec.device))   # earlier readers of the ring
        self._pipe = 0
        self._pipe_issue(0)

    def _pipe_issue(self, j):
        if j >= len(self._seq):
            return
        w, slot = self._seq[j], j % self._RING
        if j >= self._RING:
            self._side.wait_event(self._free[slot])      # the product that read this buffer last
        with torch.cuda.stream(self._side):
            ops.awq_dequantize_packed(w.packed, w.scales, w.qzeros, out=self._ring[slot][: w.K * w.N].view(w.K, w.N))
            self._ready[slot].record(self._side)

    def _pipe_matmul(self, w, x, out):
        j = self._pipe
        if w is not self._seq[j]:
            raise RuntimeError("prefill runner: projections out of order for the dequantisation pipeline")
        slot = j % self._RING
        self._pipe_issue(j + 1)
        main = torch.cuda.current_stream(self.dec.device)
        main.wait_event(self._ready[slot])
        torch.matmul(x, self._ring[slot][: w.K * w.N].view(w.K, w.N), out=out)
        self._free[slot].record(main)
        self._pipe = j + 1
        return out

    def extend(self, input_ids: torch.Tensor, positions: torch.Tensor, out_cache_loc: torch.Tensor,
               qo_indptr: torch.Tensor, kv_indptr: torch.Tensor, kv_indices: torch.Tensor, max_extend_len: int):
        """One extend forward over T = len(input_ids) new tokens of the requests described by the indptr arrays
        (ForwardMetadata of the attention backend); returns the hidden states [T, hidden] after the last layer's MLP
        (before the final residual add and norm).

        Single-request extends below the dequantise-then-GEMM switch (the tail of a request whose prefix came from the
        cache: ≈290 launches of a few microseconds each) are replayed from a HIP graph captured once per token count;
        the graph reads its inputs from fixed buffers, as the decode graphs do
        (model_executor/cuda_graph_runner.py)."""
        T = input_ids.shape[0]
        if T > self.max_tokens:
            raise ValueError(f"extend of {T} tokens, buffers hold {self.max_tokens}")
        single = qo_indptr.numel() == 2 and max_extend_len == T
        if not (single and T < DEQUANT_GEMM_MIN_TOKENS and kv_indices.numel() <= self.g_ki.numel()):
            return self._extend(input_ids, positions, out_cache_loc, qo_indptr, kv_indptr, kv_indices, max_extend_len)
        pre = int(kv_indptr[-1].item())  # cached prefix tokens (kv_indices may be a one-element placeholder without any)
        fused = 0 < T <= self.fused_rows
        if not (self.use_graph or fused):
            return self._extend(input_ids, positions, out_cache_loc, qo_indptr, kv_indptr, kv_indices, max_extend_len)
        self.g_ids[:T].copy_(input_ids)
        self.g_pos[:T].copy_(positions)
        self.g_loc[:T].copy_(out_cache_loc)
        self.g_qo.copy_(qo_indptr)
        self.g_kp.copy_(kv_indptr)
        self.g_ki[:pre].copy_(kv_indices[:pre])
        args = (self.g_ids[:T], self.g_pos[:T], self.g_loc[:T], self.g_qo, self.g_kp, self.g_ki, T)
        run = self._extend_fused if fused else self._extend
        if not self.use_graph:
            return run(*args)
        g = self.graphs.get((T, fused))
        if g is None:
            run(*args)                               # first touch outside the capture (lazy initialisations)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run(*args)
            self.graphs[(T, fused)] = g
        g.replay()
        return self.hidden[:T], self.residual[:T]

    def _extend_fused(self, input_ids, positions, out_cache_loc, qo_indptr, kv_indptr, kv_indices, max_extend_len):
        """One request's short extend on the decode step's fused layer forms (same arithmetic as _extend: the consumer
        kernels round the summed partials where the GEMM would have).  The rotation kernel leaves the rotated k and v
        in the qkv buffer as well as in the pool, so the attention is the same call as in _extend."""
        c = self.cfg
        T = input_ids.shape[0]
        hidden, residual = self.hidden[:T], self.residual[:T]
        qkv, attn_out, act = self.qkv[:T], self.attn_out[:T], self.act[:T]
        xs_h, xs_a = self.xs_hidden[:, :T], self.xs_act[:, :T]
        P = self.partial
        S = {k: ops.awq_gemm_partial_splits(T, n, kk, c.group_size) for k, (kk, n) in self._shapes.items()}
        torch.index_select(self.dec.embed, 0, input_ids, out=hidden)
        sm_scale = 1.0 / math.sqrt(c.head_dim)
        q = qkv[:, : self.q_size].view(T, c.q_heads, c.head_dim)
        k = qkv[:, self.q_size: self.q_size + self.kv_size].view(T, c.kv_heads, c.head_dim)
        v = qkv[:, self.q_size + self.kv_size:].view(T, c.kv_heads, c.head_dim)
        o = attn_out.view(T, c.q_heads, c.head_dim)
        last = len(self.dec.layers) - 1
        for li, L in enumerate(self.dec.layers):
            if li == 0:
                residual.copy_(hidden)
                ops.fused_add_rmsnorm_ex(hidden, None, L["ln1"], c.eps, xsum_out=xs_h)
            else:
                ops.fused_add_rmsnorm_ex(hidden, residual, L["ln1"], c.eps, partials=P, splits=S["down"], xsum_out=xs_h)
            w = L["qkv"]
            ops.awq_gemm_partial(hidden, w.packed, w.scales, w.qzeros, P, xs_h)
            kb, vb = self.kv.get_kv_buffer(li)
            ops.rope_and_store_kv_partial(qkv, P, S["qkv"], self.dec.cos_sin, positions, kb, vb, out_cache_loc, c.q_heads,
                                          c.kv_heads, c.head_dim, bias=L["qkv_bias"])
            ops.extend_attention_fwd(q, k, v, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None, True, None, max_extend_len,
                                     sm_scale)
            w = L["o"]
            ops.awq_gemm_partial(attn_out, w.packed, w.scales, w.qzeros, P)
            ops.fused_add_rmsnorm_ex(hidden, residual, L["ln2"], c.eps, partials=P, splits=S["o"], xsum_out=xs_h,
                                     zero_out=self.xs_act)
            il = L["gate_up_il"]
            ops.awq_gemm_silu_mul(hidden, il[0], il[1], il[2], act, xs_h, xs_a)
            w = L["down"]
            if li == last:  # the caller wants the finished rows
                ops.awq_gemm_xs(act, w.packed, w.scales, w.qzeros, xs_a, hidden, self.workspace)
            else:
                ops.awq_gemm_partial(act, w.packed, w.scales, w.qzeros, P, xs_a)
        return hidden, residual

    def _extend(self, input_ids, positions, out_cache_loc, qo_indptr, kv_indptr, kv_indices, max_extend_len):
        c = self.cfg
        T = input_ids.shape[0]
        hidden, residual = self.hidden[:T], self.residual[:T]
        qkv, attn_out = self.qkv[:T], self.attn_out[:T]
        gate_up, act = self.gate_up[:T], self.act[:T]
        torch.index_select(self.dec.embed, 0, input_ids, out=hidden)
        sm_scale = 1.0 / math.sqrt(c.head_dim)
        q = qkv[:, : self.q_size].view(T, c.q_heads, c.head_dim)
        k = qkv[:, self.q_size: self.q_size + self.kv_size].view(T, c.kv_heads, c.head_dim)
        v = qkv[:, self.q_size + self.kv_size:].view(T, c.kv_heads, c.head_dim)
        o = attn_out.view(T, c.q_heads, c.head_dim)
        self._pipe_begin(T)
        try:
            self._layers(hidden, residual, qkv, attn_out, gate_up, act, q, k, v, o, positions, out_cache_loc, qo_indptr,
                         kv_indptr, kv_indices, max_extend_len, sm_scale)
        finally:
            self._pipe = None
        return hidden, residual

    def _layers(self, hidden, residual, qkv, attn_out, gate_up, act, q, k, v, o, positions, out_cache_loc, qo_indptr,
                kv_indptr, kv_indices, max_extend_len, sm_scale):
        c = self.cfg
        for li, L in enumerate(self.dec.layers):
            if li == 0:
                residual.copy_(hidden)
                ops.rmsnorm(hidden, hidden, L["ln1"], c.eps)
            else:
                ops.fused_add_rmsnorm(hidden, residual, L["ln1"], c.eps)
            self._linear(L["qkv"], hidden, qkv)
            kb, vb = self.kv.get_kv_buffer(li)
            ops.rope_and_store_kv(qkv, self.dec.cos_sin, positions, kb, vb, out_cache_loc, c.q_heads, c.kv_heads, c.head_dim)
            ops.extend_attention_fwd(q, k, v, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None, True, None, max_extend_len,
                                     sm_scale)
            self._linear(L["o"], attn_out, hidden)
            ops.fused_add_rmsnorm(hidden, residual, L["ln2"], c.eps)
            self._linear(L["gate_up"], hidden, gate_up)
            ops.act_and_mul(act, gate_up, 0)
            self._linear(L["down"], act, hidden)

    def first_token(self, hidden: torch.Tensor, residual: torch.Tensor, last_row: int) -> torch.Tensor:
        """Greedy next token of the request whose last new token is row `last_row` (logits_processor.py: only the last
        position of an extend is projected)."""
        c = self.cfg
        h = hidden[last_row: last_row + 1].clone()
        r = residual[last_row: last_row + 1].clone()
        ops.fused_add_rmsnorm(h, r, self.dec.final_norm, c.eps)
        lm = self.dec.lm_head
        if lm.wpacked is not None:
            ops.dense_gemm_packed(h, lm.wpacked, None, self.logits)
        else:
            ops.dense_gemm_nt(h, lm.weight, None, self.logits)
        ops.argmax(self.logits, self.next_id)
        return self.next_id


class _Req:
    """The fields of managers/schedule_batch.py Req that the prefix cache reads."""

    def __init__(self, token_ids: List[int], user_id, text: str = ""):
        self.origin_input_ids = list(token_ids)
        self.output_ids: List[int] = []
        self.user_id = user_id
        self.origin_input_text = text
        self.req_pool_idx: Optional[int] = None
        self.prefix_indices = None
        self.last_node = None
        self.fill_ids = list(token_ids)


class PrefixCacheHarness:
    """One request at a time through match_prefix -> alloc -> extend -> first token -> cache_finished_req."""

    def __init__(self, dec: LlamaDecoder, pool_tokens: int, max_ctx: int, judge=None, compat_fork="fixed",
                 use_graph: bool = True, resident_bf16: bool = False, prefill_gemm: str = "library",
                 overlap_dequant: bool = False):
        c = dec.cfg
        dev = dec.device
        self.dec = dec
        self.kv = MHATokenToKVPool(pool_tokens, 1, dec.dtype, c.kv_heads, c.head_dim, c.layers, dev)
        self.alloc = TokenToKVPoolAllocator(pool_tokens, dec.dtype, dev, self.kv)
        self.req_pool = ReqToTokenPool(8, max_ctx, dev)
        self.tree = RadixCache(self.req_pool, self.alloc, 1, judge, compat_fork=compat_fork)
        self.model = LlamaPrefill(dec, self.kv, max_ctx, use_graph, resident_bf16, prefill_gemm, overlap_dequant)
        self.dev = dev

    def serve_first_token(self, token_ids: List[int], user_id, text: str = "", finish: bool = True):
        """Returns (first token id tensor, number of prefix tokens served from the cache, the request)."""
        req = _Req(token_ids, user_id, text)
        n = len(token_ids)
        # the last token is always computed (schedule_batch.py init_next_round_input matches fill_ids[:-1])
        prefix, last_node = self.tree.match_prefix(token_ids[:-1], user_id=user_id)
        pre = int(prefix.numel())
        req.prefix_indices, req.last_node = prefix, last_node
        self.tree.inc_lock_ref(last_node)
        req.req_pool_idx = self.req_pool.alloc(1)[0]
        ext = n - pre
        loc = self.alloc.alloc(ext)
        if loc is None:
            self.tree.evict(ext)
            loc = self.alloc.alloc(ext)
            if loc is None:
                raise RuntimeError("KV pool exhausted")
        row = self.req_pool.req_to_token[req.req_pool_idx]
        if pre:
            row[:pre] = prefix.to(torch.int32)
        row[pre:n] = loc.to(torch.int32)
        ids = torch.tensor(token_ids[pre:], dtype=torch.int64, device=self.dev)
        positions = torch.arange(pre, n, dtype=torch.int64, device=self.dev)
        qo_indptr = torch.tensor([0, ext], dtype=torch.int32, device=self.dev)
        kv_indptr = torch.tensor([0, pre], dtype=torch.int32, device=self.dev)
        kv_indices = row[:pre].contiguous() if pre else torch.zeros(1, dtype=torch.int32, device=self.dev)
        hidden, residual = self.model.extend(ids, positions, loc, qo_indptr, kv_indptr, kv_indices, ext)
        tok = self.model.first_token(hidden, residual, ext - 1)
        if finish:
            self.finish(req, int(tok.item()))
        return tok, pre, req

    def finish(self, req: _Req, first_token: int):
        req.output_ids = [first_token]
        self.tree.cache_finished_req(req)


def ttft_experiment(dec: LlamaDecoder, prefix_len: int, suffix_len: int, pool_tokens: int = 16384, seed: int = 0,
                    repeats: int = 3, resident_bf16: bool = False) -> dict:
    """Time to first token of a request whose first `prefix_len` tokens another user has just sent, with the cached
    nodes public and with SafeKV holding them private, next to the cold request and to the owner's own re-use.
    Wall-clock per request, host work included (match_prefix, allocation, metadata), device synchronised on both
    sides; the median of `repeats` rounds, each on a freshly reset tree."""
    import statistics
    import time

    from .private_service.private_client import PrivateJudgeClient

    g = torch.Generator().manual_seed(seed)
    vocab = dec.cfg.vocab
    rand = lambda n: torch.randint(0, vocab, (n,), generator=g).tolist()
    out = {}
    for mode, is_private in (("public", False), ("private", True)):
        judge = PrivateJudgeClient(detector=lambda text, p=is_private: p, num_workers=1)
        h = PrefixCacheHarness(dec, pool_tokens, prefix_len + suffix_len + 8, judge, resident_bf16=resident_bf16)
        cold, other, owner, hits = [], [], [], []
        for r in range(repeats + 1):                     # round 0 warms kernels, GEMM heuristics and the allocator
            h.tree.reset()
            h.alloc.clear()
            h.req_pool.clear()
            shared = rand(prefix_len)

            def timed(tokens, user):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, pre, _ = h.serve_first_token(tokens, user, "context line\nUser: question")
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) * 1e3, pre

            t_cold, pre_cold = timed(shared + rand(suffix_len), "victim")
            judge.drain()
            t_owner, pre_owner = timed(shared + rand(suffix_len), "victim")
            judge.drain()
            t_other, pre_other = timed(shared + rand(suffix_len), "attacker")
            judge.drain()
            if r:
                cold.append(t_cold)
                other.append(t_other)
                owner.append(t_owner)
                hits.append((pre_cold, pre_other, pre_owner))
        judge.close()
        med = statistics.median
        out[mode] = {"ttft_ms_cold": round(med(cold), 3), "ttft_ms_other_user": round(med(other), 3),
                     "ttft_ms_owner": round(med(owner), 3), "cached_prefix_tokens_cold_other_owner": list(hits[-1])}
        del h
    # the cold request once more (a) with this package's own large-batch GEMMs instead of dequantise + library GEMM and
    # (b) with a resident dequantised copy of the weights for the library GEMM (no per-call dequantisation pass; 2 bytes
    # per weight more memory, which 288 GB affords)
    def cold_only(**kw):
        h = PrefixCacheHarness(dec, pool_tokens, prefix_len + suffix_len + 8, None, **kw)
        ts = []
        for r in range(repeats + 1):
            h.tree.reset()
            h.alloc.clear()
            h.req_pool.clear()
            toks = rand(prefix_len + suffix_len)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h.serve_first_token(toks, "victim", "context line\nUser: question")
            torch.cuda.synchronize()
            if r:
                ts.append((time.perf_counter() - t0) * 1e3)
        del h
        return round(statistics.median(ts), 3)

    out["ttft_ms_cold_hand_written_gemm"] = cold_only(prefill_gemm="fused")
    if dec.cfg.quant == "awq" and not resident_bf16:
        out["ttft_ms_cold_dequant_overlapped"] = cold_only(overlap_dequant=True)
    if dec.cfg.quant == "awq" and not resident_bf16:
        free, _ = torch.cuda.mem_get_info(dec.device)
        c = dec.cfg
        per_layer = 2 * (c.hidden * (dec.q_size + 2 * dec.kv_size) + dec.q_size * c.hidden + 3 * c.hidden * c.inter)
        if free > 2 * per_layer * c.layers:
            out["ttft_ms_cold_resident_bf16"] = cold_only(resident_bf16=True)
            torch.cuda.empty_cache()
    pub, prv = out["public"], out["private"]
    out["ttft_speedup_prefix_hit"] = round(pub["ttft_ms_cold"] / pub["ttft_ms_other_user"], 2)
    out["side_channel_gap_ms_public"] = round(pub["ttft_ms_cold"] - pub["ttft_ms_other_user"], 3)
    out["side_channel_gap_ms_safekv_private"] = round(prv["ttft_ms_cold"] - prv["ttft_ms_other_user"], 3)
    return out
