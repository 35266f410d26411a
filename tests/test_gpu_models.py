"""GPU: whole decode steps of the model shapes the benchmark's configurations name, against a plain fp32 torch
restatement of the same model (rounded to bf16 where the kernel path stores bf16).

Qwen2 shape (BASELINE configuration 1: head size 64 -> the generic attention kernels, bias on the fused q/k/v
projection, models/qwen2.py:133-139) and the Llama shape with bf16 weights; tokens are teacher-forced so that a
near-tie in an arg-max cannot send the two computations down different paths."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r(x):
    return x.to(torch.bfloat16).float()


def _ref_logits(dec, ids):
    """ids [B, T]: logits of the last position, every sequence computed from scratch with causal attention."""
    c = dec.cfg
    B, T = ids.shape
    half = c.head_dim // 2
    cs = dec.cos_sin[torch.arange(T, device=DEV)]
    cos, sin = cs[:, :half], cs[:, half:]
    out = []

    def norm(v, w):
        return _r(v * torch.rsqrt((v * v).mean(-1, keepdim=True) + c.eps) * w.float())

    def rope(t):
        a, b = t[..., :half], t[..., half:]
        return _r(torch.cat([a * cos[:, None] - b * sin[:, None], b * cos[:, None] + a * sin[:, None]], -1))

    for bi in range(B):
        x = dec.embed[ids[bi]].float()
        res = None
        for li, L in enumerate(dec.layers):
            res = x if li == 0 else _r(res + x)
            h = norm(res, L["ln1"])
            qkv = h @ L["qkv"].weight.float().t()
            if L["qkv_bias"] is not None:
                qkv = qkv + L["qkv_bias"].float()
            qkv = _r(qkv)
            q = rope(qkv[:, : dec.q_size].view(T, c.q_heads, c.head_dim))
            k = rope(qkv[:, dec.q_size: dec.q_size + dec.kv_size].view(T, c.kv_heads, c.head_dim))
            v = qkv[:, dec.q_size + dec.kv_size:].view(T, c.kv_heads, c.head_dim)
            rep = c.q_heads // c.kv_heads
            kk, vv = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
            s = torch.einsum("thd,shd->hts", q, kk) / math.sqrt(c.head_dim)
            s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool, device=DEV), 1), float("-inf"))
            o = _r(torch.einsum("hts,shd->thd", torch.softmax(s, -1), vv)).reshape(T, -1)
            x = _r(o @ L["o"].weight.float().t())
            res = _r(res + x)
            h = norm(res, L["ln2"])
            gu = _r(h @ L["gate_up"].weight.float().t())
            act = _r(_r(torch.nn.functional.silu(gu[:, : c.inter])) * gu[:, c.inter:])
            x = _r(act @ L["down"].weight.float().t())
        final = norm(_r(res + x), dec.final_norm)
        out.append(final[-1])
    hidden = torch.stack(out)
    eye = torch.eye(c.hidden, dtype=torch.bfloat16, device=DEV)
    from sglang_security_amd import ops

    lm_t = ops.dense_gemm_packed(eye, dec.lm_head.wpacked).float() if dec.lm_head.wpacked is not None \
        else dec.lm_head.weight.float().t()
    return _r(hidden @ lm_t), hidden


@pytest.mark.parametrize("shape", ["tiny_qwen", "tiny"])
def test_decode_steps_match_torch(shape):
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = getattr(LlamaConfig, shape)("bf16")
    B, T = 3, 9
    dec = LlamaDecoder(cfg, batch=B, pool_slots=1 + B * 32, max_ctx=32, device=DEV, seed=1)
    dec.setup_shared_prefix_batch(prefix_len=0, suffix_len=0, share=False)   # nothing cached yet
    ids = torch.randint(0, cfg.vocab, (B, T), generator=torch.Generator().manual_seed(7)).to(DEV)
    for t in range(T):
        dec.input_ids.copy_(ids[:, t])
        dec.step()
    torch.cuda.synchronize()
    want_logits, want_hidden = _ref_logits(dec, ids)
    got_hidden = dec.hidden.float()
    rel = ((got_hidden - want_hidden).norm() / want_hidden.norm()).item()
    assert rel < 2e-2, rel
    scale = want_logits.abs().max().item()
    assert (dec.logits.float() - want_logits).abs().max().item() <= 3e-2 * max(scale, 1.0)
    # the rotated keys of layer 0 sit in the pool rows the requests were given
    assert int(dec.seq_lens[0]) == T


def test_qwen2_0_5b_shape_one_step():
    """The full Qwen2-0.5B shape (24 layers, hidden 896, 14 / 2 heads of 64, vocabulary 151936), batch 1 with 128
    cached tokens: one step runs, eager and as a graph, with finite logits and the same token both ways."""
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = LlamaConfig.qwen2_0_5b("bf16")
    dec = LlamaDecoder(cfg, batch=1, pool_slots=1 + 160, max_ctx=160, device=DEV, seed=0)
    dec.setup_shared_prefix_batch(prefix_len=0, suffix_len=128, share=False)
    first = dec.input_ids.clone()
    dec.step()
    torch.cuda.synchronize()
    tok_eager = dec.next_ids.clone()
    assert torch.isfinite(dec.logits.float()).all() and 0 <= int(tok_eager) < cfg.vocab
    dec.setup_shared_prefix_batch(prefix_len=0, suffix_len=128, share=False)
    dec.input_ids.copy_(first)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            dec.step()
    dec.setup_shared_prefix_batch(prefix_len=0, suffix_len=128, share=False)
    dec.input_ids.copy_(first)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(dec.next_ids, tok_eager)
