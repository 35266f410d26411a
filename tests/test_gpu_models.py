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


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("shape", ["tiny_qwen", "tiny"])
def test_decode_steps_match_torch(shape, fused):
    """fused: the 16-bit step over the streaming weight layout with the split sums folded into their consumers (head
    size 64 with a qkv bias included), where the shape allows it."""
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = getattr(LlamaConfig, shape)("bf16")
    B, T = 3, 9
    dec = LlamaDecoder(cfg, batch=B, pool_slots=1 + B * 32, max_ctx=32, device=DEV, seed=1)
    if fused and not dec.enable_fusion(True):
        pytest.skip("shape outside the fused 16-bit step")
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


def _q_group(x):
    """Per-token, per-128-group e4m3 quantisation, as a float tensor (what the block-fp8 linear multiplies)."""
    M, K = x.shape
    g = x.float().view(M, K // 128, 128)
    s = (g.abs().amax(-1, keepdim=True) / 448.0).clamp(min=1e-10)
    return ((g / s).clamp(-448, 448).to(torch.float8_e4m3fn).float() * s).view(M, K)


def _q_tensor(x):
    s = x.float().abs().max() / 448.0
    return (x.float() / s).clamp(-448, 448).to(torch.float8_e4m3fn).float() * s


def test_deepseek_v3_dense_layers_match_torch():
    """Absorbed-MLA + block-fp8 dense layers (BASELINE configuration 5's layer type) against a torch restatement with
    the same quantisation points (fp32 products of the dequantised operands), two layers, two steps; the latent
    rows the steps append to the pool must be the restatement's."""
    from sglang_security_amd.mla_runner import DeepseekV3Config, DeepseekV3DenseDecoder

    cfg = DeepseekV3Config.tiny()
    B, prefix, suffix = 4, 40, 6
    dec = DeepseekV3DenseDecoder(cfg, B, pool_slots=1 + prefix + B * 16, max_ctx=64, device=DEV, seed=0)
    dec.setup_shared_prefix_batch(prefix, suffix)
    c = cfg
    H, lat, nope, rd = c.heads, c.kv_lora_rank, c.qk_nope_head_dim, c.qk_rope_head_dim

    def norm(v, w):
        v = v.float()
        return _r(v * torch.rsqrt((v * v).mean(-1, keepdim=True) + c.eps) * w.float())

    def rope_il(t, pos):                                    # [B, heads, 64], interleaved pairs
        cs = dec.cos_sin[pos]
        cos, sin = cs[:, None, : rd // 2], cs[:, None, rd // 2:]
        a, b = t[..., 0::2].float(), t[..., 1::2].float()
        out = torch.empty_like(t, dtype=torch.float32)
        out[..., 0::2] = a * cos - b * sin
        out[..., 1::2] = b * cos + a * sin
        return _r(out)

    def ref_step(x_in, pools, r2t, seq_lens):
        pos = seq_lens.clone()
        x, res = x_in.float(), None
        for li, L in enumerate(dec.layers):
            res = x if li == 0 else _r(res + x)
            h = norm(res, L["ln1"])
            qkv_a = _r(_q_group(h) @ L["qkv_a"].dequant().t())
            q_a = norm(qkv_a[:, : c.q_lora_rank], L["q_a_ln"])
            k_nope = norm(qkv_a[:, c.q_lora_rank: c.q_lora_rank + lat], L["kv_a_ln"])
            k_pe = qkv_a[:, c.q_lora_rank + lat:]
            q = _r(_q_group(q_a) @ L["q_b"].dequant().t()).view(B, H, nope + rd)
            w_kc = L["w_kc"].float() * L["w_scale"]
            w_vc = L["w_vc"].float() * L["w_scale"]
            q_nope_out = _r(torch.einsum("bhk,hnk->bhn", _q_tensor(q[..., :nope]), w_kc))
            q_pe = rope_il(q[..., nope:], pos)
            k_pe = rope_il(k_pe.view(B, 1, rd), pos).view(B, rd)
            q_in = torch.cat([q_nope_out, q_pe], -1)
            k_in = torch.cat([k_nope, k_pe], -1)
            outs = []
            for b in range(B):
                slots = r2t[b, : int(seq_lens[b])].long()
                rows = torch.cat([pools[li][slots, 0].float(), k_in[b: b + 1]], 0)           # [len + 1, 576]
                s = (q_in[b] @ rows.t()) * dec.sm_scale
                outs.append(_r(torch.softmax(s, -1) @ rows[:, :lat]))
                pools[li][int(r2t[b, int(seq_lens[b])]), 0] = k_in[b].to(torch.bfloat16)
            attn = torch.stack(outs)                                                         # [B, H, 512]
            o_heads = _r(torch.einsum("bhk,hnk->bhn", _q_tensor(attn), w_vc)).reshape(B, -1)
            x = _r(_q_group(o_heads) @ L["o"].dequant().t())
            res = _r(res + x)
            h = norm(res, L["ln2"])
            gu = _r(_q_group(h) @ L["gate_up"].dequant().t())
            act = _r(_r(torch.nn.functional.silu(gu[:, : c.inter])) * gu[:, c.inter:])
            x = _r(_q_group(act) @ L["down"].dequant().t())
        return norm(_r(res + x), dec.final_norm)

    pools = [p.clone() for p in dec.kv_pool]
    r2t = dec.req_to_token.clone()
    seq = dec.seq_lens.clone()
    gen = torch.Generator(device=DEV).manual_seed(3)
    for step in range(2):
        x_in = torch.randn(B, c.hidden, device=DEV, generator=gen).to(torch.bfloat16)
        want = ref_step(x_in, pools, r2t, seq)
        seq += 1
        dec.step(x_in)
        torch.cuda.synchronize()
        # the latent rows appended to the pool: layer 0's depend on one quantised product only (tight), the deeper
        # ones and the output on chains of fp8 roundings, where a one-ulp difference upstream can flip a 3-bit mantissa
        new_rows = r2t[torch.arange(B), (seq - 1)].long()
        for li in range(c.layers):
            a, b_ = dec.kv_pool[li][new_rows, 0].float(), pools[li][new_rows, 0].float()
            rel_kv = ((a - b_).norm() / b_.norm()).item()
            assert rel_kv < (1e-2 if li == 0 else 6e-2), (step, li, rel_kv)
        rel = ((dec.hidden.float() - want).norm() / want.norm()).item()
        assert rel < 6e-2, (step, rel)
        cosine = torch.nn.functional.cosine_similarity(dec.hidden.float().flatten(), want.flatten(), dim=0).item()
        assert cosine > 0.998, (step, cosine)
    assert dec.seq_lens.tolist() == seq.tolist()
