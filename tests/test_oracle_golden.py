"""CPU: pin the C oracle (oracle/oracle.c) against the committed golden vectors, which were produced by the
reference itself (Triton kernels under the CPU interpreter, compiled sgl-kernel CPU kernels, and the torch
reference functions inside the reference's kernel tests) -- see oracle/gen_golden.py."""
import json
import os

import pytest
import torch

import oracle
from conftest import GOLDEN, load_golden


@pytest.mark.parametrize("name", ["kv_indices_small", "kv_indices_ragged"])
def test_kv_indices(name):
    g = load_golden(name)
    out = oracle.create_kv_indices(g["req_to_token"], g["req_pool_indices"], g["lens"], g["kv_indptr"])
    assert torch.equal(out, g["expect"])
    out = oracle.create_kv_indices(g["req_to_token"], g["req_pool_indices"], g["lens"], g["kv_indptr"], g["kv_start"])
    assert torch.equal(out, g["expect_start"])


def test_num_kv_splits():
    for c in json.load(open(os.path.join(GOLDEN, "num_kv_splits.json"))):
        out = oracle.get_num_kv_splits(torch.tensor(c["seq_lens"]), 1, c["num_head"], c["num_kv_head"],
                                       c["max_kv_splits"], c["device_core_count"])
        assert out.tolist() == c["expect"]


@pytest.mark.parametrize("case", ["gqa128", "mha64", "mqa_cap", "mla", "odd", "shared", "gqa_q05"])
def test_decode_attention(case):
    g = load_golden("decode_" + case)
    o = oracle.decode_attention(g["q"], g["k_buffer"], g["v_buffer"], g["kv_indptr"], g["kv_indices"],
                                g["num_kv_splits"], g["sm_scale"], g["logit_cap"])
    # Triton kernel, fp32 arithmetic on the same (bf16-valued) inputs
    assert (o - g["expect_triton_f32"]).abs().max().item() < 2e-5
    # compiled reference CPU kernel, bf16 output: the reference's own tolerance (test_decode.py:136-140)
    torch.testing.assert_close(o, g["expect_cpu_bf16"].float(), atol=3e-2, rtol=1e-6)


@pytest.mark.parametrize("case", ["gqa128", "mha64_cap"])
def test_extend_attention(case):
    g = load_golden("extend_" + case)
    o = oracle.extend_attention(g["q_extend"], g["k_extend"], g["v_extend"], g["k_buffer"], g["v_buffer"],
                                g["qo_indptr"], g["kv_indptr"], g["kv_indices"], g["sm_scale"], g["logit_cap"])
    assert (o - g["expect_triton_f32"]).abs().max().item() < 2e-5
    torch.testing.assert_close(o, g["expect_cpu_bf16"].float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("name", ["merge_state_bf16", "merge_state_f16", "merge_state_f32"])
def test_merge_state(name):
    g = load_golden(name)
    vo, so = oracle.merge_state(g["v_a"], g["s_a"], g["v_b"], g["s_b"])
    torch.testing.assert_close(vo.to(g["expect_v"].dtype).float(), g["expect_v"].float(), atol=1e-3, rtol=1e-2)
    torch.testing.assert_close(so, g["expect_s"], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("name", ["awq_g128_bf16", "awq_g128_f16", "awq_gall_f16", "awq_g64_bf16"])
def test_awq(name):
    g = load_golden(name)
    w = oracle.awq_dequantize(g["qweight"], g["scales"], g["qzeros"])
    assert torch.equal(w.view(torch.int16), g["expect"].view(torch.int16))  # bit-exact
    y = oracle.half_matmul(g["x"], w).float()
    want = g["expect_matmul_f32"]
    assert (y - want).abs().max().item() <= 1e-2 * want.abs().max().item()


@pytest.mark.parametrize("name", ["norm_f16", "norm_bf16", "norm_odd"])
def test_norm(name):
    g = load_golden(name)
    y = oracle.rmsnorm(g["x"], g["weight"], g["eps"])
    torch.testing.assert_close(y.float(), g["expect_rmsnorm"].float(), rtol=1e-3, atol=1e-3)
    y2, r2 = oracle.fused_add_rmsnorm(g["x"], g["residual"], g["weight"], g["eps"])
    torch.testing.assert_close(y2.float(), g["expect_fused_x"].float(), rtol=1e-3, atol=1e-3)
    assert torch.equal(r2.view(torch.int16), g["expect_fused_residual"].view(torch.int16))
    if "expect_rmsnorm_cpu" in g:
        torch.testing.assert_close(y.float(), g["expect_rmsnorm_cpu"].float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("name", ["silu_f16", "silu_bf16"])
def test_silu(name):
    g = load_golden(name)
    y = oracle.silu_and_mul(g["x"])
    torch.testing.assert_close(y.float(), g["expect"].float(), rtol=1e-3, atol=1e-3)
    if "expect_cpu" in g:
        torch.testing.assert_close(y.float(), g["expect_cpu"].float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("name", ["rope_neox128", "rope_gptj128", "rope_partial"])
def test_rope(name):
    g = load_golden(name)
    q = oracle.rope(g["q"], g["positions"], g["cos_sin_cache"], g["head_size"], bool(g["is_neox"]))
    k = oracle.rope(g["k"], g["positions"], g["cos_sin_cache"], g["head_size"], bool(g["is_neox"]))
    assert torch.equal(q.view(torch.int16), g["expect_q"].view(torch.int16))
    assert torch.equal(k.view(torch.int16), g["expect_k"].view(torch.int16))


def test_fp8_codec_matches_torch():
    """The oracle's e4m3fn codec against torch's cast over every code and a dense sweep of floats."""
    codes = torch.arange(256, dtype=torch.uint8)
    want = codes.view(torch.float8_e4m3fn).float()
    got = oracle.e4m3_decode(codes.view(torch.float8_e4m3fn))
    ok = ~torch.isnan(want)
    assert torch.equal(got[ok], want[ok]) and torch.isnan(got[~ok]).all()
    x = torch.cat([torch.linspace(-460, 460, 20001), torch.randn(20000) * 0.01, torch.randn(20000) * 3])
    enc = oracle.e4m3_encode(x).view(torch.uint8)
    ref = x.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(enc, ref)


@pytest.mark.parametrize("name", ["fp8_mm_bf16", "fp8_mm_f16"])
def test_fp8_scaled_mm(name):
    g = load_golden(name)
    out = oracle.fp8_scaled_mm(g["a"], g["b_nk"].t(), g["scale_a"], g["scale_b"], g["expect"].dtype, g.get("bias"))
    torch.testing.assert_close(out.float(), g["expect"].float(), rtol=0.02, atol=1)


def test_fp8_blockwise_mm():
    g = load_golden("fp8_block_mm")
    out = oracle.fp8_blockwise_mm(g["a"], g["b_nk"], g["scale_a"], g["scale_b_kn"].t().contiguous(), torch.bfloat16)
    torch.testing.assert_close(out.float(), g["expect"].float(), rtol=0.02, atol=1)


def test_per_token_quant_fp8():
    g = load_golden("fp8_per_token_quant")
    q, s = oracle.per_token_quant_fp8(g["x"])
    torch.testing.assert_close(s.reshape(-1), g["expect_scale"], rtol=1e-6, atol=0)
    torch.testing.assert_close(q.float(), g["expect_q"].float(), rtol=1e-3, atol=1e-3)


def test_topk_softmax_oracle_vs_torch():
    """The reference's own criterion for topk_softmax (sgl-kernel/tests/test_moe_topk_softmax.py:37-48)."""
    import oracle

    g = torch.Generator().manual_seed(3)
    for T, E, k in [(1, 4, 1), (16, 64, 4), (128, 256, 8), (33, 384, 2)]:
        gating = torch.randn(T, E, generator=g)
        w, ids, src = oracle.topk_softmax(gating, k)
        rw, ri = torch.topk(torch.softmax(gating, dim=-1), k, dim=-1)
        assert torch.equal(ids, ri.int())
        torch.testing.assert_close(w, rw, rtol=1e-5, atol=1e-7)
        assert torch.equal(src, (torch.arange(k)[None, :] * T + torch.arange(T)[:, None]).int())


def test_moe_align_oracle_properties():
    """moe_align_kernel.cu:43-100: padded prefix sums, one expert per block, every slot placed once."""
    import oracle

    g = torch.Generator().manual_seed(5)
    for T, k, E, bs in [(1, 2, 8, 16), (50, 8, 256, 64), (300, 4, 60, 16)]:
        ids = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)])
        srt, ex, tot = oracle.moe_align_block_size(ids, E, bs)
        counts = torch.bincount(ids.flatten(), minlength=E)
        assert tot == int(((counts + bs - 1) // bs * bs).sum())
        flat = ids.flatten()
        real = srt[:tot][srt[:tot] < flat.numel()]
        assert torch.equal(torch.sort(real).values, torch.arange(flat.numel(), dtype=torch.int32))
        for blk in range(tot // bs):
            seg = srt[blk * bs:(blk + 1) * bs]
            seg = seg[seg < flat.numel()]
            assert (flat[seg.long()] == int(ex[blk])).all()
            assert torch.equal(seg, torch.sort(seg).values)


# ---------------------------------------------------------------------------------------------- draft-tree helpers
def _tree_golden():
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eagle_tree.json")))
    t = lambda e: torch.tensor(e["data"], dtype=getattr(torch, e["dtype"]))
    return g, t


def test_oracle_build_tree_golden():
    """The serial restatement reproduces the expected outputs of build_eagle_tree.py:354-384."""
    import oracle
    g, t = _tree_golden()
    n = g["num_draft_token"]
    parent_list, top, draft = oracle.build_tree_preprocess(
        t(g["verified_id"]), [t(x) for x in g["score_list"]], [t(x) for x in g["token_list"]],
        [t(x) for x in g["parents_list"]], n)
    _, pos, ridx, nxt, sib = oracle.build_tree_efficient(parent_list, top, t(g["seq_lens"]), g["topk"], g["depth"], n)
    e = g["expected"]
    assert draft.tolist() == e["draft_tokens"]
    assert pos == e["position"] and ridx == e["retrive_index"]
    assert nxt == e["retrive_next_token"] and sib == e["retrive_next_sibling"]


def _walk_golden():
    cand = torch.tensor([[0, 1, 2, 3, 4, 5], [7, 8, 9, 10, 11, 12]], dtype=torch.int32)
    ridx = torch.tensor([[0, 1, 2, 3, 4, 5], [6, 7, 8, 9, 10, 11]], dtype=torch.int32)
    nxt = torch.tensor([[1, 2, -1, 4, 5, -1], [4, 2, 3, -1, 5, -1]], dtype=torch.int32)
    sib = torch.tensor([[-1, 3, -1, -1, -1, -1], [-1, -1, -1, -1, 1, -1]], dtype=torch.int32)
    logits = torch.full((2, 6, 20), 1, dtype=torch.float32)
    for i, j, k in ((0, 0, 3), (0, 3, 4), (0, 4, 5), (1, 0, 11), (1, 4, 12)):
        logits[i, j, k] = 10
    for i in range(2):
        for j in range(6):
            if torch.max(logits[i, j]) < 10:
                logits[i, j, 18] = 10
    return cand, ridx, nxt, sib, logits


def test_oracle_verify_tree_greedy_golden():
    """sgl-kernel/tests/speculative/test_eagle_utils.py:78-84."""
    import oracle
    cand, ridx, nxt, sib, logits = _walk_golden()
    got = oracle.verify_tree_greedy(cand, ridx, nxt, sib, torch.argmax(logits, -1).to(torch.int32), 4)
    assert got == ([3, -1, -1, 4, 5, 18, 11, -1, -1, -1, 12, 18], [[0, 3, 4, 5], [6, 10, 11, -1]], [3, 2])


def test_oracle_tree_speculative_sampling_golden():
    """sgl-kernel/tests/speculative/test_speculative_sampling.py:6-21."""
    import oracle
    cand, ridx, nxt, sib, logits = _walk_golden()
    probs = torch.softmax(logits / 0.01, dim=-1)
    coins = torch.rand(2, 6, generator=torch.Generator().manual_seed(0))
    got = oracle.tree_speculative_sampling_target_only(cand, ridx, nxt, sib, coins, probs, torch.zeros_like(probs), 4, 1, 1)
    assert got == ([3, -1, -1, 4, 5, 18, 11, -1, -1, -1, 12, 18], [[0, 3, 4, 5], [6, 10, 11, -1]], [3, 2])
    got = oracle.tree_speculative_sampling_target_only(cand, ridx, nxt, sib, coins, probs, torch.zeros_like(probs), 4, 0, 0)
    assert got == ([1, 2, 18, -1, -1, -1, 11, -1, -1, -1, 12, 18], [[0, 1, 2, -1], [6, 10, 11, -1]], [2, 2])


def test_oracle_gptq_dequantize_known_answer():
    """The format by hand: word 0x76543210 holds rows 0..7 of a column, zero nibbles are stored minus one (v1)."""
    import oracle
    qw = torch.full((1, 8), 0x76543210, dtype=torch.int32)                 # K = 8, N = 8
    qz = torch.tensor([[0x01234567]], dtype=torch.int32)                   # column j stores 7 - j -> zero point 8 - j
    sc = torch.full((1, 8), 0.5, dtype=torch.float16)
    w = oracle.gptq_dequantize(qw, qz, sc, 8)
    k = torch.arange(8).view(8, 1).float()
    z = (8 - torch.arange(8)).view(1, 8).float()
    assert torch.equal(w.float(), (k - z) * 0.5)
    assert torch.equal(oracle.gptq_dequantize(qw, qz, sc, 8, zero_offset=0).float(), (k - z + 1) * 0.5)


@pytest.mark.parametrize("case", ["tree", "prefix_masked"])
def test_oracle_extend_masked_golden(golden, case):
    """The masked-extend restatement against the reference Triton kernel's output (interpreter run, gen_golden.py)."""
    import oracle
    g = golden("extend_masked_" + case)
    got = oracle.extend_attention_masked(g["q_extend"], g["k_extend"], g["v_extend"], g["k_buffer"], g["v_buffer"],
                                         g["qo_indptr"], g["kv_indptr"], g["kv_indices"], g["custom_mask"],
                                         g["mask_indptr"], bool(g["skip_prefix_custom_mask"]), g["sm_scale"])
    torch.testing.assert_close(got, g["expect_triton_f32"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("case", ["w8", "w4_h64"])
def test_oracle_extend_window_golden(golden, case):
    """The sliding-window restatement against the reference Triton kernel (interpreter run, gen_golden.py), called as
    the reference's backend calls it (only the window's prefix tokens listed).  Rows whose whole listed prefix lies
    outside the window come out NaN in the reference (its running maximum stays -inf through the prefix stage and
    exp(-inf - -inf) poisons the row); the restatement gives those rows the attention over their new tokens, and only
    the rows the reference gets finite are compared."""
    import oracle
    g = golden("extend_window_" + case)
    got = oracle.extend_attention_window(g["q_extend"], g["k_extend"], g["v_extend"], g["k_buffer"], g["v_buffer"],
                                         g["qo_indptr"], g["kv_indptr"], g["kv_indices"], int(g["window"]), g["sm_scale"])
    want = g["expect_triton_f32"]
    ok = ~torch.isnan(want).any(-1).any(-1)
    assert 0 < int(ok.sum()) < want.shape[0]
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got[ok], want[ok], rtol=2e-4, atol=2e-4)


def test_oracle_moe_fused_gate_matches_torch_formula():
    """Without ties (fp32) the kernel restatement equals biased_grouped_topk_impl (topk.py:170-211), restated inline."""
    import oracle
    n, E, ng, tg, k = 200, 256, 8, 4, 8
    g = torch.Generator().manual_seed(0)
    x, b = torch.randn(n, E, generator=g), torch.rand(E, generator=g)
    scores = x.sigmoid()
    choice = scores + b
    gs = choice.view(n, ng, -1).topk(2, dim=-1)[0].sum(-1)
    gmask = torch.zeros_like(gs).scatter_(1, torch.topk(gs, tg, dim=-1)[1], 1)
    smask = gmask.unsqueeze(-1).expand(n, ng, E // ng).reshape(n, -1)
    ids = torch.topk(choice.masked_fill(~smask.bool(), float("-inf")), k, dim=-1)[1]
    w = scores.gather(1, ids)
    w = w / w.sum(-1, keepdim=True)
    out, idx = oracle.moe_fused_gate(x, b, k, ng, tg)
    assert torch.equal(idx.long(), ids)                       # torch.topk returns them best first as well
    torch.testing.assert_close(out, w, rtol=1e-5, atol=1e-7)
    out, idx = oracle.moe_fused_gate(x, b, k + 1, ng, tg, 4, 2.5)
    assert torch.equal(idx[:, -1].long(), E + torch.arange(n) % 4)
    torch.testing.assert_close(out[:, -1], torch.full((n,), 0.4))


@pytest.mark.parametrize("fmt,dt", [("e4m3fn", torch.float8_e4m3fn), ("e5m2", torch.float8_e5m2),
                                    ("e5m2fnuz", torch.float8_e5m2fnuz)])
def test_oracle_kv_fp8_against_torch_conversions(fmt, dt):
    """The fp8 KV cache is `cache_k.to(float8 type)` on the way in and `.to(q.dtype)` on the way out
    (mem_cache/memory_pool.py:389-400): torch's conversions are the reference's arithmetic.  The integer restatement in
    oracle/ agrees with them on every byte, every bf16 and every fp16 bit pattern, and on a wide fp32 sample."""
    import numpy as np

    import oracle

    codes = torch.arange(256, dtype=torch.uint8)
    want = codes.view(dt).float().numpy()
    got = oracle.kv_fp8_decode(codes.numpy(), fmt)
    assert np.array_equal(np.isnan(want), np.isnan(got))
    ok = ~np.isnan(want)
    assert np.array_equal(want[ok].view(np.uint32), got[ok].view(np.uint32))   # signed zeros included
    pats = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    for src in (torch.bfloat16, torch.float16):
        x = pats.view(src)
        assert np.array_equal(x.to(dt).view(torch.uint8).numpy(), oracle.kv_fp8_encode(x.float().numpy(), fmt))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 18, generator=g) * torch.exp(torch.randn(1 << 18, generator=g) * 6)
    assert np.array_equal(x.to(dt).view(torch.uint8).numpy(), oracle.kv_fp8_encode(x.numpy(), fmt))
    # a scaled write divides in the rows' own type first (product with the fp32 reciprocal, as the device does)
    rows = (torch.randn(64, 32, generator=g) * 4).to(torch.bfloat16)
    inv = np.float32(1.0) / np.float32(0.37)
    want8 = torch.from_numpy(rows.float().numpy() * inv).to(torch.bfloat16).to(dt).view(torch.uint8).numpy()
    assert np.array_equal(oracle.set_kv_buffer_fp8(rows, 0.37, fmt), want8)
