"""GPU parity: metadata kernels (bit-exact) and paged decode attention (fp tolerance) through the C ABI,
against the committed golden vectors (reference Triton / CPU kernels) and the C oracle on seeded inputs.

Tolerances are the reference's own: kv-index gather torch.equal (test/srt/test_create_kvindices.py:64);
decode attention atol 3e-2 + cosine > 0.99 on bf16 (test/srt/cpu/test_decode.py:136-140).  Against the
fp32 Triton vector the kernel is additionally held to 2e-2 absolute (bf16 output rounding of |o| <~ 4)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from sglang_security_amd import ops

    return ops


@pytest.mark.parametrize("name", ["kv_indices_small", "kv_indices_ragged"])
@pytest.mark.parametrize("idx_dtype", [torch.int32, torch.int64])
def test_create_kv_indices_golden(golden, name, idx_dtype):
    ops = _ops()
    g = golden(name)
    r2t = g["req_to_token"].to(DEV)
    idx = g["req_pool_indices"].to(idx_dtype).to(DEV)
    lens = g["lens"].to(idx_dtype).to(DEV)
    ptr = g["kv_indptr"].to(DEV)
    out = torch.full((int(g["kv_indptr"][-1]),), -1, dtype=torch.int32, device=DEV)
    ops.create_kv_indices(r2t, idx, lens, ptr, None, out)
    assert torch.equal(out.cpu(), g["expect"])
    out2 = torch.full_like(out, -1)
    ops.create_kv_indices(r2t, idx, lens, ptr, g["kv_start"].to(idx_dtype).to(DEV), out2)
    assert torch.equal(out2.cpu(), g["expect_start"])


@pytest.mark.parametrize("bs", [1, 37, 1786])
def test_create_kv_indices_reference_shapes(bs):
    """The reference's own test shape: 4096 x 4096 table, batch {1, 37, 1786} (test_create_kvindices.py:18-71)."""
    import oracle

    ops = _ops()
    torch.manual_seed(42)
    max_batch, max_ctx = 4096, 4096
    r2t = torch.arange(max_batch * max_ctx, dtype=torch.int32).reshape(max_batch, max_ctx)
    idx = torch.randperm(max_batch)[:bs].to(torch.int32)
    lens = torch.randint(1, max_ctx, (bs,), dtype=torch.int32)
    ptr = torch.zeros(bs + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(lens, 0)
    want = oracle.create_kv_indices(r2t, idx, lens, ptr)
    out = torch.empty(int(ptr[-1]), dtype=torch.int32, device=DEV)
    ptr_d = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    ops.kv_indptr_cumsum(lens.to(DEV), ptr_d)
    assert torch.equal(ptr_d.cpu(), ptr)
    ops.create_kv_indices(r2t.to(DEV), idx.to(DEV), lens.to(DEV), ptr_d, None, out)
    assert torch.equal(out.cpu(), want)


def test_num_kv_splits_golden():
    import json
    import os

    from conftest import GOLDEN

    ops = _ops()
    cases = json.load(open(os.path.join(GOLDEN, "num_kv_splits.json")))
    for c in cases:
        sl = torch.tensor(c["seq_lens"], dtype=torch.int64, device=DEV)
        out = torch.zeros(len(c["seq_lens"]), dtype=torch.int32, device=DEV)
        ops.get_num_kv_splits(out, sl, c["num_head"], c["num_kv_head"], c["max_kv_splits"], c["device_core_count"])
        assert out.cpu().tolist() == c["expect"], c


@pytest.mark.parametrize("bs,ctx", [(1, 7), (5, 300), (64, 1200), (200, 40)])
@pytest.mark.parametrize("ldt", [torch.int64, torch.int32])
def test_decode_metadata_one_launch_and_prepare(bs, ctx, ldt):
    """sglk_decode_metadata (one launch) against the three separate entry points, bit for bit; prepare_for_decode
    against the torch sequence of schedule_batch.py:1464-1527."""
    ops = _ops()
    g = torch.Generator().manual_seed(bs + ctx)
    lens = torch.randint(1, ctx + 1, (bs,), generator=g).to(ldt)
    r2t = torch.randint(1, 1 << 20, (bs + 3, ctx + 8), generator=g, dtype=torch.int32).to(DEV)
    pool_idx = torch.randperm(bs + 3, generator=g)[:bs].to(DEV)
    lens_d = lens.to(DEV)
    total = int(lens.sum())
    ptr = torch.full((bs + 1,), -1, dtype=torch.int32, device=DEV)
    ind = torch.full((total + 5,), -1, dtype=torch.int32, device=DEV)
    ns = torch.full((bs,), -1, dtype=torch.int32, device=DEV)
    ops.decode_metadata(r2t, pool_idx, lens_d, ptr, ind, ns, 32, 8, 8, 256)
    ptr2 = torch.zeros_like(ptr)
    ind2 = torch.full_like(ind, -1)
    ns2 = torch.zeros_like(ns)
    ops.kv_indptr_cumsum(lens_d, ptr2)
    ops.get_num_kv_splits(ns2, lens_d, 32, 8, 8, 256)
    ops.create_kv_indices(r2t, pool_idx, lens_d, ptr2, None, ind2)
    assert torch.equal(ptr, ptr2) and torch.equal(ns, ns2) and torch.equal(ind, ind2)
    # prepare_for_decode
    pos = torch.empty(bs, dtype=torch.int64, device=DEV)
    loc = torch.empty(bs, dtype=torch.int64, device=DEV)
    want_loc = r2t[pool_idx, lens_d.long()].long()
    lens_before = lens_d.clone()
    ops.prepare_for_decode(r2t, pool_idx, lens_d, pos, loc)
    assert torch.equal(pos, lens_before.long()) and torch.equal(lens_d, lens_before + 1) and torch.equal(loc, want_loc)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(64, 128256), (3, 1001), (1, 8), (7, 4096)])
def test_argmax_rows(dtype, rows, cols):
    ops = _ops()
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g).to(dtype).to(DEV)
    x[0, cols // 2] = 100.0
    if cols > 4:
        x[rows - 1, 3] = 50.0
        x[rows - 1, cols - 1] = 50.0          # a tie: the first one wins
    out = torch.empty(rows, dtype=torch.int64, device=DEV)
    ops.argmax(x, out)
    assert torch.equal(out, torch.argmax(x, dim=-1))
    if cols > 4 and rows > 1:
        assert int(out[rows - 1]) == 3
    view = x[:, : cols - 1] if cols > 8 else x  # a row stride that is not the row length
    out2 = torch.empty(rows, dtype=torch.int64, device=DEV)
    ops.argmax(view, out2)
    assert torch.equal(out2, torch.argmax(view, dim=-1))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_argmax_rows_special_values(dtype):
    """Rows the fast scan hands to the full rule, or that never beat the start value: NaNs (the first one wins, whatever
    its sign and whatever else the row holds), nothing but -inf (index 0), a maximum of zero met first as -0.0 and
    later as +0.0 (equal: the first stays), +inf twice."""
    ops = _ops()
    cols = 40000
    x = -torch.rand(8, cols, generator=torch.Generator().manual_seed(3)) - 1.0
    x[0, 777] = float("nan")
    x[0, 30001] = float("nan")
    x[0, 5] = float("inf")
    x[1, :] = float("-inf")
    x[2, 9000] = -0.0
    x[2, 20000] = 0.0
    x[3, 12345] = float("inf")
    x[3, 39999] = float("inf")
    x[4, cols - 1] = -float("nan")
    x[5, :] = float("-inf")
    x[5, 31111] = -3.0e38
    x = x.to(dtype).to(DEV)
    out = torch.empty(8, dtype=torch.int64, device=DEV)
    ops.argmax(x, out)
    assert out.tolist()[:6] == [777, 0, 9000, 12345, cols - 1, 31111]
    assert torch.equal(out[6:], torch.argmax(x[6:], dim=-1))


def _run_decode(ops, g, dtype, max_splits=8, splits=None):
    q = g["q"].to(dtype).to(DEV)
    kb, vb = g["k_buffer"].to(dtype).to(DEV), g["v_buffer"].to(dtype).to(DEV)
    B, Hq, D = q.shape
    Dv = vb.shape[-1]
    o = torch.zeros(B, Hq, Dv, dtype=dtype, device=DEV)
    al = torch.empty(B, Hq, max_splits, Dv, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, max_splits, dtype=torch.float32, device=DEV)
    ns = (g["num_kv_splits"] if splits is None else splits).to(torch.int32).to(DEV)
    ops.decode_attention_fwd(q, kb, vb, o, g["kv_indptr"].to(DEV), g["kv_indices"].to(DEV), al, lse, ns, max_splits,
                             g["sm_scale"], g["logit_cap"])
    torch.cuda.synchronize()
    return o.float().cpu()


DECODE_CASES = ["gqa128", "mha64", "mqa_cap", "mla", "odd", "shared", "gqa_q05"]


@pytest.mark.parametrize("case", DECODE_CASES)
def test_decode_attention_golden_bf16(golden, case):
    ops = _ops()
    g = golden("decode_" + case)
    o = _run_decode(ops, g, torch.bfloat16)
    ref32 = g["expect_triton_f32"]
    assert torch.isfinite(o).all()
    assert (o - ref32).abs().max().item() < 2e-2
    cpu = g["expect_cpu_bf16"].float()
    cos = torch.nn.functional.cosine_similarity(o.flatten(), cpu.flatten(), dim=0)
    assert cos.item() > 0.99
    torch.testing.assert_close(o, cpu, atol=3e-2, rtol=1e-6)


@pytest.mark.parametrize("case", ["gqa128", "odd", "mha64"])
def test_decode_attention_golden_f32_f16(golden, case):
    ops = _ops()
    g = golden("decode_" + case)
    o32 = _run_decode(ops, g, torch.float32)
    assert (o32 - g["expect_triton_f32"]).abs().max().item() < 2e-5
    o16 = _run_decode(ops, g, torch.float16)
    assert (o16 - g["expect_triton_f32"]).abs().max().item() < 4e-3


def _synthetic_batch(B, Hq, Hkv, D, prefix, suffix_lo, suffix_hi, dtype, seed=0, ragged=True):
    gen = torch.Generator().manual_seed(seed)
    lens = [prefix + int(torch.randint(suffix_lo, suffix_hi + 1, (1,), generator=gen)) for _ in range(B)]
    if not ragged:
        lens = [prefix + suffix_hi] * B
    n = 1 + prefix + sum(L - prefix for L in lens) + 3
    kb = torch.randn(n, Hkv, D, generator=gen).to(dtype)
    vb = torch.randn(n, Hkv, D, generator=gen).to(dtype)
    q = torch.randn(B, Hq, D, generator=gen).to(dtype)
    r2t = torch.zeros(B, max(lens) + 4, dtype=torch.int32)
    nxt = 1 + prefix
    for b, L in enumerate(lens):
        r2t[b, :prefix] = torch.arange(1, 1 + prefix)
        r2t[b, prefix:L] = torch.arange(nxt, nxt + L - prefix)
        nxt += L - prefix
    return q, kb, vb, r2t, torch.tensor(lens, dtype=torch.int64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,Hkv,prefix,max_splits", [(5, 32, 8, 96, 8), (3, 8, 8, 10, 4), (2, 64, 2, 300, 16),
                                                        (4, 40, 8, 0, 8)])
def test_decode_attention_vs_oracle(dtype, B, Hq, Hkv, prefix, max_splits):
    """Seeded shared-prefix batches (head size 128: the matrix-core path) against the C oracle; includes a
    GQA group of 32 (two head tiles), MHA, ragged lengths and sequences shorter than one tile."""
    import oracle

    ops = _ops()
    q, kb, vb, r2t, seq = _synthetic_batch(B, Hq, Hkv, 128, prefix, 1, 200, dtype, seed=B)
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(seq, 0)
    ind = oracle.create_kv_indices(r2t, torch.arange(B), seq, ptr)
    ns_d = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(ns_d, seq.to(DEV), Hq, Hkv, max_splits, 256)
    ns = ns_d.cpu()
    sm = 1.0 / math.sqrt(128)
    want = oracle.decode_attention(q, kb, vb, ptr, ind, ns, sm)
    g = dict(q=q, k_buffer=kb, v_buffer=vb, kv_indptr=ptr, kv_indices=ind, num_kv_splits=ns, sm_scale=sm,
             logit_cap=0.0)
    o = _run_decode(ops, g, dtype, max_splits)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert (o - want).abs().max().item() < tol
    cos = torch.nn.functional.cosine_similarity(o.flatten(), want.flatten(), dim=0)
    assert cos.item() > 0.999


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,Hkv,prefix,max_splits", [(5, 14, 2, 96, 8), (3, 8, 4, 10, 4), (2, 64, 2, 300, 16),
                                                        (4, 16, 16, 0, 8), (3, 40, 8, 33, 8)])
def test_decode_attention_head64_pairs_vs_oracle(dtype, B, Hq, Hkv, prefix, max_splits):
    """Head size 64 with an even number of contiguous kv heads: pairs of kv heads are walked as one 128-element row on
    the matrix-core kernel (query rows zero-padded into their head's half).  Against the C oracle, and against the
    generic kernel, which a pool whose heads are not contiguous selects (Qwen2-0.5B's 14 / 2 heads, MHA, a GQA group of
    32 = four head tiles per pair, ragged lengths)."""
    import oracle

    ops = _ops()
    q, kb, vb, r2t, seq = _synthetic_batch(B, Hq, Hkv, 64, prefix, 1, 200, dtype, seed=B + Hq)
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(seq, 0)
    ind = oracle.create_kv_indices(r2t, torch.arange(B), seq, ptr)
    ns_d = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(ns_d, seq.to(DEV), Hq, Hkv, max_splits, 256)
    ns = ns_d.cpu()
    sm = 1.0 / math.sqrt(64)
    want = oracle.decode_attention(q, kb, vb, ptr, ind, ns, sm)
    g = dict(q=q, k_buffer=kb, v_buffer=vb, kv_indptr=ptr, kv_indices=ind, num_kv_splits=ns, sm_scale=sm,
             logit_cap=0.0)
    o = _run_decode(ops, g, dtype, max_splits)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want).abs().max().item() < tol
    # the same through the generic kernel: pools padded to a head stride of 72 elements are not pairable
    pad = lambda t: torch.nn.functional.pad(t.to(dtype), (0, 8)).to(DEV)[..., :64]
    kd, vd = pad(kb), pad(vb)
    assert kd.stride(1) == 72
    o2 = torch.zeros(B, Hq, 64, dtype=dtype, device=DEV)
    al = torch.empty(B, Hq, max_splits, 64, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, max_splits, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd(q.to(dtype).to(DEV), kd, vd, o2, ptr.to(DEV), ind.to(DEV), al, lse, ns.to(DEV), max_splits,
                             sm, 0.0)
    assert (o - o2.float().cpu()).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,max_len,max_splits,cap", [(3, 16, 70, 4, 0.0), (4, 128, 500, 8, 0.0), (2, 24, 1300, 16, 30.0),
                                                        (5, 16, 31, 2, 0.0), (2, 40, 200, 4, 0.0), (3, 48, 333, 8, 20.0)])
def test_decode_attention_mla(dtype, B, Hq, max_len, max_splits, cap):
    """Absorbed-MLA decode (memory_pool.py:492-521: one latent row of 576 per token, values = a view of its first
    512 elements) on the matrix-core kernel, against the C oracle and against the generic kernel (which a separate
    copy of the values selects)."""
    import oracle

    ops = _ops()
    gen = torch.Generator().manual_seed(B * 1000 + Hq)
    lens = [int(torch.randint(1, max_len + 1, (1,), generator=gen)) for _ in range(B)]
    lens[0] = max_len
    n = sum(lens) + 7
    kv = torch.randn(n, 1, 576, generator=gen).to(dtype)
    q = (torch.randn(B, Hq, 576, generator=gen) * 0.5).to(dtype)
    perm = torch.randperm(n - 1, generator=gen)[:sum(lens)] + 1
    ind = perm.to(torch.int32)
    ptr = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    ns_d = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(ns_d, torch.tensor(lens, device=DEV), Hq, 1, max_splits, 256)
    sm = 1.0 / math.sqrt(192)
    want = oracle.decode_attention(q, kv, kv[..., :512].contiguous(), ptr, ind, ns_d.cpu(), sm, cap)
    kv_d = kv.to(DEV)

    def run(vb):
        o = torch.full((B, Hq, 512), float("nan"), dtype=dtype, device=DEV)
        al = torch.empty(B, Hq, max_splits, 512, dtype=torch.float32, device=DEV)
        lse = torch.empty(B, Hq, max_splits, dtype=torch.float32, device=DEV)
        ops.decode_attention_fwd(q.to(DEV), kv_d, vb, o, ptr.to(DEV), ind.to(DEV), al, lse, ns_d, max_splits, sm, cap)
        torch.cuda.synchronize()
        return o.float().cpu()

    fast = run(kv_d[..., :512])
    generic = run(kv_d[..., :512].contiguous())
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(fast).all()
    assert (fast - want).abs().max().item() < tol
    assert (fast - generic).abs().max().item() < tol


def test_decode_attention_noncontiguous_pool():
    """Pools with permuted head/token strides, as the reference CPU test builds them
    (test/srt/cpu/test_decode.py:103-107)."""
    import oracle

    ops = _ops()
    q, kb, vb, r2t, seq = _synthetic_batch(3, 32, 8, 128, 40, 5, 60, torch.bfloat16, seed=7)
    B = 3
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(seq, 0)
    ind = oracle.create_kv_indices(r2t, torch.arange(B), seq, ptr)
    ns = torch.tensor([2, 1, 3], dtype=torch.int32)
    sm = 1.0 / math.sqrt(128)
    want = oracle.decode_attention(q, kb, vb, ptr, ind, ns, sm)
    kb_d = kb.to(DEV).transpose(0, 1).contiguous().transpose(0, 1)
    vb_d = vb.to(DEV).transpose(0, 1).contiguous().transpose(0, 1)
    assert not kb_d.is_contiguous()
    o = torch.zeros(B, 32, 128, dtype=torch.bfloat16, device=DEV)
    al = torch.empty(B, 32, 4, 128, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, 32, 4, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kb_d, vb_d, o, ptr.to(DEV), ind.to(DEV), al, lse, ns.to(DEV), 4, sm, 0.0)
    assert (o.float().cpu() - want).abs().max().item() < 2e-2


def test_decode_attention_full_size_properties():
    """The benchmark workload's size (batch 64, 1024 shared + 128 private tokens, Llama-3-8B heads).  Too large
    for the scalar oracle in seconds, so: (1) the result must not depend on the split count (8 vs 1 vs 16);
    (2) requests with identical slots and queries must give identical rows; (3) a sample of rows is checked
    against the oracle."""
    import oracle

    ops = _ops()
    B, Hq, Hkv, D = 64, 32, 8, 128
    q, kb, vb, r2t, seq = _synthetic_batch(B, Hq, Hkv, D, 1024, 128, 128, torch.bfloat16, seed=1, ragged=False)
    q[1] = q[0]
    r2t[1] = r2t[0]
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(seq, 0)
    ind = oracle.create_kv_indices(r2t, torch.arange(B), seq, ptr)
    sm = 1.0 / math.sqrt(D)
    g = dict(q=q, k_buffer=kb, v_buffer=vb, kv_indptr=ptr, kv_indices=ind, sm_scale=sm, logit_cap=0.0)
    o8 = _run_decode(ops, g, torch.bfloat16, 8, torch.full((B,), 8))
    o1 = _run_decode(ops, g, torch.bfloat16, 8, torch.full((B,), 1))
    o16 = _run_decode(ops, g, torch.bfloat16, 16, torch.full((B,), 16))
    assert (o8 - o1).abs().max().item() < 1e-2
    assert (o8 - o16).abs().max().item() < 1e-2
    assert torch.equal(o8[0], o8[1])
    sel = [0, 17, 63]
    ptr_s = torch.zeros(len(sel) + 1, dtype=torch.int32)
    ptr_s[1:] = torch.cumsum(seq[sel], 0)
    ind_s = torch.cat([r2t[b, : int(seq[b])] for b in sel])
    want = oracle.decode_attention(q[sel], kb, vb, ptr_s, ind_s, torch.full((3,), 8), sm)
    assert (o8[sel] - want).abs().max().item() < 2e-2


def test_decode_attention_rejects_bad_arguments():
    ops = _ops()
    from sglang_security_amd import SglkError

    q = torch.zeros(1, 6, 128, dtype=torch.bfloat16, device=DEV)
    kb = torch.zeros(4, 4, 128, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(1, 6, 128, dtype=torch.bfloat16, device=DEV)
    al = torch.zeros(1, 6, 2, 128, device=DEV)
    lse = torch.zeros(1, 6, 2, device=DEV)
    ptr = torch.tensor([0, 2], dtype=torch.int32, device=DEV)
    ind = torch.tensor([1, 2], dtype=torch.int32, device=DEV)
    ns = torch.tensor([1], dtype=torch.int32, device=DEV)
    with pytest.raises(SglkError):
        ops.decode_attention_fwd(q, kb, kb, o, ptr, ind, al, lse, ns, 2, 1.0)  # 6 heads over 4 kv heads
    with pytest.raises(SglkError):
        ops.decode_attention_fwd(q.cpu(), kb, kb, o, ptr, ind, al, lse, ns, 2, 1.0)  # CPU tensor: no fallback


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("rotary_dim", [64, 32])
def test_decode_attention_mla_fused_rope(dtype, neox, rotary_dim):
    """decode_attention_fwd_grouped_rope (rocm_mla_decode_rope.py:402-440): rotating q's rope part and the newest
    token's rope part inside the call equals rotating them beforehand (torch restatement of :116-202) and running
    the plain absorbed-MLA decode; the rotated key part comes back in k_pe_tokens and sits in the pool row."""
    import oracle

    ops = _ops()
    B, H, lora, rope = 5, 16, 512, 64
    gen = torch.Generator().manual_seed(11)
    lens = [1, 40, 97, 33, 64]
    n = sum(lens) + 3
    kv = torch.randn(n, 1, lora + rope, generator=gen).to(dtype)
    q = (torch.randn(B, H, lora + rope, generator=gen) * 0.5).to(dtype)
    ind = (torch.randperm(n - 1, generator=gen)[:sum(lens)] + 1).to(torch.int32)
    ptr = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    positions = torch.tensor([l - 1 for l in lens], dtype=torch.int64)
    inv = 1.0 / (10000.0 ** (torch.arange(0, rotary_dim, 2).float() / rotary_dim))
    fr = torch.einsum("i,j->ij", torch.arange(256).float(), inv)
    cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1)

    def rot(x, pos):                                        # x [..., rope] fp32, one position
        c, s = cos_sin[pos, : rotary_dim // 2], cos_sin[pos, rotary_dim // 2:]
        y = x.clone()
        if neox:
            a, b = x[..., : rotary_dim // 2], x[..., rotary_dim // 2: rotary_dim]
            y[..., : rotary_dim // 2], y[..., rotary_dim // 2: rotary_dim] = a * c - b * s, b * c + a * s
        else:
            a, b = x[..., 0:rotary_dim:2], x[..., 1:rotary_dim:2]
            y[..., 0:rotary_dim:2], y[..., 1:rotary_dim:2] = a * c - b * s, b * c + a * s
        return y

    q_ref, kv_ref = q.clone(), kv.clone()
    kpe_ref = torch.zeros(B, rope, dtype=dtype)
    for b in range(B):
        p = int(positions[b])
        q_ref[b, :, lora:] = rot(q[b, :, lora:].float(), p).to(dtype)
        last = int(ind[ptr[b + 1] - 1])
        kpe_ref[b] = rot(kv[last, 0, lora:].float(), p).to(dtype)
        kv_ref[last, 0, lora:] = kpe_ref[b]
    S = 4
    ns = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(ns, torch.tensor(lens, device=DEV), H, 1, S, 256)
    sm = 1.0 / math.sqrt(192)
    want = oracle.decode_attention(q_ref, kv_ref, kv_ref[..., :lora].contiguous(), ptr, ind, ns.cpu(), sm, 0.0)

    kv_d = kv.to(DEV)
    o = torch.full((B, H, lora), float("nan"), dtype=dtype, device=DEV)
    kpe = torch.zeros(B, 1, rope, dtype=dtype, device=DEV)
    logits = torch.empty(B, H, S, lora + 1, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd_grouped_rope(q.to(DEV), kv_d, kv_d[..., :lora], o, ptr.to(DEV), ind.to(DEV), kpe, lora,
                                          rotary_dim, cos_sin.to(DEV), positions.to(DEV), logits, ns, sm, 0.0,
                                          use_rope=True, is_neox_style=neox)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert (o.float().cpu() - want).abs().max().item() < tol
    step = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10           # one unit in the last place at magnitude < 4
    assert (kpe.view(B, rope).float().cpu() - kpe_ref.float()).abs().max().item() <= 4 * step
    assert torch.equal(kv_d.cpu()[ind[ptr[1:] - 1].long(), 0, lora:], kpe.view(B, rope).cpu())   # pool row == output
    untouched = torch.ones(n, dtype=torch.bool)
    untouched[ind[ptr[1:] - 1].long()] = False
    assert torch.equal(kv_d.cpu()[untouched], kv[untouched])
    assert torch.equal(kv_d.cpu()[..., :lora], kv[..., :lora])
    # without rotation the call is the plain decode
    o2 = torch.empty_like(o)
    kv_d2 = kv.to(DEV)
    ops.decode_attention_fwd_grouped_rope(q.to(DEV), kv_d2, kv_d2[..., :lora], o2, ptr.to(DEV), ind.to(DEV), kpe, lora,
                                          rotary_dim, cos_sin.to(DEV), positions.to(DEV), logits, ns, sm, 0.0)
    want2 = oracle.decode_attention(q, kv, kv[..., :lora].contiguous(), ptr, ind, ns.cpu(), sm, 0.0)
    assert (o2.float().cpu() - want2).abs().max().item() < tol
    assert torch.equal(kv_d2.cpu(), kv)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("B,H", [(5, 16), (64, 128), (1, 3)])
def test_mla_rope_and_store(dtype, neox, B, H):
    """The one-launch rotary step + cache write of the absorbed-MLA decode (models/deepseek_v2.py:1127-1193) writes
    the bytes of the sequence it replaces: rope on copies of q_pe / k_pe (the rope kernel the golden rotary vectors
    pin), q_input[..., lora:] = q_pe, the latent row [k_nope | k_pe] stored at the requests' slots."""
    ops = _ops()
    lora, nope, rope = 512, 128, 64
    gen = torch.Generator().manual_seed(B * 7 + H)
    q = (torch.randn(B, H, nope + rope, generator=gen) * 0.5).to(dtype).to(DEV)       # q_b's output: [nope | rope]
    qkv_a = torch.randn(B, 96 + lora + rope, generator=gen).to(dtype).to(DEV)        # ... | k_nope | k_pe
    slots = 3 * B + 2
    pool = torch.randn(slots, 1, lora + rope, generator=gen).to(dtype).to(DEV)
    loc = (torch.randperm(slots - 1, generator=gen)[:B] + 1).to(DEV)
    positions = torch.randint(0, 200, (B,), generator=gen).to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, rope, 2).float() / rope))
    fr = torch.einsum("i,j->ij", torch.arange(256).float(), inv)
    cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).to(DEV)
    q_input = torch.randn(B, H, lora + rope, generator=gen).to(dtype).to(DEV)

    q_pe = q[..., nope:].clone().reshape(B, H * rope)
    k_pe = qkv_a[:, 96 + lora:].clone()
    ops.rope(q_pe, k_pe, cos_sin, positions, rope, interleave=not neox)
    want_q = q_input.clone()
    want_q[..., lora:] = q_pe.view(B, H, rope)
    want_pool = pool.clone()
    want_pool[loc, 0, :lora] = qkv_a[:, 96: 96 + lora]
    want_pool[loc, 0, lora:] = k_pe

    for loc_t in (loc, loc.to(torch.int32)):
        got_q, got_pool = q_input.clone(), pool.clone()
        ops.mla_rope_and_store(q[..., nope:], got_q, qkv_a[:, 96:], got_pool, loc_t, cos_sin, positions, lora, is_neox=neox)
        assert torch.equal(got_q, want_q)
        assert torch.equal(got_pool, want_pool)
    from sglang_security_amd import SglkError

    with pytest.raises(SglkError):
        ops.mla_rope_and_store(q[..., nope:], q_input[:, :, : lora + 8], qkv_a[:, 96:], pool, loc, cos_sin, positions, lora)
