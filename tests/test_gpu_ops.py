"""GPU parity for the non-attention ops of the decode path, through the C ABI: AWQ dequant (bit-exact) and
fused AWQ GEMM, dense skinny GEMM, RMSNorm / fused add, gated activations, RoPE, KV-pool writes, merge_state.

Tolerances are the reference's (sgl-kernel/tests): awq_dequantize rtol 1e-3 atol 1e-5 (we require bit
equality), norm / activation rtol=atol=1e-3 (fp16), rope 1e-2, merge_state atol 1e-3 rtol 1e-2.

GEMMs: per element atol = rtol = 1e-2 for bf16 (test/srt/cpu/utils.py:6-10), 2e-3 for fp16, on outputs of order one
(activations are scaled so that the products have a standard deviation near 0.5, the magnitude the reference's bar is
written for).  The fused kernel multiplies by (q - z) * scale without first rounding the dequantised weight to 16 bits,
as the reference's dequantise-then-matmul does; that rounding (relative 2^-9 per bf16 weight) is what the tolerance
absorbs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from sglang_security_amd import ops

    return ops


# ---------------------------------------------------------------------------------------------- AWQ
@pytest.mark.parametrize("name", ["awq_g128_bf16", "awq_g128_f16", "awq_gall_f16", "awq_g64_bf16"])
def test_awq_dequantize_golden_bitexact(golden, name):
    ops = _ops()
    g = golden(name)
    out = ops.awq_dequantize(g["qweight"].to(DEV), g["scales"].to(DEV), g["qzeros"].to(DEV)).cpu()
    assert torch.equal(out.view(torch.int16), g["expect"].view(torch.int16))


@pytest.mark.parametrize("rows,qcols", [(3584, 448), (128, 16), (1024, 576), (18944, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_awq_dequantize_reference_shapes(rows, qcols, dtype):
    """Shapes of sgl-kernel/tests/test_awq_dequant.py:66-75 with group_size = rows, against the C oracle."""
    import oracle

    ops = _ops()
    torch.manual_seed(rows + qcols)
    qw = torch.randint(0, torch.iinfo(torch.int32).max, (rows, qcols), dtype=torch.int32)
    sc = torch.rand(1, qcols * 8).to(dtype)
    qz = torch.randint(0, torch.iinfo(torch.int32).max, (1, qcols), dtype=torch.int32)
    want = oracle.awq_dequantize(qw, sc, qz)
    out = ops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV)).cpu()
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


def _awq_case(M, K, N, group, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g)
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // group, N // 8), dtype=torch.int32, generator=g)
    sc = (torch.rand(K // group, N, generator=g) * 0.01 + 0.001).to(dtype)
    # (q - z) * scale has a standard deviation near 0.04: outputs of standard deviation ~0.5
    x = (torch.randn(M, K, generator=g) * (12.0 / math.sqrt(K))).to(dtype)
    return x, qw, sc, qz


def _gemm_close(got, want, dtype):
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    torch.testing.assert_close(got.float(), want.float(), atol=tol, rtol=tol)


@pytest.mark.parametrize("M,K,N,group", [(64, 512, 128, 128), (8, 1024, 256, 128), (1, 256, 64, 128),
                                         (37, 4096, 512, 128), (64, 256, 192, 256), (100, 512, 128, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_awq_gemm_vs_oracle(M, K, N, group, dtype):
    """Fused dequant GEMM vs oracle (dequantise to half, fp32-accumulate matmul): split-K, ragged M, M > 64."""
    import oracle

    ops = _ops()
    x, qw, sc, qz = _awq_case(M, K, N, group, dtype, seed=M + K)
    want = oracle.awq_gemm(x, qw, sc, qz).float()
    packed = ops.awq_repack(qw.to(DEV))
    out = ops.awq_gemm(x.to(DEV), packed, sc.to(DEV), qz.to(DEV)).float().cpu()
    _gemm_close(out, want, dtype)


def test_awq_gemm_golden(golden):
    ops = _ops()
    for name in ("awq_g128_bf16", "awq_g128_f16"):
        g = golden(name)
        packed = ops.awq_repack(g["qweight"].to(DEV))
        out = ops.awq_gemm(g["x"].to(DEV), packed, g["scales"].to(DEV), g["qzeros"].to(DEV)).float().cpu()
        want = g["expect_matmul_f32"]
        # the fixture's outputs have a standard deviation near 60: the reference's bar after normalising to unit RMS
        rms = want.pow(2).mean().sqrt().item()
        torch.testing.assert_close(out / rms, want / rms, atol=1e-2, rtol=1e-2)


def test_awq_gemm_exact_on_integers():
    """Small-integer activations, unit scales: every product and sum is exact in fp32, so the fused kernel must
    reproduce x @ (q - z) exactly -- checks nibble order, repack layout and the zero-point correction."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    M, K, N = 64, 512, 256
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g)
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, generator=g)
    sc = torch.ones(K // 128, N, dtype=torch.bfloat16)
    x = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16)
    w = ops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV)).float().cpu()
    want = x.float() @ w
    assert want.abs().max() < 256 * 8  # representable in bf16 only approximately; compare after rounding
    packed = ops.awq_repack(qw.to(DEV))
    out = ops.awq_gemm(x.to(DEV), packed, sc.to(DEV), qz.to(DEV)).cpu()
    assert torch.equal(out, want.to(torch.bfloat16))


@pytest.mark.parametrize("M,K,N", [(64, 4096, 6144), (64, 4096, 4096), (64, 14336, 4096), (64, 4096, 28672)])
def test_awq_gemm_llama8b_shapes_linearity(M, K, N):
    """Llama-3-8B projection shapes (the benchmark's model).  Size-independent properties: (1) linearity in x on
    exactly representable inputs, (2) a sampled set of columns against the oracle."""
    import oracle

    ops = _ops()
    x, qw, sc, qz = _awq_case(M, K, N, 128, torch.bfloat16, seed=N)
    packed = ops.awq_repack(qw.to(DEV))
    scd, qzd = sc.to(DEV), qz.to(DEV)
    y1 = ops.awq_gemm(x.to(DEV), packed, scd, qzd).float()
    y2 = ops.awq_gemm((2 * x).to(DEV), packed, scd, qzd).float()
    assert torch.equal(y2, 2 * y1)  # doubling is exact in every product, sum and rounding
    cols = torch.arange(0, N, max(N // 64, 8))[:64]
    blocks = torch.unique(cols // 8)
    # oracle on the int32 columns that hold the sampled outputs
    sub_cols = (blocks[:, None] * 8 + torch.arange(8)[None, :]).flatten()
    want = oracle.awq_gemm(x, qw[:, blocks].contiguous(), sc[:, sub_cols].contiguous(), qz[:, blocks].contiguous())
    got = y1.cpu()[:, sub_cols]
    _gemm_close(got, want, torch.bfloat16)


@pytest.mark.parametrize("M,K,N", [(64, 512, 256), (3, 1024, 100), (64, 4096, 1000)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("strips", [1, 2])
def test_dense_gemm_nt(M, K, N, dtype, strips, monkeypatch):
    import oracle

    monkeypatch.setenv("SGLK_DENSE_NS", str(strips))  # both kernel shapes (two strips per wave is the lm_head one)
    ops = _ops()
    g = torch.Generator().manual_seed(N)
    x = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    want = oracle.half_matmul(x, w.t().contiguous()).float()
    out = ops.dense_gemm_nt(x.to(DEV), w.to(DEV)).float().cpu()
    _gemm_close(out, want, dtype)


@pytest.mark.parametrize("M,K,N,bias", [(64, 512, 256, False), (3, 1024, 96, True), (64, 4096, 16384, True), (64, 4096, 32064, False)])
def test_dense_gemm_packed(M, K, N, bias):
    """The sequential-streaming weight layout: same product, bit for bit, as the row-major kernel of the same shape
    (same accumulation order), and the oracle's on sampled rows."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16) if bias else None
    wd, bd = w.to(DEV), (b.to(DEV) if bias else None)
    packed = ops.dense_repack(wd)
    assert packed.shape == wd.shape and not torch.equal(packed, wd)
    got = ops.dense_gemm_packed(x.to(DEV), packed, bd)
    import os
    os.environ["SGLK_DENSE_NS"] = "2" if N >= 16 * 1024 else "1"
    try:
        ref = ops.dense_gemm_nt(x.to(DEV), wd, bd)
    finally:
        os.environ.pop("SGLK_DENSE_NS", None)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    rows = torch.arange(0, M, max(M // 4, 1))
    want = oracle.half_matmul(x[rows], w.t().contiguous()).float()
    if bias:
        want = (want + b.float()).to(torch.bfloat16).float()
    _gemm_close(got.float().cpu()[rows], want, torch.bfloat16)


@pytest.mark.parametrize("M,K,N,bias", [(64, 4096, 4096, False), (64, 14336, 4096, True), (7, 4096, 6144, False),
                                        (130, 1024, 512, True), (64, 512, 28672, False)])
def test_dense_gemm_nt_split_k(M, K, N, bias):
    """Two-launch split-K form (workspace given) against the single-launch kernel and the oracle."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16) if bias else None
    ws = torch.empty(max(ops.dense_gemm_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=DEV)
    bd = b.to(DEV) if bias else None
    got = ops.dense_gemm_nt(x.to(DEV), w.to(DEV), bd, None, ws).float().cpu()
    one = ops.dense_gemm_nt(x.to(DEV), w.to(DEV), bd).float().cpu()
    scale = one.abs().max().item()
    assert (got - one).abs().max().item() <= 8e-3 * scale          # another summation order, one bf16 rounding
    rows = torch.arange(0, M, max(M // 8, 1))
    want = oracle.half_matmul(x[rows], w.t().contiguous()).float()
    if bias:
        want = (want + b.float()).to(torch.bfloat16).float()
    _gemm_close(got[rows], want, torch.bfloat16)


# ---------------------------------------------------------------------------------------------- norm / act / rope
@pytest.mark.parametrize("name", ["norm_f16", "norm_bf16", "norm_odd"])
def test_rmsnorm_golden(golden, name):
    ops = _ops()
    g = golden(name)
    x, r, w = g["x"].to(DEV), g["residual"].to(DEV), g["weight"].to(DEV)
    out = torch.empty_like(x)
    ops.rmsnorm(out, x, w, g["eps"])
    torch.testing.assert_close(out.cpu().float(), g["expect_rmsnorm"].float(), rtol=1e-3, atol=1e-3)
    x2, r2 = x.clone(), r.clone()
    ops.fused_add_rmsnorm(x2, r2, w, g["eps"])
    torch.testing.assert_close(x2.cpu().float(), g["expect_fused_x"].float(), rtol=1e-3, atol=1e-3)
    # the residual is a single rounded sum: bit-exact
    assert torch.equal(r2.cpu().view(torch.int16), g["expect_fused_residual"].view(torch.int16))


@pytest.mark.parametrize("rows,hidden", [(1, 111), (19, 4096), (64, 8192), (7, 16384)])
def test_rmsnorm_vs_oracle(rows, hidden):
    import oracle

    ops = _ops()
    torch.manual_seed(hidden)
    x = torch.randn(rows, hidden).to(torch.float16)
    r = torch.randn(rows, hidden).to(torch.float16)
    w = torch.randn(hidden).to(torch.float16)
    out = torch.empty(rows, hidden, dtype=torch.float16, device=DEV)
    ops.rmsnorm(out, x.to(DEV), w.to(DEV), 1e-6)
    torch.testing.assert_close(out.cpu().float(), oracle.rmsnorm(x, w, 1e-6).float(), rtol=1e-3, atol=1e-3)
    xd, rd = x.to(DEV), r.to(DEV)
    ops.fused_add_rmsnorm(xd, rd, w.to(DEV), 1e-6)
    wx, wr = oracle.fused_add_rmsnorm(x, r, w, 1e-6)
    torch.testing.assert_close(xd.cpu().float(), wx.float(), rtol=1e-3, atol=1e-3)
    assert torch.equal(rd.cpu().view(torch.int16), wr.view(torch.int16))


@pytest.mark.parametrize("name", ["silu_f16", "silu_bf16"])
def test_silu_and_mul_golden(golden, name):
    ops = _ops()
    g = golden(name)
    x = g["x"].to(DEV)
    out = torch.empty(x.shape[0], x.shape[1] // 2, dtype=x.dtype, device=DEV)
    ops.act_and_mul(out, x, 0)
    torch.testing.assert_close(out.cpu().float(), g["expect"].float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("name", ["rope_neox128", "rope_gptj128", "rope_partial"])
def test_rope_golden(golden, name):
    ops = _ops()
    g = golden(name)
    q, k = g["q"].to(DEV), g["k"].to(DEV)
    ops.rope(q, k, g["cos_sin_cache"].to(DEV), g["positions"].to(DEV), g["head_size"], interleave=not g["is_neox"])
    torch.testing.assert_close(q.cpu().float(), g["expect_q"].float(), atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(k.cpu().float(), g["expect_k"].float(), atol=1e-2, rtol=1e-2)


def test_rope_and_store_kv_matches_separate_ops(golden):
    """Fused rope + pool write == rope then set_kv_buffer, bit for bit."""
    ops = _ops()
    g = golden("rope_neox128")
    T = g["q"].shape[0]
    hq, hk, hs = 32, 8, 128
    torch.manual_seed(0)
    v = torch.randn(T, hk * hs).to(torch.bfloat16)
    qkv = torch.cat([g["q"], g["k"], v], dim=1).contiguous().to(DEV)
    cs, pos = g["cos_sin_cache"].to(DEV), g["positions"].to(DEV)
    kbuf = torch.zeros(50, hk, hs, dtype=torch.bfloat16, device=DEV)
    vbuf = torch.zeros(50, hk, hs, dtype=torch.bfloat16, device=DEV)
    loc = torch.randperm(49)[:T].to(torch.int64).to(DEV) + 1
    ops.rope_and_store_kv(qkv, cs, pos, kbuf, vbuf, loc, hq, hk, hs)
    q2, k2 = g["q"].to(DEV), g["k"].to(DEV)
    ops.rope(q2, k2, cs, pos, hs, interleave=False)
    kb2, vb2 = torch.zeros_like(kbuf), torch.zeros_like(vbuf)
    ops.set_kv_buffer(kb2, vb2, loc, k2.view(T, hk, hs), v.to(DEV).view(T, hk, hs))
    assert torch.equal(qkv[:, : hq * hs], q2)
    assert torch.equal(qkv[:, hq * hs : (hq + hk) * hs], k2)
    assert torch.equal(kbuf, kb2) and torch.equal(vbuf, vb2)
    assert torch.equal(kbuf[loc].view(T, -1), k2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_set_kv_buffer(dtype):
    ops = _ops()
    torch.manual_seed(1)
    kb = torch.zeros(100, 8, 128, dtype=dtype, device=DEV)
    vb = torch.zeros(100, 8, 96, dtype=dtype, device=DEV)
    k = torch.randn(5, 8, 128).to(dtype).to(DEV)
    v = torch.randn(5, 8, 96).to(dtype).to(DEV)
    for idt in (torch.int64, torch.int32):
        loc = torch.tensor([7, 0, 99, 3, 50], dtype=idt, device=DEV)
        ops.set_kv_buffer(kb, vb, loc, k, v)
        assert torch.equal(kb[loc.long()], k) and torch.equal(vb[loc.long()], v)
    assert kb[1].abs().sum() == 0


@pytest.mark.parametrize("name,rtol", [("merge_state_bf16", 1e-2), ("merge_state_f16", 1e-3), ("merge_state_f32", 1e-3)])
def test_merge_state_golden(golden, name, rtol):
    ops = _ops()
    g = golden(name)
    va, vb, sa, sb = (g[k].to(DEV) for k in ("v_a", "v_b", "s_a", "s_b"))
    vo, so = torch.empty_like(va), torch.empty_like(sa)
    ops.merge_state(va, sa, vb, sb, vo, so)
    torch.testing.assert_close(vo.cpu().float(), g["expect_v"].float(), atol=1e-3, rtol=rtol)
    torch.testing.assert_close(so.cpu(), g["expect_s"], atol=1e-3, rtol=1e-3)


# ------------------------------------------------------------------------------------- AWQ, wide kernel
@pytest.fixture
def force_wide(monkeypatch):
    """SGLK_AWQ_WIDE=2 routes every eligible shape (K % 512 == 0, N % 128 == 0, group 128) through the 8-wave kernel,
    whatever the planner would pick."""
    monkeypatch.setenv("SGLK_AWQ_WIDE", "2")


@pytest.mark.parametrize("M,K,N", [(64, 512, 128), (64, 2048, 256), (7, 4096, 384), (33, 1536, 128), (64, 7168, 512)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_awq_gemm_wide_vs_oracle(force_wide, M, K, N, dtype):
    """One stage, several stages, split-K with and without a ragged last split, ragged M."""
    import oracle

    ops = _ops()
    x, qw, sc, qz = _awq_case(M, K, N, 128, dtype, seed=M + K + N)
    want = oracle.awq_gemm(x, qw, sc, qz).float()
    packed = ops.awq_repack(qw.to(DEV))
    got = ops.awq_gemm(x.to(DEV), packed, sc.to(DEV), qz.to(DEV)).float().cpu()
    _gemm_close(got, want, dtype)
    # twice in a row on the same workspace: the ticket words must have been left at zero
    work = torch.zeros(max(ops.awq_gemm_workspace_bytes(M, N, K), 256), dtype=torch.uint8, device=DEV)
    a = ops.awq_gemm(x.to(DEV), packed, sc.to(DEV), qz.to(DEV), None, None, work)
    b = ops.awq_gemm(x.to(DEV), packed, sc.to(DEV), qz.to(DEV), None, None, work)
    assert torch.equal(a, b) and torch.equal(a.float().cpu(), got)


def test_awq_gemm_wide_zero_scales(force_wide):
    """Columns whose scale is exactly zero in some groups (first, middle, last, all): they contribute nothing
    there and the rest of the sum is unaffected."""
    import oracle

    ops = _ops()
    M, K, N = 64, 2048, 256
    x, qw, sc, qz = _awq_case(M, K, N, 128, torch.bfloat16, seed=5)
    sc = sc.clone()
    sc[0, 3] = 0          # first group
    sc[7, 17] = 0         # a middle group
    sc[15, 100] = 0       # last group
    sc[:, 200] = 0        # every group: the whole column is zero
    sc[4:9, 64:80] = 0    # a whole strip, consecutive groups
    want = oracle.awq_gemm(x, qw, sc, qz).float()
    got = ops.awq_gemm(x.to(DEV), ops.awq_repack(qw.to(DEV)), sc.to(DEV), qz.to(DEV)).float().cpu()
    assert torch.isfinite(got).all()
    assert (got[:, 200] == 0).all()
    _gemm_close(got, want, torch.bfloat16)


def test_awq_gemm_full_size_shapes():
    """The workload's own projections (Llama-3-8B, batch 64) against the oracle: the planner's choice of kernel."""
    import oracle

    ops = _ops()
    for (K, N) in [(4096, 28672), (14336, 4096)]:
        x, qw, sc, qz = _awq_case(64, K, N, 128, torch.bfloat16, seed=K)
        want = oracle.awq_gemm(x, qw, sc, qz).float()
        got = ops.awq_gemm(x.to(DEV), ops.awq_repack(qw.to(DEV)), sc.to(DEV), qz.to(DEV)).float().cpu()
        _gemm_close(got, want, torch.bfloat16)



@pytest.mark.parametrize("K,N,group", [(128, 64, 128), (512, 256, 128), (4096, 6144, 128), (256, 48, 128), (256, 128, 64),
                                       (1024, 192, 256)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_awq_dequantize_packed_matches_reference_layout(K, N, group, dtype):
    """Dequantising the re-laid-out copy gives the bits of awq_dequantize on the reference layout (both the
    whole-tile kernel, N % 64 == 0 and group % 128 == 0, and the word-per-thread one)."""
    from sglang_security_amd import ops

    g = torch.Generator().manual_seed(K + N)
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g).to("cuda")
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // group, N // 8), dtype=torch.int32, generator=g).to("cuda")
    sc = (torch.rand(K // group, N, generator=g) * 0.02).to(dtype).to("cuda")
    want = ops.awq_dequantize(qw, sc, qz)
    got = ops.awq_dequantize_packed(ops.awq_repack(qw), sc, qz)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


# ---------------------------------------------------------------------------------------------- large-M (prefill) GEMMs
@pytest.mark.parametrize("M,K,N", [(256, 256, 128), (300, 512, 256), (1152, 1024, 384), (77, 4096, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_awq_gemm_prefill(M, K, N, dtype):
    """Dequant-fused large-M GEMM: the weights it multiplies by are bit for bit awq_dequantize's, so the product has to
    agree with the fp32-accumulated matmul on that matrix to accumulation-order noise (ragged last row tile included)."""
    import oracle

    ops = _ops()
    x, qw, sc, qz = _awq_case(M, K, N, 128, dtype, seed=M + K + N)
    packed = ops.awq_repack(qw.to(DEV))
    got = ops.awq_gemm_prefill(x.to(DEV), packed, sc.to(DEV), qz.to(DEV)).float().cpu()
    want = oracle.awq_gemm(x, qw, sc, qz).float()
    tol = 4e-3 if dtype == torch.bfloat16 else 5e-4        # one rounding of the output, no weight-rounding difference
    torch.testing.assert_close(got, want, atol=tol, rtol=2 * tol)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(1)).to(dtype)
    gb = ops.awq_gemm_prefill(x.to(DEV), packed, sc.to(DEV), qz.to(DEV), bias.to(DEV)).float().cpu()
    torch.testing.assert_close(gb, (want + bias.float()).to(dtype).float(), atol=2 * tol, rtol=2 * tol)


@pytest.mark.parametrize("M,K,N", [(256, 128, 128), (513, 1024, 256), (1152, 4096, 512)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dense_gemm_prefill(M, K, N, dtype):
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * (8.0 / math.sqrt(K))).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.1).to(dtype)
    got = ops.dense_gemm_prefill(x.to(DEV), w.to(DEV)).float().cpu()
    rows = torch.arange(0, M, max(M // 16, 1))
    want = oracle.half_matmul(x[rows], w.t().contiguous()).float()
    _gemm_close(got[rows], want, dtype)
    # against torch's fp32 product everywhere (checks every tile's placement)
    full = (x.float().to(DEV) @ w.float().to(DEV).t()).cpu()
    assert (got - full).abs().max().item() <= 1e-2 * max(full.abs().max().item(), 1.0)
