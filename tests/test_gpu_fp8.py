"""GPU parity for the fp8 (e4m3fn) path through the C ABI: activation quantisation (bit-exact codes and scales
against the C oracle) and the two scaled matmuls (fp32-accumulated; tolerance rtol 2e-2 as in
sgl-kernel/tests/test_fp8_gemm.py:45 and test_fp8_blockwise_gemm.py:86 (rtol 0.02 on the relative mean error))."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
F8 = torch.float8_e4m3fn


def _ops():
    from sglang_security_amd import ops

    return ops


def _u8(t):
    return t.cpu().view(torch.uint8)


def test_hw_cast_matches_oracle_codec():
    """Every e4m3 rounding boundary: the device conversion (scale 1 row: absmax == 448) equals the oracle codec."""
    import oracle

    ops = _ops()
    x = torch.cat([torch.linspace(-448, 448, 8192 * 4 - 2), torch.tensor([448.0, -448.0])]).reshape(1, -1)
    x = torch.cat([x, x * 0.001, torch.randn(1, x.shape[1]).clamp(-3.9, 3.9) * 100])
    x[1, 0], x[2, 0] = 448.0, 448.0  # scale == 1 in every row
    q = torch.empty(x.shape, dtype=F8, device=DEV)
    s = torch.empty(x.shape[0], 1, dtype=torch.float32, device=DEV)
    ops.per_token_quant_fp8(x.to(DEV), q, s)
    assert torch.equal(s.cpu().reshape(-1), torch.ones(3))
    assert torch.equal(_u8(q), oracle.e4m3_encode(x).view(torch.uint8))


@pytest.mark.parametrize("rows,hidden", [(1, 128), (7, 4096), (64, 14336), (33, 1368), (5, 100), (128, 8192)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_per_token_quant_bitexact(rows, hidden, dtype):
    import oracle

    ops = _ops()
    torch.manual_seed(rows * 31 + hidden)
    x = (torch.randn(rows, hidden) * 3).to(dtype)
    wq, ws = oracle.per_token_quant_fp8(x)
    q = torch.empty(rows, hidden, dtype=F8, device=DEV)
    s = torch.empty(rows, 1, dtype=torch.float32, device=DEV)
    ops.per_token_quant_fp8(x.to(DEV), q, s)
    assert torch.equal(s.cpu(), ws)
    assert torch.equal(_u8(q), wq.view(torch.uint8))


def test_per_token_quant_golden(golden):
    """Fixture minted from the reference formula (sgl-kernel/tests/test_per_token_quant_fp8.py:15-27)."""
    ops = _ops()
    g = golden("fp8_per_token_quant")
    x = g["x"]
    q = torch.empty(x.shape, dtype=F8, device=DEV)
    s = torch.empty(x.shape[0], 1, dtype=torch.float32, device=DEV)
    ops.per_token_quant_fp8(x.to(DEV), q, s)
    torch.testing.assert_close(s.cpu().reshape(-1), g["expect_scale"], rtol=1e-6, atol=0)
    torch.testing.assert_close(q.cpu().float(), g["expect_q"].float(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("rows,hidden,group", [(1, 128, 128), (16, 7168, 128), (33, 4096, 64), (5, 512, 32),
                                                (3, 96, 16), (64, 2048, 256), (7, 1536, 512)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("col_major", [False, True])
def test_per_token_group_quant_bitexact(rows, hidden, group, dtype, col_major):
    import oracle

    ops = _ops()
    torch.manual_seed(rows + hidden + group)
    x = (torch.randn(rows, hidden) * 2).to(dtype)
    x[0, :group] = 0  # an all-zero group exercises eps
    wq, ws = oracle.per_token_group_quant_fp8(x, group, 1e-10, -448.0, 448.0)
    q = torch.empty(rows, hidden, dtype=F8, device=DEV)
    ng = hidden // group
    if col_major:
        s = torch.empty(ng, rows, dtype=torch.float32, device=DEV).t()
    else:
        s = torch.empty(rows, ng, dtype=torch.float32, device=DEV)
    ops.per_token_group_quant_fp8(x.to(DEV), q, s, group, 1e-10, -448.0, 448.0)
    assert torch.equal(s.cpu(), ws)
    assert torch.equal(_u8(q), wq.view(torch.uint8))


@pytest.mark.parametrize("n", [1, 1000, 4096, 256 * 1024 + 3, 1024 * 1024])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_per_tensor_quant(n, dtype):
    """sgl-kernel/tests/test_per_tensor_quant_fp8.py: dynamic scale = absmax / 448; static uses the given scale."""
    import oracle

    ops = _ops()
    torch.manual_seed(n)
    x = (torch.randn(n) * 5).to(dtype)
    scale = x.float().abs().max() / 448.0
    want = oracle.e4m3_encode((x.float() * (1.0 / scale)).clamp(-448, 448)).view(torch.uint8)
    q = torch.empty(n, dtype=F8, device=DEV)
    s = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.per_tensor_quant_fp8(x.to(DEV), q, s, False)
    assert s.cpu().item() == scale.item()
    assert torch.equal(_u8(q), want)
    q2 = torch.empty(n, dtype=F8, device=DEV)
    s2 = torch.tensor([0.0625], device=DEV)
    ops.per_tensor_quant_fp8(x.to(DEV), q2, s2, True)
    assert torch.equal(_u8(q2), oracle.e4m3_encode((x.float() * 16.0).clamp(-448, 448)).view(torch.uint8))


@pytest.mark.parametrize("shape,cols,off", [((64, 128, 192), 128, 0), ((5, 3, 40), 24, 8), ((7, 2, 33), 20, 13),
                                            ((4, 16, 512), 512, 0)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_per_tensor_quant_rows(shape, cols, off, dtype):
    """The strided form (a slice of the last dimension, rows one pitch apart; vector and scalar paths) gives the bytes
    and the scale of per_tensor_quant_fp8 on a contiguous copy of the same view."""
    ops = _ops()
    torch.manual_seed(shape[0] + cols)
    x = (torch.randn(shape) * 3).to(dtype).to(DEV)
    view = x[..., off: off + cols]
    want_q = torch.empty(view.shape, dtype=F8, device=DEV)
    want_s = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.per_tensor_quant_fp8(view.contiguous(), want_q, want_s, False)
    q = torch.empty(view.shape, dtype=F8, device=DEV)
    s = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.per_tensor_quant_fp8_rows(view, q, s, False)
    assert s.item() == want_s.item() and torch.equal(_u8(q), _u8(want_q))
    # a padded output pitch, and the static form
    wide = torch.zeros(shape[0], shape[1], cols + 16, dtype=torch.uint8, device=DEV).view(F8)
    ops.per_tensor_quant_fp8_rows(view, wide[..., :cols], s, True)
    assert torch.equal(_u8(wide[..., :cols]), _u8(want_q)) and int(_u8(wide[..., cols:]).max()) == 0
    with pytest.raises(Exception):
        ops.per_tensor_quant_fp8_rows(x.transpose(0, 1)[..., :cols], q.transpose(0, 1), s, False)


def _fp8_rand(shape, g):
    return (torch.randn(shape, generator=g) * 0.5).clamp(-448, 448).to(F8)


@pytest.mark.parametrize("name", ["fp8_mm_bf16", "fp8_mm_f16"])
def test_fp8_scaled_mm_golden(golden, name):
    ops = _ops()
    g = golden(name)
    bias = g.get("bias")
    out = ops.fp8_scaled_mm(g["a"].to(DEV), g["b_nk"].to(DEV).t(), g["scale_a"].to(DEV), g["scale_b"].to(DEV),
                            g["expect"].dtype, None if bias is None else bias.to(DEV))
    torch.testing.assert_close(out.cpu().float(), g["expect"].float(), rtol=0.02, atol=1)


@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (16, 2048, 4096), (17, 512, 1040), (33, 128, 512), (100, 4096, 1024),
                                   (512, 1024, 2048), (3, 16, 16)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_bias", [False, True])
def test_fp8_scaled_mm_vs_oracle(M, N, K, out_dtype, with_bias):
    """Shapes after sgl-kernel/tests/test_fp8_gemm.py:49-56 (M 1..8192, N, K up to 16384), incl. K not a multiple
    of 128 and M not a multiple of 16."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a, b_nk = _fp8_rand((M, K), g), _fp8_rand((N, K), g)
    sa, sb = torch.rand(M, 1, generator=g) * 0.1, torch.rand(N, 1, generator=g) * 0.1
    bias = torch.randn(N, generator=g).to(out_dtype) if with_bias else None
    want = oracle.fp8_scaled_mm(a, b_nk.t(), sa, sb, out_dtype, bias)
    out = ops.fp8_scaled_mm(a.to(DEV), b_nk.to(DEV).t(), sa.to(DEV), sb.to(DEV), out_dtype,
                            None if bias is None else bias.to(DEV))
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M,N,K", [(64, 1280, 8192), (5, 128, 4096), (64, 512, 1040), (100, 256, 2048), (64, 8192, 1024)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_fp8_scaled_mm_split_k(M, N, K, with_bias):
    """Two-launch split-K form (workspace given): narrow outputs, a ragged last k-block, a shape that is not split."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a, b_nk = _fp8_rand((M, K), g), _fp8_rand((N, K), g)
    sa, sb = torch.rand(M, 1, generator=g) * 0.1, torch.rand(N, 1, generator=g) * 0.1
    bias = torch.randn(N, generator=g).to(torch.bfloat16) if with_bias else None
    ws = torch.empty(max(ops.fp8_mm_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=DEV)
    args = (a.to(DEV), b_nk.to(DEV).t(), sa.to(DEV), sb.to(DEV), torch.bfloat16, None if bias is None else bias.to(DEV))
    out = ops.fp8_scaled_mm(*args, None, ws)
    one = ops.fp8_scaled_mm(*args)
    want = oracle.fp8_scaled_mm(a, b_nk.t(), sa, sb, torch.bfloat16, bias)
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(out.cpu().float(), one.cpu().float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K", [(64, 1280, 8192), (64, 8192, 1024), (64, 7168, 8192), (64, 8192, 3584), (64, 6144, 4096),
                                   (5, 128, 4096), (100, 256, 2048), (1, 16, 128), (64, 28672, 4096)])
@pytest.mark.parametrize("with_bias,with_ws", [(False, True), (True, True), (True, False)])
def test_fp8_scaled_mm_packed(M, N, K, with_bias, with_ws):
    """The streaming-layout product (sglk_fp8_repack + sglk_fp8_scaled_mm_packed: split along K through the workspace or
    unsplit without one) on the decode shapes of the 70B TP = 8 shard and of the 8B model, more than 64 rows, one row:
    the oracle's product at the reference test's tolerance, and the row-major kernel's result."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a, b_nk = _fp8_rand((M, K), g), _fp8_rand((N, K), g)
    sa, sb = torch.rand(M, 1, generator=g) * 0.1, torch.rand(N, 1, generator=g) * 0.1
    bias = torch.randn(N, generator=g).to(torch.bfloat16) if with_bias else None
    packed = ops.fp8_repack(b_nk.to(DEV))
    assert packed.shape == b_nk.shape
    ws = torch.empty(max(ops.fp8_mm_packed_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=DEV) if with_ws else None
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.fp8_scaled_mm_packed(a.to(DEV), packed, sa.to(DEV), sb.to(DEV), torch.bfloat16, None if bias is None else bias.to(DEV),
                             out, ws)
    ref = ops.fp8_scaled_mm(a.to(DEV), b_nk.to(DEV).t(), sa.to(DEV), sb.to(DEV), torch.bfloat16,
                            None if bias is None else bias.to(DEV))
    want = oracle.fp8_scaled_mm(a, b_nk.t(), sa, sb, torch.bfloat16, bias)
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(out.cpu().float(), ref.cpu().float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,hidden", [(64, 8192), (5, 512), (3, 14336), (1, 4096)])
@pytest.mark.parametrize("with_residual", [True, False])
def test_rmsnorm_quant_fp8(dtype, rows, hidden, with_residual):
    """fused_add_rmsnorm + sgl_per_token_quant_fp8 in one launch against the oracle's two steps: the residual is
    bit-exact; a normalised value may sit one 16-bit ulp away (another summation order of the squares), which can
    move an fp8 code by one step."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(rows + hidden)
    x = torch.randn(rows, hidden, generator=g).to(dtype)
    r = torch.randn(rows, hidden, generator=g).to(dtype) if with_residual else None
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
    if with_residual:
        want_y, want_r = oracle.fused_add_rmsnorm(x, r, w, 1e-5)
    else:
        want_y, want_r = oracle.rmsnorm(x, w, 1e-5), None
    want_q, want_s = oracle.per_token_quant_fp8(want_y)
    q = torch.empty(rows, hidden, dtype=F8, device=DEV)
    sc = torch.empty(rows, dtype=torch.float32, device=DEV)
    rd = r.to(DEV) if with_residual else None
    ops.rmsnorm_quant_fp8(x.to(DEV), rd, w.to(DEV), q, sc, 1e-5)
    if with_residual:
        assert torch.equal(rd.cpu().view(torch.int16), want_r.view(torch.int16))
    torch.testing.assert_close(sc.cpu(), want_s.flatten(), rtol=1e-2, atol=0)
    deq = q.cpu().float() * sc.cpu()[:, None]
    ref = want_q.float() * want_s.reshape(-1, 1)
    assert (deq - ref).abs().max().item() <= 0.08 * ref.abs().max().item()   # one e4m3 step at the top of the range
    assert (q.cpu().view(torch.uint8) != want_q.view(torch.uint8)).float().mean().item() < 0.02


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,d", [(64, 3584), (7, 14336), (2, 128)])
def test_silu_and_mul_quant_fp8(dtype, rows, d):
    """silu_and_mul + sgl_per_token_quant_fp8 in one launch, bit for bit the two-step result."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows + d)
    x = torch.randn(rows, 2 * d, generator=g).to(dtype).to(DEV)
    y = torch.empty(rows, d, dtype=dtype, device=DEV)
    ops.act_and_mul(y, x, 0)
    q2 = torch.empty(rows, d, dtype=F8, device=DEV)
    s2 = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.per_token_quant_fp8(y, q2, s2)
    q = torch.empty(rows, d, dtype=F8, device=DEV)
    sc = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.silu_and_mul_quant_fp8(x, q, sc)
    assert torch.equal(sc.cpu(), s2.cpu())
    assert torch.equal(q.cpu().view(torch.uint8), q2.cpu().view(torch.uint8))


@pytest.mark.parametrize("M,N,K", [(64, 4608, 7168), (16, 512, 7168), (100, 1536, 1024), (64, 2112, 2048)])
def test_fp8_blockwise_split_k(M, N, K):
    """Block-scaled product with a workspace (K split over blocks) against the single-launch form and the oracle."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a, b_nk = _fp8_rand((M, K), g), _fp8_rand((N, K), g)
    kb, nb = (K + 127) // 128, (N + 127) // 128
    sa = torch.rand(M, kb, generator=g) * 0.1 + 0.01
    sb = torch.rand(kb, nb, generator=g) * 0.1 + 0.01
    ws = torch.empty(max(ops.fp8_mm_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=DEV)
    args = (a.to(DEV), b_nk.to(DEV).t(), sa.to(DEV), sb.to(DEV), torch.bfloat16)
    out = ops.fp8_blockwise_scaled_mm(*args, ws).float().cpu()
    one = ops.fp8_blockwise_scaled_mm(*args).float().cpu()
    want = oracle.fp8_blockwise_mm(a, b_nk, sa, sb.t().contiguous(), torch.bfloat16).float()
    scale = want.abs().max().item()
    assert (out - want).abs().max().item() <= 1e-2 * scale
    assert (out - one).abs().max().item() <= 1e-2 * scale


def test_fp8_scaled_mm_rejects_bad_shapes():
    ops = _ops()
    from sglang_security_amd._lib import SglkError

    a = torch.zeros(4, 24, dtype=F8, device=DEV)  # K % 16 != 0
    b = torch.zeros(32, 24, dtype=F8, device=DEV)
    s = torch.ones(32, device=DEV)
    with pytest.raises(SglkError):
        ops.fp8_scaled_mm(a, b.t(), s[:4], s, torch.bfloat16)
    with pytest.raises(SglkError):  # row-major B
        ops.fp8_scaled_mm(torch.zeros(4, 32, dtype=F8, device=DEV), torch.zeros(32, 32, dtype=F8, device=DEV), s[:4], s,
                          torch.bfloat16)


def test_fp8_blockwise_golden(golden):
    ops = _ops()
    g = golden("fp8_block_mm")
    out = ops.fp8_blockwise_scaled_mm(g["a"].to(DEV), g["b_nk"].to(DEV).t(), g["scale_a"].to(DEV),
                                      g["scale_b_kn"].to(DEV), torch.bfloat16)
    torch.testing.assert_close(out.cpu().float(), g["expect"].float(), rtol=0.02, atol=1)


@pytest.mark.parametrize("M,N,K", [(1, 128, 128), (16, 512, 7168), (100, 1536, 1024), (64, 2112, 512), (257, 256, 384)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("a_col_major", [False, True])
def test_fp8_blockwise_vs_oracle(M, N, K, out_dtype, a_col_major):
    """sgl-kernel/tests/test_fp8_blockwise_gemm.py:62-90: scale_a column-major (M-contiguous) as the reference
    passes it, scale_b [K/128, N/128]; N need not be a multiple of 128 (2112 = 16.5 blocks)."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a, b_nk = _fp8_rand((M, K), g), _fp8_rand((N, K), g)
    kb, nb = K // 128, (N + 127) // 128
    sa = torch.rand(M, kb, generator=g) * 0.2 + 0.01
    sb_nk = torch.rand(nb, kb, generator=g) * 0.2 + 0.01
    want = oracle.fp8_blockwise_mm(a, b_nk, sa, sb_nk, out_dtype)
    sa_d = sa.to(DEV)
    if a_col_major:
        sa_d = sa_d.t().contiguous().t()
    out = ops.fp8_blockwise_scaled_mm(a.to(DEV), b_nk.to(DEV).t(), sa_d, sb_nk.to(DEV).t(), out_dtype)
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=2e-2, atol=2e-2)


def test_fp8_split_sums_handed_to_consumers():
    """The raw split sums of the streaming product consumed by the rotation kernel and by the fused add + norm +
    quantisation kernel give the bits of the product's own reduction launch followed by the plain kernels."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    M, K, N = 64, 4096, 4096
    a, b_nk = _fp8_rand((M, K), g).to(DEV), _fp8_rand((N, K), g).to(DEV)
    sa = (torch.rand(M, generator=g) * 0.1).to(DEV)
    sb = (torch.rand(N, generator=g) * 0.1).to(DEV)
    packed = ops.fp8_repack(b_nk)
    splits = ops.fp8_mm_packed_partial_splits(M, N, K)
    assert splits > 1
    ws = torch.empty(ops.fp8_mm_packed_workspace_bytes(M, N, K), dtype=torch.uint8, device=DEV)
    full = ops.fp8_scaled_mm_packed(a, packed, sa, sb, torch.bfloat16, None, None, ws)
    part = torch.empty(splits * M * N, dtype=torch.float32, device=DEV)
    assert ops.fp8_mm_packed_partial(a, packed, part) == splits
    # fused add + norm + quantisation
    res0 = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
    w = (1 + 0.1 * torch.randn(N, generator=g)).to(torch.bfloat16).to(DEV)
    q_ref = torch.empty(M, N, dtype=F8, device=DEV)
    s_ref = torch.empty(M, dtype=torch.float32, device=DEV)
    r_ref = res0.clone()
    ops.rmsnorm_quant_fp8(full, r_ref, w, q_ref, s_ref, 1e-5)
    q = torch.empty_like(q_ref)
    s = sa.clone()                                 # the product's activation scales, replaced by the kernel's own
    r = res0.clone()
    ops.rmsnorm_quant_fp8_partial(part, splits, s, sb, r, w, q, s, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(r, r_ref) and torch.equal(s, s_ref) and torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8))
    # gated activation + quantisation: the product read as [gate | up] rows of 2 x 2048
    qa_ref = torch.empty(M, N // 2, dtype=F8, device=DEV)
    sa_ref = torch.empty(M, dtype=torch.float32, device=DEV)
    ops.silu_and_mul_quant_fp8(full, qa_ref, sa_ref)
    qa = torch.empty_like(qa_ref)
    s2 = sa.clone()
    ops.silu_and_mul_quant_fp8_partial(part, splits, s2, sb, qa, s2, torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(s2, sa_ref) and torch.equal(qa.view(torch.uint8), qa_ref.view(torch.uint8))
    # rotation + pool write: the product read as a qkv row of 24 + 2 x 4 heads of 128
    Hq, Hkv, D = 24, 4, 128
    cs = torch.randn(64, D, generator=g).to(DEV)
    pos = torch.randint(0, 64, (M,), generator=g).to(DEV)
    loc = (torch.randperm(100, generator=g)[:M] + 1).to(DEV)
    kb_ref = torch.zeros(128, Hkv, D, dtype=torch.bfloat16, device=DEV)
    vb_ref = torch.zeros_like(kb_ref)
    qkv_ref = full.clone()
    ops.rope_and_store_kv(qkv_ref, cs, pos, kb_ref, vb_ref, loc, Hq, Hkv, D)
    kb, vb = torch.zeros_like(kb_ref), torch.zeros_like(kb_ref)
    qkv = torch.empty_like(full)
    ops.rope_and_store_kv_partial(qkv, part, splits, cs, pos, kb, vb, loc, Hq, Hkv, D, row_scale=sa, col_scale=sb)
    torch.cuda.synchronize()
    assert torch.equal(qkv, qkv_ref) and torch.equal(kb, kb_ref) and torch.equal(vb, vb_ref)


def test_fp8_decode_step_hand_over_matches_plain():
    """The fp8 decode step with the streaming layout and the split sums handed to their consumers against the same
    step on the row-major kernels (tiny model with K >= 4096 would not be tiny: the packing threshold is lowered)."""
    from sglang_security_amd import decode_runner
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = LlamaConfig.tiny("fp8")
    decs = []
    for packed in (False, True):
        dec = LlamaDecoder(cfg, batch=6, pool_slots=1 + 64 + 6 * 24, max_ctx=96, device=DEV, seed=0)
        dec.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
        dec.enable_shared_prefix(True)
        if packed:
            for L in dec.layers:                      # every projection on the streaming layout, whatever its K
                for name in ("qkv", "o", "gate_up", "down"):
                    L[name].wpacked = _ops().fp8_repack(L[name].weight)
            need = max(_ops().fp8_mm_packed_partial_splits(6, L[n].N, L[n].K) * 6 * L[n].N for n in ("qkv", "o", "down"))
            dec.fp8_partial = torch.empty(need, dtype=torch.float32, device=DEV)
        decs.append(dec)
    ref, fus = decs
    for _ in range(3):
        ref.step()
        fus.step()
        torch.cuda.synchronize()
        l0, l1 = ref.logits.float().cpu(), fus.logits.float().cpu()
        assert torch.isfinite(l1).all()
        assert (l0 - l1).abs().max().item() < 0.05 * max(1.0, l0.abs().max().item())
        fus.input_ids.copy_(ref.input_ids)


@pytest.mark.parametrize("M,N,K", [(64, 7168, 16384), (64, 36864, 7168), (64, 7168, 18432), (64, 2112, 7168), (5, 576, 4096),
                                   (100, 256, 2048), (1, 16, 128)])
@pytest.mark.parametrize("with_ws", [True, False])
def test_fp8_blockwise_scaled_mm_packed(M, N, K, with_ws):
    """The block-scaled product on the streaming layout (the dense projections of a DeepSeek-V3 layer, a few odd
    shapes, with and without the split-K workspace) against the oracle and the row-major kernel."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a, b_nk = _fp8_rand((M, K), g), _fp8_rand((N, K), g)
    kb, nb = K // 128, (N + 127) // 128
    sa = torch.rand(M, kb, generator=g) * 0.05 + 0.01
    sb_nk = torch.rand(nb, kb, generator=g) * 0.05 + 0.01
    packed = ops.fp8_repack(b_nk.to(DEV))
    ws = torch.empty(max(ops.fp8_mm_packed_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=DEV) if with_ws else None
    out = ops.fp8_blockwise_scaled_mm_packed(a.to(DEV), packed, sa.to(DEV), sb_nk.to(DEV).t(), torch.bfloat16, ws)
    ref = ops.fp8_blockwise_scaled_mm(a.to(DEV), b_nk.to(DEV).t(), sa.to(DEV), sb_nk.to(DEV).t(), torch.bfloat16)
    want = oracle.fp8_blockwise_mm(a, b_nk, sa, sb_nk, torch.bfloat16)
    scale = want.float().abs().max().item()
    torch.testing.assert_close(out.cpu().float(), want.float(), rtol=2e-2, atol=2e-2 * max(scale, 1.0))
    torch.testing.assert_close(out.cpu().float(), ref.cpu().float(), rtol=1e-2, atol=1e-2 * max(scale, 1.0))


@pytest.mark.parametrize("serialized", [False, True])
@pytest.mark.parametrize("K,N", [(4096, 1024), (512, 256)])
def test_w8a8_fp8_linear_method(serialized, K, N):
    """The W8A8Fp8LinearMethod mirror (w8a8_fp8.py:100-182): weights quantised per output channel at load time (or taken
    from an fp8-serialised checkpoint), activations per token in apply; decode-size batches on the streaming copy where
    it exists (K >= 4096), larger ones on the row-major kernel; against the oracle's quantise + scaled product."""
    import oracle
    from sglang_security_amd.quantization import W8A8Fp8Config, W8A8Fp8LinearMethod

    g = torch.Generator().manual_seed(K + N)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    wq_ref, ws_ref = oracle.per_token_quant_fp8(w)                 # one scale per output channel
    method = W8A8Fp8LinearMethod(W8A8Fp8Config(is_checkpoint_fp8_serialized=serialized))
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.bfloat16, device=DEV)
    if serialized:
        assert layer.weight.dtype == F8 and tuple(layer.weight_scale.shape) == (N, 1)
        layer.weight.data.copy_(wq_ref.to(DEV))
        layer.weight_scale.data.copy_(ws_ref.reshape(N, 1).to(DEV))
    else:
        assert layer.weight.dtype == torch.bfloat16 and layer.weight_scale is None
        layer.weight.data.copy_(w.to(DEV))
    method.process_weights_after_loading(layer)
    assert tuple(layer.weight.shape) == (K, N) and layer.weight.dtype == F8 and tuple(layer.weight_scale.shape) == (N, 1)
    assert torch.equal(layer.weight.t().contiguous().view(torch.uint8).cpu(), wq_ref.view(torch.uint8))
    assert torch.equal(layer.weight_scale.cpu().reshape(-1), ws_ref.reshape(-1))
    assert (layer.sglk_packed is not None) == (K >= 4096)
    for m in (5, 300):
        x = torch.randn(2, m, K, generator=g).to(torch.bfloat16)
        xq, xs = oracle.per_token_quant_fp8(x.reshape(-1, K))
        want = oracle.fp8_scaled_mm(xq, wq_ref.t(), xs.reshape(-1, 1), ws_ref.reshape(-1, 1), torch.bfloat16, bias)
        out = method.apply(layer, x.to(DEV), bias.to(DEV))
        assert tuple(out.shape) == (2, m, N) and out.dtype == torch.bfloat16
        torch.testing.assert_close(out.reshape(-1, N).cpu().float(), want.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("rows,hidden", [(1, 128), (5, 1536), (64, 7168), (3, 18432)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_residual", [False, True])
def test_rmsnorm_group_quant_matches_the_two_calls(rows, hidden, dtype, with_residual):
    """The norm with the 128-group quantisation folded in against fused_add_rmsnorm / rmsnorm followed by
    sgl_per_token_group_quant_fp8 (to the bar of test_rmsnorm_quant_fp8); rows may be a column slice."""
    ops = _ops()
    torch.manual_seed(rows + hidden)
    wide = (torch.randn(rows, hidden + 64) * 2).to(dtype).to(DEV)
    x = wide[:, :hidden]                                           # rows a pitch apart
    x[0, :128] = 0                                                 # an all-zero group exercises eps
    w = (1 + 0.1 * torch.randn(hidden)).to(dtype).to(DEV)
    res = (torch.randn(rows, hidden)).to(dtype).to(DEV) if with_residual else None
    xc = x.clone().contiguous()
    if with_residual:
        res_want = res.clone()
        ops.fused_add_rmsnorm(xc, res_want, w, 1e-6)
    else:
        ops.rmsnorm(xc, xc, w, 1e-6)
    want_q = torch.empty(rows, hidden, dtype=F8, device=DEV)
    want_s = torch.empty(rows, hidden // 128, dtype=torch.float32, device=DEV)
    ops.per_token_group_quant_fp8(xc, want_q, want_s, 128, 1e-10, -448.0, 448.0)
    q = torch.empty(rows, hidden, dtype=F8, device=DEV)
    s = torch.empty(rows, hidden // 128, dtype=torch.float32, device=DEV)
    res_got = res.clone() if with_residual else None
    ops.rmsnorm_group_quant_fp8(x, res_got, w, q, s, 1e-6)
    # the residual is bit-exact; a normalised value may sit one 16-bit ulp away (another summation order of the
    # squares), which can move a group's scale or an fp8 code by one step
    if with_residual:
        assert torch.equal(res_got, res_want)
    torch.testing.assert_close(s, want_s, rtol=1e-2, atol=1e-12)
    deq = q.float().view(rows, -1, 128) * s[..., None]
    ref = want_q.float().view(rows, -1, 128) * want_s[..., None]
    assert (deq - ref).abs().max().item() <= 0.08 * ref.abs().max().item()
    assert (_u8(q) != _u8(want_q)).float().mean().item() < 0.02
    with pytest.raises(Exception):
        ops.rmsnorm_group_quant_fp8(x[:, :64], None, w[:64], q[:, :64].contiguous(), s[:, :1].contiguous(), 1e-6)


@pytest.mark.parametrize("rows,d", [(1, 128), (7, 2048), (64, 18432)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_silu_and_mul_group_quant_matches_the_two_calls(rows, d, dtype):
    ops = _ops()
    torch.manual_seed(rows + d)
    x = (torch.randn(rows, 2 * d) * 2).to(dtype).to(DEV)
    x[0, :128] = 0
    act = torch.empty(rows, d, dtype=dtype, device=DEV)
    ops.act_and_mul(act, x, 0)
    want_q = torch.empty(rows, d, dtype=F8, device=DEV)
    want_s = torch.empty(rows, d // 128, dtype=torch.float32, device=DEV)
    ops.per_token_group_quant_fp8(act, want_q, want_s, 128, 1e-10, -448.0, 448.0)
    q = torch.empty(rows, d, dtype=F8, device=DEV)
    s = torch.empty(rows, d // 128, dtype=torch.float32, device=DEV)
    ops.silu_and_mul_group_quant_fp8(x, q, s)
    # bit for bit the two-step result, except that an exact zero may carry the other sign (codes 0x00 / 0x80)
    unsigned_zero = lambda t: torch.where(_u8(t) == 0x80, torch.zeros_like(_u8(t)), _u8(t))
    assert torch.equal(s, want_s) and torch.equal(unsigned_zero(q), unsigned_zero(want_q))
