"""GPU tests of the GPTQ int4 path: the one-time re-layout into the operands of the AWQ kernels (bit-exact against
the format's restatement) and the linear method on top of it.  vllm, where the reference gets this arithmetic from,
is not available here: parity is unpinned beyond the published checkpoint format."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gptq(K, N, group, dtype, seed=0, max_zero=14):
    g = torch.Generator().manual_seed(seed)
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, generator=g)
    z = torch.randint(0, max_zero + 1, (K // group, N), generator=g)                       # stored zero nibbles
    sh = torch.arange(8) * 4
    qz64 = (z.view(K // group, N // 8, 8) << sh).sum(-1)
    qz = torch.where(qz64 >= 2 ** 31, qz64 - 2 ** 32, qz64).to(torch.int32)
    sc = (torch.rand(K // group, N, generator=g) * 0.02 + 0.001).to(dtype)
    return qw, qz, sc


@pytest.mark.parametrize("K,N,group", [(128, 64, 128), (1024, 256, 128), (4096, 6144, 128), (512, 48, 128), (256, 128, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("offset", [1, 0])
def test_gptq_repack_dequantize_bit_exact(K, N, group, dtype, offset):
    import oracle
    from sglang_security_amd import ops

    qw, qz, sc = _gptq(K, N, group, dtype, seed=K + N, max_zero=14 if offset else 15)
    packed, zeros = ops.gptq_repack(qw.to(DEV), qz.to(DEV), group, offset)
    got = ops.awq_dequantize_packed(packed, sc.to(DEV), zeros).cpu()
    want = oracle.gptq_dequantize(qw, qz, sc, group, offset)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_gptq_repack_rejects_zero_point_16():
    from sglang_security_amd import ops
    from sglang_security_amd.ops import SglkError

    qw, qz, sc = _gptq(128, 64, 128, torch.bfloat16, max_zero=15)
    qz[0, 0] = -1                                                          # every stored nibble 15 -> zero point 16
    with pytest.raises(SglkError, match="zero point"):
        ops.gptq_repack(qw.to(DEV), qz.to(DEV), 128, 1)
    ops.gptq_repack(qw.to(DEV), qz.to(DEV), 128, 0)                        # fits when stored as is (v2)


@pytest.mark.parametrize("M", [1, 64, 200, 300])
def test_gptq_linear_method(M):
    """Decode-sized batches through the fused int4 kernel, larger ones through dequantise + GEMM, both against
    x @ W of the format's restatement (tolerances of the AWQ GEMM tests)."""
    import oracle
    from sglang_security_amd.quantization import GPTQConfig, GPTQLinearMethod

    K, N, group = 1024, 512, 128
    qw, qz, sc = _gptq(K, N, group, torch.bfloat16, seed=M)
    method = GPTQLinearMethod(GPTQConfig(4, group, False))
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.bfloat16, device=DEV)
    layer.qweight.data.copy_(qw)
    layer.qzeros.data.copy_(qz)
    layer.scales.data.copy_(sc)
    method.process_weights_after_loading(layer)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16)
    got = method.apply(layer, x.to(DEV), bias.to(DEV)).float().cpu()
    want = x.float() @ oracle.gptq_dequantize(qw, qz, sc, group).float() + bias.float()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2 * want.abs().max().item())


@pytest.mark.parametrize("M", [3, 64, 300])
def test_gptq_activation_reordering(M):
    """desc_act: g_idx assigns the input rows to groups in a shuffled order.  The method sorts the rows by group once
    and gathers the activation columns to match; the product equals x @ W of the format's restatement with that g_idx
    (tolerances of the AWQ GEMM tests), and equals, bit for bit, the product of the same layer stored in group order
    on the gathered activations."""
    import oracle
    from sglang_security_amd.quantization import GPTQConfig, GPTQLinearMethod

    K, N, group = 512, 128, 128
    qw, qz, sc = _gptq(K, N, group, torch.bfloat16, seed=7 + M)
    g_idx = (torch.randperm(K, generator=torch.Generator().manual_seed(3)) // group).to(torch.int32)
    method = GPTQLinearMethod(GPTQConfig(4, group, True))
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.bfloat16, device=DEV)
    layer.qweight.data.copy_(qw)
    layer.qzeros.data.copy_(qz)
    layer.scales.data.copy_(sc)
    layer.g_idx.data.copy_(g_idx)
    method.process_weights_after_loading(layer)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
    got = method.apply(layer, x.to(DEV)).float().cpu()
    want = x.float() @ oracle.gptq_dequantize(qw, qz, sc, group, 1, g_idx).float()
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2 * want.abs().max().item())
    # the dequantised, re-ordered weight is the restatement's rows in sorted order, bit for bit
    from sglang_security_amd import ops
    perm = torch.argsort(g_idx.long(), stable=True)
    w_sorted = ops.awq_dequantize_packed(layer.sglk_packed, layer.scales, layer.sglk_zeros).cpu()
    ref_rows = oracle.gptq_dequantize(qw, qz, sc, group, 1, g_idx)[perm]
    assert torch.equal(w_sorted.view(torch.int16), ref_rows.view(torch.int16))


def test_gptq_rejects_malformed_g_idx():
    from sglang_security_amd.quantization import GPTQConfig, GPTQLinearMethod

    K, N, group = 256, 64, 128
    qw, qz, sc = _gptq(K, N, group, torch.bfloat16)
    method = GPTQLinearMethod(GPTQConfig(4, group, True))
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.bfloat16, device=DEV)
    layer.qweight.data.copy_(qw)
    layer.qzeros.data.copy_(qz)
    layer.scales.data.copy_(sc)
    bad = torch.zeros(K, dtype=torch.int32)
    bad[:100] = 1                                                          # 156 rows in group 0, 100 in group 1
    layer.g_idx.data.copy_(bad)
    with pytest.raises(ValueError, match="g_idx"):
        method.process_weights_after_loading(layer)
