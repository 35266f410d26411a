"""GPU: the vllm stand-ins the reference imports on ROCm (sglang_security_amd/vllm_shim.py), called the way the
reference calls them (layernorm.py:73-85, activation.py:168-172, rotary_embedding.py:166-174,
fp8_utils.py:525-530,613-635), against the same golden vectors and oracle as the sgl_kernel mirror.  vllm itself is
not in the reference tree: these compare with the reference's formulas, not with vllm's binaries."""
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
F8 = torch.float8_e4m3fn


@pytest.fixture()
def vops():
    from sglang_security_amd import vllm_shim

    saved = {k: v for k, v in sys.modules.items() if k == "vllm" or k.startswith("vllm.")}
    vllm_shim.install(force=True)
    from vllm import _custom_ops as ops

    yield ops
    for k in [k for k in sys.modules if k == "vllm" or k.startswith("vllm.")]:
        del sys.modules[k]
    sys.modules.update(saved)


@pytest.mark.parametrize("name", ["norm_f16", "norm_bf16", "norm_odd"])
def test_rms_norm_as_rmsnorm_forward_hip(vops, golden, name):
    g = golden(name)
    x, r, w = g["x"].to(DEV), g["residual"].to(DEV), g["weight"].to(DEV)
    out = torch.empty_like(x)
    vops.rms_norm(out, x, w, g["eps"])                       # layernorm.py:84-85
    torch.testing.assert_close(out.cpu().float(), g["expect_rmsnorm"].float(), rtol=1e-3, atol=1e-3)
    x2, r2 = x.clone(), r.clone()
    vops.fused_add_rms_norm(x2, r2, w, g["eps"])             # layernorm.py:81-82
    torch.testing.assert_close(x2.cpu().float(), g["expect_fused_x"].float(), rtol=1e-3, atol=1e-3)
    assert torch.equal(r2.cpu().view(torch.int16), g["expect_fused_residual"].view(torch.int16))


@pytest.mark.parametrize("name", ["silu_f16", "silu_bf16"])
def test_silu_and_mul_module(vops, golden, name):
    from vllm.model_executor.layers.activation import GeluAndMul, SiluAndMul

    g = golden(name)
    x = g["x"].to(DEV)
    out = SiluAndMul()(x)                                    # activation.py:172
    torch.testing.assert_close(out.cpu().float(), g["expect"].float(), rtol=1e-3, atol=1e-3)
    x3 = x.reshape(2, x.shape[0] // 2, x.shape[1])           # leading batch dimensions
    torch.testing.assert_close(SiluAndMul()(x3).reshape(out.shape).cpu().float(), g["expect"].float(), rtol=1e-3, atol=1e-3)
    xf = x.float().cpu()
    d = x.shape[1] // 2
    for approx in ("none", "tanh"):
        want = torch.nn.functional.gelu(xf[:, :d], approximate=approx) * xf[:, d:]
        torch.testing.assert_close(GeluAndMul(approx)(x).cpu().float(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("name", ["rope_neox128", "rope_gptj128", "rope_partial"])
@pytest.mark.parametrize("cache_in_activation_dtype", [False, True])
def test_rotary_embedding(vops, golden, name, cache_in_activation_dtype):
    import oracle

    g = golden(name)
    q, k = g["q"].to(DEV), g["k"].to(DEV)
    cache = g["cos_sin_cache"]
    if cache_in_activation_dtype:                            # rotary_embedding.py:83-84,166: the non-CUDA path casts it
        cache = cache.to(q.dtype)
        want_q = oracle.rope(g["q"], g["positions"], cache.float(), g["head_size"], g["is_neox"])
        want_k = oracle.rope(g["k"], g["positions"], cache.float(), g["head_size"], g["is_neox"])
    else:
        want_q, want_k = g["expect_q"], g["expect_k"]
    vops.rotary_embedding(g["positions"].to(DEV), q, k, g["head_size"], cache.to(DEV), g["is_neox"])
    torch.testing.assert_close(q.cpu().float(), want_q.float(), atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(k.cpu().float(), want_k.float(), atol=1e-2, rtol=1e-2)


def _u8(t):
    return t.cpu().view(torch.uint8)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_scaled_fp8_quant_modes(vops, dtype):
    import oracle

    torch.manual_seed(3)
    x = (torch.randn(37, 1024) * 3).to(dtype)
    # dynamic per tensor (fp8_utils.py:362): one scale = absmax / 448
    q, s = vops.scaled_fp8_quant(x.to(DEV))
    scale = x.float().abs().max() / 448.0
    assert s.shape == (1,) and s.cpu().item() == scale.item()
    assert torch.equal(_u8(q), oracle.e4m3_encode((x.float() * (1.0 / scale)).clamp(-448, 448)).view(torch.uint8))
    # dynamic per token (fp8_utils.py:614-618)
    q, s = vops.scaled_fp8_quant(x.to(DEV), None, use_per_token_if_dynamic=True)
    wq, ws = oracle.per_token_quant_fp8(x)
    assert s.shape == (37, 1) and torch.equal(s.cpu().reshape(-1), ws.reshape(-1))
    assert torch.equal(_u8(q), wq.view(torch.uint8))
    # static (a given scale), padded rows zero (fp8_utils.py:526-530)
    given = torch.tensor([0.0625], device=DEV)
    q, s = vops.scaled_fp8_quant(x.to(DEV), given, num_token_padding=48)
    assert q.shape == (48, 1024) and s is given
    assert torch.equal(_u8(q[:37]), oracle.e4m3_encode((x.float() * 16.0).clamp(-448, 448)).view(torch.uint8))
    assert int(_u8(q[37:]).max()) == 0


def test_cutlass_scaled_mm_per_tensor_and_per_channel(vops):
    import oracle

    g = torch.Generator().manual_seed(5)
    M, N, K = 24, 512, 1024
    a = (torch.randn(M, K, generator=g) * 0.5).to(F8)
    b_nk = (torch.randn(N, K, generator=g) * 0.5).to(F8)
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    sa = torch.rand(M, 1, generator=g) * 0.02 + 0.01
    sb = torch.rand(N, 1, generator=g) * 0.02 + 0.01
    got = vops.cutlass_scaled_mm(a.to(DEV), b_nk.to(DEV).t(), out_dtype=torch.bfloat16, scale_a=sa.to(DEV),
                                 scale_b=sb.to(DEV), bias=bias.to(DEV))
    want = oracle.fp8_scaled_mm(a, b_nk.t(), sa.reshape(-1), sb.reshape(-1), torch.bfloat16, bias)
    torch.testing.assert_close(got.cpu().float(), want.float(), rtol=2e-2, atol=1.0)   # the reference's fp8 GEMM bar
    one_a, one_b = torch.tensor([0.015]), torch.tensor([0.02])
    got = vops.cutlass_scaled_mm(a.to(DEV), b_nk.to(DEV).t(), out_dtype=torch.bfloat16, scale_a=one_a.to(DEV),
                                 scale_b=one_b.to(DEV), bias=None)
    want = oracle.fp8_scaled_mm(a, b_nk.t(), one_a.expand(M).contiguous(), one_b.expand(N).contiguous(), torch.bfloat16, None)
    torch.testing.assert_close(got.cpu().float(), want.float(), rtol=2e-2, atol=1.0)


def test_awq_dequantize_spelling(vops):
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (256, 32), dtype=torch.int32)
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (2, 32), dtype=torch.int32)
    sc = (torch.rand(2, 256) * 0.01).to(torch.float16)
    import oracle

    got = vops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV), 0, 0, 0)
    assert torch.equal(got.cpu().view(torch.int16), oracle.awq_dequantize(qw, sc, qz).view(torch.int16))
