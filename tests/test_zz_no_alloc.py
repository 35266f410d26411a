"""sglk_awq_gemm_static: the allocation-free form of sglk_awq_gemm (static device workspace inside the library).

Called through the C ABI, checked bit for bit against sglk_awq_gemm with a caller-provided workspace, on two calls in
a row (the second with a smaller input, so stale workspace contents from the first would show), and for the
rejection of shapes whose workspace does not fit."""
import pytest
import torch

from sglang_security_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _case(M, K, N, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=g)
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=g)
    sc = (torch.rand(K // 128, N, device=dev, generator=g) * 0.01).to(torch.bfloat16)
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    return x, ops.awq_repack(qw), sc, qz


def _static(x, packed, sc, qz, out):
    L = _lib.lib()
    M, K = x.shape
    N = sc.shape[1]
    rc = L.sglk_awq_gemm_static(x.data_ptr(), packed.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, out.data_ptr(), M, N,
                                K, 128, x.stride(0), out.stride(0), _lib.BF16,
                                torch.cuda.current_stream(x.device).cuda_stream)
    return rc


def test_awq_gemm_static_matches_workspace_form_twice():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    assert L.sglk_awq_gemm_static_init() == 0, L.sglk_last_error()
    # split-K shapes (the workspace is really used): Llama-3-8B qkv, then a smaller one
    for (M, K, N, seed) in ((64, 4096, 6144, 1), (24, 2048, 1024, 2)):
        x, packed, sc, qz = _case(M, K, N, seed, dev)
        assert ops.awq_gemm_workspace_bytes(M, N, K) <= L.sglk_awq_gemm_static_max_workspace_bytes()
        want = ops.awq_gemm(x, packed, sc, qz)
        got = torch.full_like(want, float("nan"))
        rc = _static(x, packed, sc, qz, got)
        assert rc == 0, L.sglk_last_error()
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (M, K, N)


def test_awq_gemm_static_rejects_oversize():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    M, K, N = 1024, 16384, 256  # 16 row blocks x 8 splits of fp32 slabs: more than the static workspace holds
    assert ops.awq_gemm_workspace_bytes(M, N, K) > L.sglk_awq_gemm_static_max_workspace_bytes()
    x = torch.zeros(1, 8, dtype=torch.bfloat16, device=dev)  # never read: the size check comes first
    rc = L.sglk_awq_gemm_static(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), M, N, K, 128, K,
                                N, _lib.BF16, None)
    assert rc != 0
    assert b"static" in L.sglk_last_error()
