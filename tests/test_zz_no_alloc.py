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


_CAPTURE_FIRST = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from sglang_security_amd import _lib, ops
L = _lib.lib()
dev = torch.device("cuda:0")
M, K, N = 64, 4096, 6144
g = torch.Generator(device=dev).manual_seed(1)
qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=g)
qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=g)
sc = (torch.rand(K // 128, N, device=dev, generator=g) * 0.01).to(torch.bfloat16)
x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
packed = ops.awq_repack(qw)
want = ops.awq_gemm(x, packed, sc, qz)
got = torch.zeros_like(want)
def call(stream):
    return L.sglk_awq_gemm_static(x.data_ptr(), packed.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, got.data_ptr(), M, N,
                                  K, 128, x.stride(0), got.stride(0), _lib.BF16, stream.cuda_stream)
s = torch.cuda.Stream()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(graph, stream=s):
        rc_first = call(s)                      # nothing resolved on this device yet: must be refused, not resolved
        msg = L.sglk_last_error().decode()
assert rc_first != 0 and "sglk_awq_gemm_static_init" in msg, (rc_first, msg)
assert L.sglk_awq_gemm_static_init() == 0
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(graph, stream=s):
        rc = call(s)
assert rc == 0, L.sglk_last_error()
graph.replay()
torch.cuda.synchronize()
assert torch.equal(got.view(torch.int16), want.view(torch.int16))
print("ok")
'''


def test_awq_gemm_static_refuses_a_first_use_under_capture(tmp_path):
    """In a fresh process: a call whose stream is being captured before sglk_awq_gemm_static_init() ran on the device is
    refused (no lock, no symbol look-up under capture); after the init the captured call replays bit-identical to the
    workspace form."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "capture_first.py"
    script.write_text(_CAPTURE_FIRST)
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
