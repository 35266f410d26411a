"""Micro-benchmark of the fp8 matmuls on the Llama-3-70B TP=8 per-GPU projection shapes (M = 64) and one DeepSeek
block-scaled shape; weights rotate over enough copies to come from HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F8 = torch.float8_e4m3fn
gen = torch.Generator(device=dev).manual_seed(0)
cases = [("qkv", 8192, 1280, False), ("o", 1024, 8192, False), ("gate_up", 8192, 7168, False), ("down", 3584, 8192, False),
         ("block 7168x4608", 7168, 4608, True)]
for name, K, N, block in cases:
    copies = max(2, int(600e6 // (K * N)) + 1)
    pad_a, pad_b = int(os.environ.get("PAD_A", "0")), int(os.environ.get("PAD_B", "0"))  # row pitch experiments
    a = (torch.randn(M, K + pad_a, device=dev, generator=gen) * 0.5).to(F8)[:, :K]
    ws = [(torch.randn(N, K + pad_b, device=dev, generator=gen) * 0.5).to(F8)[:, :K] for _ in range(copies)]
    if block:
        # M-contiguous activation scales, as the reference's quantiser hands them to this product
        sa = (torch.rand(M, K // 128, device=dev, generator=gen) + 0.1).t().contiguous().t()
        sb = torch.rand(K // 128, (N + 127) // 128, device=dev, generator=gen) + 0.1
        bwork = torch.empty(max(ops.fp8_mm_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=dev)
        run = lambda i: ops.fp8_blockwise_scaled_mm(a, ws[i % copies].t(), sa, sb, torch.bfloat16, bwork)
    else:
        sa = torch.rand(M, 1, device=dev, generator=gen)
        sb = torch.rand(N, 1, device=dev, generator=gen)
        work = torch.empty(max(ops.fp8_mm_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=dev)
        outb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        run = lambda i: ops.fp8_scaled_mm(a, ws[i % copies].t(), sa, sb, torch.bfloat16, None, outb, work)
    for i in range(copies):
        run(i)
    torch.cuda.synchronize()
    iters = 40
    torch.cuda._sleep(int(20e-3 * 2e9))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    byts = K * N + M * K + 2 * M * N
    print(f"{name:18s} M={M} K={K:5d} N={N:5d}  {us:8.2f} us  {byts / us / 1e3:8.1f} GB/s  {2 * M * K * N / us / 1e6:7.1f} TFLOP/s", flush=True)
