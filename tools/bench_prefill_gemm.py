"""Large-M GEMM timings on the extend shapes of Llama-3-8B (M tokens x the four projections): the hand-written
dequant-fused AWQ kernel, the hand-written dense kernel, and for comparison the dequantise-then-library-GEMM route."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=1152)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = "cuda"
M = args.m
gen = torch.Generator(device=dev).manual_seed(0)


def timeit(fn):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.iters * 1e3


for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]:
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand(K // 128, N, device=dev, generator=gen) * 0.01).to(torch.bfloat16)
    packed = ops.awq_repack(qw)
    del qw
    x = torch.randn(M, K, device=dev, generator=gen).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * M * K * N
    t_fused = timeit(lambda: ops.awq_gemm_prefill(x, packed, sc, qz, None, out))
    wd = ops.awq_dequantize_packed(packed, sc, qz)
    t_deq = timeit(lambda: ops.awq_dequantize_packed(packed, sc, qz))
    t_lib = timeit(lambda: torch.matmul(x, wd, out=out))
    wt = wd.t().contiguous()
    t_dense = timeit(lambda: ops.dense_gemm_prefill(x, wt, None, out))
    print(f"{name:8s} M={M} K={K:6d} N={N:6d}  fused int4 {t_fused:8.1f} us ({flops / t_fused / 1e6:6.0f} TFLOP/s)   "
          f"dense hand-written {t_dense:8.1f} us ({flops / t_dense / 1e6:6.0f})   library {t_lib:8.1f} us "
          f"({flops / t_lib / 1e6:6.0f}) + dequantise {t_deq:7.1f} us", flush=True)
