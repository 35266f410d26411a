"""Micro-benchmark of the extend attention (Llama-3-8B heads): one request cold / after a prefix hit, and batch 64."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

dev = "cuda"
Hq, Hkv, D = 32, 8, 128
gen = torch.Generator(device=dev).manual_seed(0)
for name, B, pre, ext in (("one request, cold 1152", 1, 0, 1152), ("one request, 128 after a 1024 hit", 1, 1024, 128),
                          ("batch 64, 128 after a shared 1024 hit", 64, 1024, 128), ("batch 64, cold 1152", 64, 0, 1152)):
    n = 1 + pre + 8
    pools = [(torch.randn(n, Hkv, D, device=dev, generator=gen).to(torch.bfloat16),
              torch.randn(n, Hkv, D, device=dev, generator=gen).to(torch.bfloat16)) for _ in range(4)]
    T = B * ext
    q = torch.randn(T, Hq, D, device=dev, generator=gen).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, device=dev, generator=gen).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, device=dev, generator=gen).to(torch.bfloat16)
    o = torch.empty_like(q)
    qo = (torch.arange(B + 1, dtype=torch.int32) * ext).to(dev)
    kp = (torch.arange(B + 1, dtype=torch.int32) * pre).to(dev)
    ki = torch.arange(1, 1 + pre, dtype=torch.int32).repeat(B).to(dev) if pre else torch.zeros(1, dtype=torch.int32, device=dev)
    sm = 1.0 / math.sqrt(D)
    st = torch.cuda.Stream()
    iters = 16
    with torch.cuda.stream(st):
        def run(i):
            kb, vb = pools[i % 4]
            ops.extend_attention_fwd(q, k, v, o, kb, vb, qo, kp, ki, None, True, None, ext, sm)
        for i in range(4):
            run(i)
        st.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(iters):
                run(i)
        gr.replay()
        st.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        gr.replay()
        b.record(st)
        st.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    flop = 4.0 * B * Hq * D * (ext * pre + ext * (ext + 1) / 2)
    print(f"{name:40s} {us:9.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)
