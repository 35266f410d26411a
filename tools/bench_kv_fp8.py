"""Decode attention over an fp8 KV pool next to the same batch over a bf16 pool (Llama-3-8B heads, batch 64, 1152
tokens per request, per-request walk): microseconds per layer."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

dev = "cuda"
B, Hq, Hkv, D, L, S = 64, 32, 8, 128, 1152, 8
g = torch.Generator(device=dev).manual_seed(0)
n = B * L + 1
q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
kb = torch.randn(n, Hkv, D, device=dev, generator=g).to(torch.bfloat16)
vb = torch.randn(n, Hkv, D, device=dev, generator=g).to(torch.bfloat16)
ptr = torch.arange(B + 1, dtype=torch.int32, device=dev) * L
ind = (torch.randperm(n - 1, device=dev, generator=g)[: B * L] + 1).to(torch.int32)
ns = torch.full((B,), S, dtype=torch.int32, device=dev)
o = torch.empty_like(q)
al = torch.empty(B, Hq, S, D, dtype=torch.float32, device=dev)
lse = torch.empty(B, Hq, S, dtype=torch.float32, device=dev)


def timeit(k, v, iters=20):
    for _ in range(3):
        ops.decode_attention_fwd(q, k, v, o, ptr, ind, al, lse, ns, S, 1 / math.sqrt(D))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.decode_attention_fwd(q, k, v, o, ptr, ind, al, lse, ns, S, 1 / math.sqrt(D))
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


t16 = timeit(kb, vb)
print(f"bf16 pool       {t16:8.1f} us  ({2 * B * L * Hkv * D * 2 / t16 / 1e6:6.2f} TB/s of rows)")
for name, dt in (("e5m2fnuz", torch.float8_e5m2fnuz), ("e5m2", torch.float8_e5m2), ("e4m3fn", torch.float8_e4m3fn)):
    t8 = timeit(kb.to(dt), vb.to(dt))
    print(f"{name:9s} pool  {t8:8.1f} us  ({2 * B * L * Hkv * D / t8 / 1e6:6.2f} TB/s of rows)")
