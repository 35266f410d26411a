"""Time sglk_argmax on the logits of one decode step (64 x 128256, 16-bit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sglang_security_amd import ops

dev = torch.device("cuda:0")
x = torch.randn(64, 128256, device=dev).to(torch.bfloat16)
pool = [x.clone() for _ in range(8)]          # 16 MB each: rotate so that not every call finds its rows in the caches
out = torch.empty(64, dtype=torch.int64, device=dev)
for _ in range(10):
    ops.argmax(x, out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(64):
        ops.argmax(pool[i % 8], out)
g.replay()
torch.cuda.synchronize()
a.record()
for _ in range(10):
    g.replay()
b.record()
torch.cuda.synchronize()
print(f"argmax 64 x 128256 bf16: {a.elapsed_time(b) * 1e3 / 640:.2f} us per call")
assert torch.equal(out, torch.argmax(pool[7], dim=-1))
