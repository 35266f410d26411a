"""Micro-benchmark of absorbed-MLA decode (latent rows of 576, values = first 512) at DeepSeek-V3's per-rank shape
under TP8 (16 heads) or unsharded (128 heads): one graph of launches rotating over several latent pools."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--ctx", type=int, default=1152)
ap.add_argument("--heads", type=int, default=16)
ap.add_argument("--splits", type=int, default=8)
ap.add_argument("--pools", type=int, default=4)
ap.add_argument("--iters", type=int, default=32)
ap.add_argument("--generic", action="store_true", help="separate value copy: the generic kernel")
ap.add_argument("--prefix", type=int, default=0, help="the first PREFIX tokens of every request are the same pool rows")
args = ap.parse_args()
dev = "cuda"
B, Hq, L = args.batch, args.heads, args.ctx
ntok = 1 + B * L
gen = torch.Generator(device=dev).manual_seed(0)
pools = [torch.randn(ntok, 1, 576, device=dev, generator=gen).to(torch.bfloat16) for _ in range(args.pools)]
vals = [p[..., :512].contiguous() if args.generic else p[..., :512] for p in pools]
q = torch.randn(B, Hq, 576, device=dev, generator=gen).to(torch.bfloat16)
o = torch.empty(B, Hq, 512, dtype=torch.bfloat16, device=dev)
ind = (torch.randperm(ntok - 1, device=dev, generator=gen)[:B * L] + 1).to(torch.int32)
if args.prefix:
    ind = ind.view(B, L)
    ind[:, :args.prefix] = ind[0, :args.prefix].clone()
    ind = ind.reshape(-1).contiguous()
ptr = (torch.arange(B + 1, dtype=torch.int32) * L).to(dev)
ns = torch.zeros(B, dtype=torch.int32, device=dev)
ops.get_num_kv_splits(ns, torch.full((B,), L, device=dev), Hq, 1, args.splits, 256)
al = torch.empty(B, Hq, args.splits, 512, dtype=torch.float32, device=dev)
lse = torch.empty(B, Hq, args.splits, dtype=torch.float32, device=dev)
sm = 1.0 / math.sqrt(192)


def run(i):
    ops.decode_attention_fwd(q, pools[i % args.pools], vals[i % args.pools], o, ptr, ind, al, lse, ns, args.splits, sm, 0.0)


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for i in range(args.pools):
        run(i)
    st.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for i in range(args.iters):
            run(i)
    gr.replay()
    st.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    gr.replay()
    b_.record(st)
    st.synchronize()
us = a.elapsed_time(b_) / args.iters * 1e3
kv_bytes = B * L * 576 * 2
print(f"MLA decode B={B} ctx={L} heads={Hq} splits={args.splits} (num_kv_splits {int(ns[0])}) "
      f"{'generic' if args.generic else 'matrix-core'}: {us:9.2f} us  ({kv_bytes / us / 1e3:7.1f} GB/s of latent rows)",
      flush=True)
