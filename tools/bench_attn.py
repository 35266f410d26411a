"""Micro-benchmark of the shared-prefix decode attention on the benchmark's shape (batch 64, 32 / 8 heads of 128,
1024-token shared prefix + `tail` private tokens per request), rotating over several KV pools, one graph of launches."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--prefix", type=int, default=1024)
ap.add_argument("--tail", type=int, default=128)
ap.add_argument("--splits", type=int, default=16)
ap.add_argument("--pools", type=int, default=8)
ap.add_argument("--iters", type=int, default=64)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--head-major", action="store_true",
                help="experiment: pools laid out [Hkv][tokens][D] (a kv head's rows contiguous) instead of [tokens][Hkv][D]")
ap.add_argument("--rope", action="store_true",
                help="the launch that assembles q and the new K / V rows from split-K partial sums of the qkv projection")
ap.add_argument("--planned", action="store_true", help="the kernel driven by a per-step plan laid out on the device")
ap.add_argument("--stamps", action="store_true",
                help="phase stamps of one launch (library built by ATTN=1 tools/build_abl.sh, named by SGLK_LIB)")
args = ap.parse_args()
dev = "cuda"
B, Hq, Hkv, D = args.batch, 32, 8, 128
P, Tl = args.prefix, args.tail
ntok = 1 + P + B * Tl
gen = torch.Generator(device=dev).manual_seed(0)
def _pool():
    if args.head_major:
        return torch.randn(Hkv, ntok, D, device=dev, generator=gen).to(torch.bfloat16).permute(1, 0, 2)
    return torch.randn(ntok, Hkv, D, device=dev, generator=gen).to(torch.bfloat16)


pools = [(_pool(), _pool()) for _ in range(args.pools)]
q = torch.randn(B, Hq, D, device=dev, generator=gen).to(torch.bfloat16)
o = torch.empty_like(q)
rows = []
for b in range(B):
    rows.append(torch.cat([torch.arange(1, 1 + P), torch.arange(1 + P + b * Tl, 1 + P + (b + 1) * Tl)]))
ind = torch.cat(rows).to(torch.int32).to(dev)
ptr = (torch.arange(B + 1, dtype=torch.int32) * (P + Tl)).to(dev)
gi = torch.tensor([0, B], dtype=torch.int32, device=dev)
pls = torch.tensor([P], dtype=torch.int32, device=dev)
ws = ops.decode_shared_workspace(B, Hq, D, args.splits, dev)
xs = torch.zeros(Hq, B, dtype=torch.float32, device=dev)
sm = 1.0 / math.sqrt(D)


row_len = (Hq + 2 * Hkv) * D
qsplits = 4
parts = [torch.randn(qsplits, B, row_len, device=dev, generator=gen) for _ in range(args.pools if args.rope else 0)]
cos_sin = torch.randn(8192, D, device=dev, generator=gen)
positions = torch.full((B,), P + Tl - 1, dtype=torch.int64, device=dev)
loc = ind[(ptr[1:] - 1).long()].to(torch.int64)


plan = None
if args.planned:
    r2t = torch.zeros(B, P + Tl + 4, dtype=torch.int32, device=dev)
    for b in range(B):
        r2t[b, : P + Tl] = rows[b].to(dev)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, args.splits, dev)
    ops.decode_shared_plan(r2t, torch.arange(B, device=dev), torch.full((B,), P + Tl, device=dev), ptr, ind, Hq, Hkv,
                           args.splits, plan)


def run(i):
    kb, vb = pools[i % args.pools]
    if plan is not None:
        ops.decode_attention_shared_planned(q, kb, vb, o, ptr, ind, plan, ws, args.splits, sm, 0.0, xs)
        return
    if args.rope:
        ops.decode_attention_shared_prefix_rope(parts[i % args.pools], qsplits, B * row_len, None, cos_sin, positions, loc,
                                                kb, vb, o, ptr, ind, gi, pls, ws, args.splits, sm, 0.0, xs)
        return
    ops.decode_attention_shared_prefix_xs(q, kb, vb, o, ptr, ind, gi, pls, ws, args.splits, sm, 0.0, xs)


if args.stamps:
    import ctypes

    import numpy as np

    from sglang_security_amd._lib import lib

    for i in range(args.pools + 1):
        run(i)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.int64)
    fn = lib().sglk_debug_attn_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn(buf.ctypes.data_as(ctypes.c_void_p))
    st_ = buf.reshape(1024, 8).astype(np.float64) / 100.0  # us
    heads = Hq // Hkv
    col_blocks = (B * heads + 63) // 64
    sp = int(os.environ.get("SGLK_SHARED_SPLITS", "0")) or max(1, min(args.splits, 256 // (col_blocks * Hkv)))
    n_prefix = sp * col_blocks * Hkv
    n_all = min(n_prefix + B * Hkv, 1024)
    st_ = st_[:n_all]
    # the 100 MHz counters of the eight XCDs carry constant offsets: align every XCD's earliest block start
    xcd = np.arange(n_all) % 8
    for x in range(8):
        st_[xcd == x] -= st_[xcd == x][:, 0].min()

    def show(name, rows, cols):
        if len(rows) == 0:
            return
        print(f"{name}: {len(rows)} blocks")
        for label, a_, b_ in cols:
            d = rows[:, b_] - (rows[:, a_] if a_ >= 0 else 0)
            print(f"   {label:28s} mean {d.mean():7.2f}  p10 {np.percentile(d, 10):7.2f}  p50 {np.percentile(d, 50):7.2f}"
                  f"  p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f} us")

    show("prefix blocks", st_[:min(n_prefix, n_all)],
         [("start (since launch)", -1, 0), ("stream prefix slice", 0, 1), ("store partials", 1, 2), ("end (since launch)", -1, 2)])
    show("tail blocks", st_[n_prefix:n_all],
         [("start (since launch)", -1, 0), ("walk remainder", 0, 1), ("wait for prefix", 1, 2), ("merge + write", 2, 3),
          ("end (since launch)", -1, 3)])
    sys.exit(0)
if args.eager:
    for i in range(args.iters):
        run(i)
    torch.cuda.synchronize()
    print(f"{args.iters} eager launches", flush=True)
    sys.exit(0)
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
kv_bytes = (P + B * Tl) * Hkv * D * 2 * 2
print(f"shared-prefix decode B={B} prefix={P} tail={Tl} splits={args.splits} "
      f": {us:8.2f} us  ({kv_bytes / us / 1e3:7.1f} GB/s of KV)",
      flush=True)
