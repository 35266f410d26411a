"""Micro-benchmark of the weight-streaming GEMMs on the Llama-3-8B decode shapes (M = 64).

Rotates over enough distinct weight copies that no launch finds its weights in the 256 MiB Infinity Cache,
times with events on the launch stream, prints GB/s of algorithmic bytes per shape."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--copies", type=int, default=0, help="weight copies to rotate over (1: weights stay in the Infinity Cache)")
ap.add_argument("--eager", action="store_true", help="plain launches, no graph (counter collection)")
ap.add_argument("--fused", action="store_true", help="the forms the fused decode step launches: fp32 split-K partials with the "
                "producer's block sums (qkv, o, down) and the gated-activation epilogue (gate_up)")
ap.add_argument("--packed", action="store_true", help="with --dense: the sequential-streaming weight layout")
ap.add_argument("--stamps", action="store_true", help="per-phase clock stamps of the wide AWQ kernel (run with SGLK_AWQ_ABL=16)")
args = ap.parse_args()
dev = "cuda"
M = args.m
shapes = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]
if args.dense:
    shapes = shapes + [("lm_head", 4096, 128256)]
gen = torch.Generator(device=dev).manual_seed(0)
for name, K, N in shapes:
    wbytes = K * N // 2 + (K // 128) * N * 2 + (K // 128) * N // 2 if not args.dense else K * N * 2
    copies = args.copies or max(2, int(600e6 // wbytes) + 1)
    x = torch.randn(M, K, device=dev, generator=gen).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ws = []
    for _ in range(copies):
        if args.dense:
            wt = (torch.randn(N, K, device=dev, generator=gen) * 0.02).to(torch.bfloat16)
            ws.append((ops.dense_repack(wt) if args.packed and N % 16 == 0 else wt,))
            del wt
        else:
            qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
            qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen)
            sc = (torch.rand(K // 128, N, device=dev, generator=gen) * 0.01).to(torch.bfloat16)
            ws.append((ops.awq_repack(qw), sc, qz))
            del qw
    work = torch.zeros(max(ops.awq_gemm_workspace_bytes(M, N, K), ops.dense_gemm_workspace_bytes(M, N, K) if args.dense else 0,
                           256), dtype=torch.uint8, device=dev)

    if args.fused:
        xs = torch.zeros(K // 128, M, dtype=torch.float32, device=dev)
        splits = ops.awq_gemm_partial_splits(M, N, K, 128)
        part = torch.empty(max(splits, 1) * M * N, dtype=torch.float32, device=dev)
        act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
        xs_out = torch.zeros(N // 256, M, dtype=torch.float32, device=dev)

    def run(i):
        w = ws[i % copies]
        if args.fused and name == "gate_up":
            ops.awq_gemm_silu_mul(x, w[0], w[1], w[2], act, xs, xs_out)
        elif args.fused:
            ops.awq_gemm_partial(x, w[0], w[1], w[2], part, xs)
        elif args.dense and args.packed:
            ops.dense_gemm_packed(x, w[0], None, out)
        elif args.dense:
            ops.dense_gemm_nt(x, w[0], None, out, work)
        else:
            ops.awq_gemm(x, w[0], w[1], w[2], None, out, work)

    if args.stamps and args.fused:
        import ctypes
        import numpy as np
        from sglang_security_amd import _lib
        L = ctypes.CDLL(_lib.LIB_PATH)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        L.sglk_debug_clear_stamps()
        run(3)
        torch.cuda.synchronize()
        buf = np.zeros(256 * 24, dtype=np.int64)
        L.sglk_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
        t = buf.reshape(256, 24).astype(np.float64)
        t = t[t[:, 0] != 0]
        t0 = t[:, 0:1]  # per block: the clocks of different XCDs are not aligned
        cols = ["start", "prologue"] + [f"st{k}" for k in range(10)] + ["drained", "reduced", "stored", "sums_read", "s2:w_landed", "s2:x_landed", "s2:barrier", "p:issued", "p:w0", "p:x0", "e:written", "e:barrier"]
        msg = [f"{name}: blocks={t.shape[0]}"]
        for k in range(24):
            col = t[:, k]
            ok = col != 0
            if ok.any():
                d = (col[ok] - t0[ok, 0]) / 100.0
                msg.append(f"{cols[k]} {np.median(d):.2f}/{d.max():.2f}")
        print("  ".join(msg), flush=True)
        st = (t[:, 0] - t[:, 0].min()) / 100.0
        print("   start by block id:", " ".join(f"{v:.2f}" for v in st[:24]), "...", " ".join(f"{v:.2f}" for v in st[-8:]), flush=True)
        print("   start percentiles 0/1/5/25/50/95/100:", " ".join(f"{np.percentile(st, q):.2f}" for q in (0, 1, 5, 25, 50, 95, 100)),
              flush=True)
        continue
    if args.stamps:
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        work.zero_()
        run(3)
        torch.cuda.synchronize()
        t = work.view(torch.int32)[2048:4096].view(256, 8).cpu().double()
        t = t[t[:, 0] != 0]
        if t.shape[0] == 0:
            print(f"{name}: no stamps", flush=True)
            continue
        t0 = t[:, 0].min()
        names = ["start", "prologue", "loop", "reduce", "stored", "split-reduce"]
        if int(os.environ.get("SGLK_AWQ_ABL", "0")) & 32:
            names = ["stage3", "issued", "computed", "landed", "barrier1", "xsum+barrier2"]
            t0 = None
        msg = [f"{name}: blocks={t.shape[0]}"]
        for k in range(6):
            col = t[:, k]
            ok = col != 0
            if ok.any():
                d = (col[ok] - (t0 if t0 is not None else t[ok][:, 0])) / 100.0
                msg.append(f"{names[k]} {d.mean():.2f}/{d.max():.2f}us")
        print("  ".join(msg), flush=True)
        continue
    if args.eager:
        for i in range(args.iters):
            run(i)
        torch.cuda.synchronize()
        print(f"{name}: {args.iters} eager launches", flush=True)
        continue
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(copies):
            run(i)
        st.synchronize()
        # one graph of `iters` launches: no host launch cost inside the timed region
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(args.iters):
                run(i)
        gr.replay()
        st.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        gr.replay()
        b.record(st)
        st.synchronize()
    us = a.elapsed_time(b) / args.iters * 1e3
    tot = wbytes + 2 * M * (K + N)
    print(f"{name:8s} K={K:6d} N={N:6d} copies={copies:3d}  {us:8.2f} us  {tot / us / 1e3:8.1f} GB/s  "
          f"({2 * M * K * N / us / 1e6:7.1f} TFLOP/s)", flush=True)
