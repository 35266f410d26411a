"""Per-wave cycle timeline of one steady-state stage of the wide AWQ GEMM (development build with SGLK_ABL bit 1024:
SGLK_LIB=tools/libsglk_abl1024.so python tools/gemm_timeline.py)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import _lib, ops  # noqa: E402

dev = "cuda"
M = 64
gen = torch.Generator(device=dev).manual_seed(0)
L = ctypes.CDLL(_lib.LIB_PATH)
for name, K, N in [("gate_up", 4096, 28672), ("down", 14336, 4096)]:
    x = torch.randn(M, K, device=dev, generator=gen).to(torch.bfloat16)
    ws = []
    for _ in range(6):
        qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen)
        sc = (torch.rand(K // 128, N, device=dev, generator=gen) * 0.01).to(torch.bfloat16)
        ws.append((ops.awq_repack(qw), sc, qz))
        del qw
    xs = torch.zeros(K // 128, M, dtype=torch.float32, device=dev)
    splits = ops.awq_gemm_partial_splits(M, N, K, 128)
    part = torch.empty(max(splits, 1) * M * N, dtype=torch.float32, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
    xs_out = torch.zeros(N // 256, M, dtype=torch.float32, device=dev)
    for i in range(200):  # long enough for the clock to settle under load
        w = ws[i % 6]
        if name == "gate_up":
            ops.awq_gemm_silu_mul(x, w[0], w[1], w[2], act, xs, xs_out)
        else:
            ops.awq_gemm_partial(x, w[0], w[1], w[2], part, xs)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 16, dtype=np.uint32)
    L.sglk_debug_read_timeline(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(256, 8, 16).astype(np.int64)
    t = t[t[:, 0, 0] != 0]
    d = (t[:, :, 1:13] - t[:, :, 0:12]) & 0xFFFFFFFF
    names = ["prep"] + [f"k{s}" for s in range(8)] + ["tail", "vmwait", "barrier"]
    med = np.median(d.reshape(-1, 12), axis=0)
    p90 = np.percentile(d.reshape(-1, 12), 90, axis=0)
    print(f"{name}: blocks={t.shape[0]} stage cycles median {np.median((t[:, :, 12] - t[:, :, 0]) & 0xFFFFFFFF):.0f}")
    wall = ((t[:, :, 15] - t[:, :, 14]) & 0xFFFFFFFF).astype(np.float64) * 10e-9
    cyc = ((t[:, :, 12] - t[:, :, 0]) & 0xFFFFFFFF).astype(np.float64)
    print(f"   clock GHz median {np.median(cyc / wall) / 1e9:.3f}")
    print("   " + "  ".join(f"{n} {m:.0f}/{q:.0f}" for n, m, q in zip(names, med, p90)))
    for w in range(8):
        mw = np.median(d[:, w, :], axis=0)
        print(f"   wave {w}: " + " ".join(f"{v:5.0f}" for v in mw))
