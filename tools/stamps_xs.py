"""Phase stamps (SGLK_AWQ_ABL=16) of the wide AWQ kernel in the block-sums-supplied form, cold weights."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGLK_AWQ_ABL", "16")
os.environ["SGLK_AWQ_WIDE"] = "2"
from sglang_security_amd import ops
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
M = 64
for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]:
    copies = 6
    x = torch.randn(M, K, device=dev, generator=gen).to(torch.bfloat16)
    xs = x.float().view(M, K // 128, 128).sum(-1).t().contiguous()
    ws = []
    for _ in range(copies):
        qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen)
        sc = (torch.rand(K // 128, N, device=dev, generator=gen) * 0.01 + 0.001).to(torch.bfloat16)
        ws.append((ops.awq_repack(qw), sc, qz))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    work = torch.zeros(max(ops.awq_gemm_workspace_bytes(M, N, K), 32768), dtype=torch.uint8, device=dev)
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for i in range(copies):
        w = ws[i]
        big.fill_(i)  # push everything out of the caches
        torch.cuda.synchronize()
        work.zero_()
        ops.awq_gemm_xs(x, w[0], w[1], w[2], xs, out, work)
        torch.cuda.synchronize()
    t = work.view(torch.int32)[2048:4096].view(256, 8).cpu().double()
    t = t[t[:, 0] != 0]
    if t.shape[0] == 0:
        print(f"{name}: no stamps")
        continue
    t0 = t[:, 0].min()
    names = ["start", "prologue", "loop", "reduce", "stored", "split-reduce"]
    if int(os.environ["SGLK_AWQ_ABL"]) & 32:
        names = ["stage3", "issued", "computed", "landed", "barrier", "end"]
        t0 = None
    msg = [f"{name}: blocks={t.shape[0]}"]
    for k in range(6):
        col = t[:, k]
        ok = col != 0
        if ok.any():
            d = (col[ok] - (t0 if t0 is not None else t[ok][:, 0])) / 100.0
            msg.append(f"{names[k]} {d.mean():.2f}/{d.max():.2f}us")
    print("  ".join(msg), flush=True)
