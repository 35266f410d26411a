"""The one-launch shared-prefix decode kernel next to the per-request kernels on the same batch (64 requests, 1024
shared + 128 private tokens), for the pool formats and head shapes it covers: microseconds per layer, from a graph of
32 launches over 32 different pools' worth of slots (so rows come from HBM, as in a model's layers)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

dev = "cuda"
B, P, T, LAYERS = 64, 1024, 128, 16
S = int(os.environ.get("S", "8"))   # max_kv_splits


def run(name, Hq, Hkv, D, pool_dtype, Dv=None):
    g = torch.Generator(device=dev).manual_seed(0)
    n = 1 + P + B * T
    q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    if Dv is None:
        Dv = D
        pools = [(torch.randn(n, Hkv, D, device=dev, generator=g).to(torch.bfloat16).to(pool_dtype),
                  torch.randn(n, Hkv, D, device=dev, generator=g).to(torch.bfloat16).to(pool_dtype)) for _ in range(LAYERS)]
    else:  # absorbed MLA: the values are a view of the key rows
        ks = [torch.randn(n, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(LAYERS)]
        pools = [(k, k[..., :Dv]) for k in ks]
    r2t = torch.zeros(B, P + T + 4, dtype=torch.int32, device=dev)
    for b in range(B):
        r2t[b, :P] = torch.arange(1, 1 + P, device=dev)
        r2t[b, P:P + T] = torch.arange(1 + P + b * T, 1 + P + (b + 1) * T, device=dev)
    seq = torch.full((B,), P + T, dtype=torch.int64, device=dev)
    ptr = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    ind = torch.zeros(B * (P + T), dtype=torch.int32, device=dev)
    ns = torch.zeros(B, dtype=torch.int32, device=dev)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, S, dev, head_dim=D)
    ops.decode_metadata_shared(r2t, torch.arange(B, device=dev), seq, ptr, ind, ns, Hq, Hkv, S, 256, plan, head_dim=D)
    ws = ops.decode_shared_workspace(B, Hq, Dv, S, dev)
    o = torch.empty(B, Hq, Dv, dtype=torch.bfloat16, device=dev)
    al = torch.empty(B, Hq, S, Dv, dtype=torch.float32, device=dev)
    lse = torch.empty(B, Hq, S, dtype=torch.float32, device=dev)
    sm = 1 / math.sqrt(192 if Dv != D else D)

    def shared():
        for k, v in pools:
            ops.decode_attention_shared_planned(q, k, v, o, ptr, ind, plan, ws, S, sm)

    def per_request():
        for k, v in pools:
            ops.decode_attention_fwd(q, k, v, o, ptr, ind, al, lse, ns, S, sm)

    out = {}
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for nm, fn in (("shared", shared), ("per_request", per_request)):
            fn()
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                fn()
            for _ in range(3):
                gr.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(10):
                gr.replay()
            b.record(st)
            st.synchronize()
            out[nm] = a.elapsed_time(b) / 10 / LAYERS * 1e3
    print(f"{name:28s} shared {out['shared']:7.2f} us   per-request {out['per_request']:7.2f} us   "
          f"x{out['per_request'] / out['shared']:.2f}", flush=True)


only = os.environ.get("ONLY", "")
if only == "mla":
    run("absorbed MLA, 16 heads (TP8 rank)", 16, 1, 576, torch.bfloat16, Dv=512)
    run("absorbed MLA, 128 heads", 128, 1, 576, torch.bfloat16, Dv=512)
    sys.exit(0)
run("Llama-3-8B heads, bf16 pool", 32, 8, 128, torch.bfloat16)
run("Llama-3-8B heads, e4m3 pool", 32, 8, 128, torch.float8_e4m3fn)
run("Llama-3-8B heads, e5m2fnuz", 32, 8, 128, torch.float8_e5m2fnuz)
run("70B TP8 rank (8 / 1), bf16", 8, 1, 128, torch.bfloat16)
run("Qwen2-0.5B heads (14 / 2 x 64)", 14, 2, 64, torch.bfloat16)
run("16 / 4 heads of 64, bf16", 16, 4, 64, torch.bfloat16)
run("absorbed MLA, 16 heads (TP8 rank)", 16, 1, 576, torch.bfloat16, Dv=512)
run("absorbed MLA, 128 heads", 128, 1, 576, torch.bfloat16, Dv=512)
