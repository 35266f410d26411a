"""The planned shared-prefix launch on batches of different make-up (Llama-3-8B heads, bf16 pool): everybody shares,
nobody shares, a sharing group beside lone long requests.  Microseconds per layer from graphs of 16 launches over 16
pools: the planned kernel alone, the planned launch with the per-request kernels beside it (the device picks), and the
per-request kernels."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402

dev = "cuda"
Hq, Hkv, D, S, LAYERS = 32, 8, 128, 8, 16


def run(name, rows):
    B = len(rows)
    n = 1 + max(max(r) for r in rows)
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    pools = [(torch.randn(n + 1, Hkv, D, device=dev, generator=g).to(torch.bfloat16),
              torch.randn(n + 1, Hkv, D, device=dev, generator=g).to(torch.bfloat16)) for _ in range(LAYERS)]
    lens = [len(r) for r in rows]
    r2t = torch.zeros(B, max(lens) + 4, dtype=torch.int32)
    for b, r in enumerate(rows):
        r2t[b, : len(r)] = torch.tensor(r, dtype=torch.int32)
    r2t = r2t.to(dev)
    seq = torch.tensor(lens, dtype=torch.int64, device=dev)
    ptr = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    ind = torch.zeros(sum(lens), dtype=torch.int32, device=dev)
    ns = torch.zeros(B, dtype=torch.int32, device=dev)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, S, dev)
    ops.decode_metadata_shared(r2t, torch.arange(B, device=dev), seq, ptr, ind, ns, Hq, Hkv, S, 256, plan)
    items = ops.decode_shared_plan_read(plan, B)["n_items"]
    ws = ops.decode_shared_workspace(B, Hq, D, S, dev)
    o = torch.empty_like(q)
    al = torch.empty(B, Hq, S, D, dtype=torch.float32, device=dev)
    lse = torch.empty(B, Hq, S, dtype=torch.float32, device=dev)
    sm = 1 / math.sqrt(D)
    forms = {
        "planned": lambda k, v: ops.decode_attention_shared_planned(q, k, v, o, ptr, ind, plan, ws, S, sm),
        "planned+gate": lambda k, v: ops.decode_attention_shared_planned(q, k, v, o, ptr, ind, plan, ws, S, sm, 0.0, None, ns),
        "per_request": lambda k, v: ops.decode_attention_fwd(q, k, v, o, ptr, ind, al, lse, ns, S, sm),
    }
    out, outs = {}, {}
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for nm, fn in forms.items():
            for k, v in pools:
                fn(k, v)
            st.synchronize()
            outs[nm] = o.float().clone()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for k, v in pools:
                    fn(k, v)
            for _ in range(3):
                gr.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(10):
                gr.replay()
            b.record(st)
            st.synchronize()
            out[nm] = a.elapsed_time(b) / 10 / LAYERS * 1e3
    err = max((outs[k] - outs["per_request"]).abs().max().item() for k in outs)
    print(f"{name:44s} items {items:4d}  planned {out['planned']:7.2f}  planned+gate {out['planned+gate']:7.2f}  "
          f"per-request {out['per_request']:7.2f} us   max diff {err:.4f}", flush=True)


def batch(groups, lone):
    """groups: (members, shared, private); lone: list of lengths"""
    rows, nxt = [], 1
    for members, shared, private in groups:
        base = list(range(nxt, nxt + shared))
        nxt += shared
        for _ in range(members):
            rows.append(base + list(range(nxt, nxt + private)))
            nxt += private
    for L in lone:
        rows.append(list(range(nxt, nxt + L)))
        nxt += L
    return rows


if os.environ.get("ONLY") == "groups":
    run("64 x (1024 shared + 128)", batch([(64, 1024, 128)], []))
    run("10 groups x 32 x (1024 + 128)", batch([(32, 1024, 128)] * 10, []))
    run("4 groups x 32 x (2048 + 64)", batch([(32, 2048, 64)] * 4, []))
    run("128 x (1024 + 128)", batch([(128, 1024, 128)], []))
    sys.exit(0)
run("64 x (1024 shared + 128)", batch([(64, 1024, 128)], []))
run("64 lone x 1152", batch([], [1152] * 64))
run("1 lone x 1152", batch([], [1152]))
run("8 lone x 4096", batch([], [4096] * 8))
run("64 lone x 4096", batch([], [4096] * 64))
run("32 lone x 2048", batch([], [2048] * 32))
run("60 x (1024 + 128) + 4 lone x 4096", batch([(60, 1024, 128)], [4096] * 4))
run("10 groups x 32 x (1024 + 128)", batch([(32, 1024, 128)] * 10, []))
