"""Probe: the fused add + RMSNorm launch and the qkv / gate_up GEMM that follows it, as two dependent launches (what the
step does) against the same two launches on two branches of a graph with no dependency between them (the upper bound of
what an early-launched GEMM that prefetches its weights under the norm could gain; results of the concurrent form are
meaningless, only its time is read)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd import ops  # noqa: E402
from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder  # noqa: E402

dev = "cuda"
cfg = LlamaConfig.llama3_8b("awq")
B = 64
dec = LlamaDecoder(cfg, B, 1 + 1024 + B * 160, 1200, device=dev)
dec.setup_shared_prefix_batch(1024, 128)
dec.enable_shared_prefix(True)
dec.enable_fusion(True)
for _ in range(2):
    dec.step()
torch.cuda.synchronize()
P, S = dec.partial, dec.part_splits
c = cfg


def pair(L, which, side):
    """norm (consumes the partials of the projection before it) then the GEMM; side: stream for the GEMM or None"""
    if which == "qkv":
        ops.fused_add_rmsnorm_ex(dec.hidden, dec.residual, L["ln1"], c.eps, partials=P, splits=S["down"], xsum_out=dec.xs_hidden)
        w = L["qkv"]
        run = lambda: ops.awq_gemm_partial(dec.hidden, w.packed, w.scales, w.qzeros, P, dec.xs_hidden)
    else:
        ops.fused_add_rmsnorm_ex(dec.hidden, dec.residual, L["ln2"], c.eps, partials=P, splits=S["o"], xsum_out=dec.xs_hidden,
                                 zero_out=dec.xs_act)
        il = L["gate_up_il"]
        run = lambda: ops.awq_gemm_silu_mul(dec.hidden, il[0], il[1], il[2], dec.act, dec.xs_hidden, dec.xs_act)
    run()


def build(which, concurrent):
    main = torch.cuda.Stream()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            for L in dec.layers:
                if not concurrent:
                    pair(L, which, None)
                    continue
                # fork: the GEMM on the side stream, the norm on the main one, nothing between them
                side.wait_stream(main)
                if which == "qkv":
                    ops.fused_add_rmsnorm_ex(dec.hidden, dec.residual, L["ln1"], c.eps, partials=P, splits=S["down"],
                                             xsum_out=dec.xs_hidden)
                    with torch.cuda.stream(side):
                        w = L["qkv"]
                        ops.awq_gemm_partial(dec.hidden2, w.packed, w.scales, w.qzeros, dec.partial2, dec.xs2)
                else:
                    ops.fused_add_rmsnorm_ex(dec.hidden, dec.residual, L["ln2"], c.eps, partials=P, splits=S["o"],
                                             xsum_out=dec.xs_hidden, zero_out=dec.xs_act)
                    with torch.cuda.stream(side):
                        il = L["gate_up_il"]
                        ops.awq_gemm_silu_mul(dec.hidden2, il[0], il[1], il[2], dec.act2, dec.xs2, dec.xs_act2)
                main.wait_stream(side)
    return g, main


dec.hidden2 = dec.hidden.clone()
dec.partial2 = torch.empty_like(P)
dec.xs2 = dec.xs_hidden.clone()
dec.act2 = dec.act.clone()
dec.xs_act2 = dec.xs_act.clone()
for which in ("qkv", "gate_up"):
    res = {}
    for concurrent in (False, True, False, True):
        g, st = build(which, concurrent)
        with torch.cuda.stream(st):
            for _ in range(3):
                g.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(10):
                g.replay()
            b.record(st)
            st.synchronize()
        res.setdefault(concurrent, []).append(a.elapsed_time(b) / 10 / cfg.layers * 1e3)
    print(f"{which:8s} norm -> GEMM dependent: {min(res[False]):6.2f} us per layer   two branches: {min(res[True]):6.2f} us", flush=True)
