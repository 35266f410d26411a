#!/usr/bin/env python3
"""Decode-throughput benchmark of the sglk hot path.

Workload (the bench_in_batch_prefix shape of the reference, benchmark/bench_in_batch_prefix): Llama-3-8B, AWQ int4 (G=128) weights,
bf16 activations/KV, batch 64 per GPU, every request = 1024 shared prefix tokens (same KV-pool slots, a radix
hit) + 128 private tokens, greedy decode.  One step = one full decode step of the whole batch: metadata, 32
decoder layers (norm, fused AWQ GEMMs, RoPE + KV write, paged attention, gated MLP), final norm, bf16 lm_head,
argmax, captured in one HIP graph.  Synthetic weights and KV; inputs resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched through torch.distributed.run)

Rank 0 prints ONE JSON line.  With N > 1 every rank runs its own replica of the batch (no data-path
collective; weak scaling), and value = all ranks' tokens / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prefix", type=int, default=1024)
    ap.add_argument("--suffix", type=int, default=128)
    ap.add_argument("--quant", default="awq", choices=["awq", "bf16", "fp8"])
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b-tp8-rank", "qwen2-0.5b", "deepseek-v3-dense"],
                    help="llama3-70b-tp8-rank: one rank's shard of Llama-3-70B under TP=8 (no all-reduce on one GPU); "
                         "qwen2-0.5b: the reference's CPU-runnable plumbing case, use with --quant bf16 --batch 1 "
                         "--prefix 0 --suffix 128 (head size 64: the generic attention kernels); deepseek-v3-dense: "
                         "the three dense layers of DeepSeek-V3 (absorbed MLA over a latent KV pool + block-fp8 "
                         "projections), a layer-stack workload, not a whole model")
    ap.add_argument("--tp", action="store_true",
                    help="the ranks form ONE tensor-parallel group (use with --model llama3-70b-tp8-rank --gpus 8): "
                         "peer-to-peer all-reduce after o / down, value = tokens/s of the one sharded model")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the device group: nccl (= RCCL, one rank per GPU) or gloo (rehearsal "
                         "with several ranks sharing one GPU; ranks then map to GPU rank %% device_count)")
    ap.add_argument("--attn", default="auto", choices=["auto", "per_request", "shared_prefix"])
    ap.add_argument("--tiny", action="store_true", help="2-layer toy model (plumbing check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--kv-splits", type=int, default=8, help="max_kv_splits of the attention backend")
    ap.add_argument("--no-fusion", action="store_true", help="plain op sequence per layer (no partial / block-sum hand-over)")
    ap.add_argument("--no-prefix-hit", action="store_true", help="skip the prefix-hit speedup measurements")
    return ap.parse_args()


def cpu_baseline(cfg, batch, prefix, suffix):
    """Reference CPU path (sgl-kernel pyproject_cpu: decode_attention_cpu + weight_packed_linear, bf16 -- the
    reference has no int4 CPU kernel) on ONE decoder layer of the same batch shape, on the host cores;
    scaled by the layer count.  Falls back to the scalar C oracle ("port") when the compiled reference
    library cannot run on this host."""
    import oracle
    from oracle import ref_loaders as RL

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(cores, 16)
    torch.set_num_threads(threads)
    B, Hq, Hkv, D = batch, cfg.q_heads, cfg.kv_heads, cfg.head_dim
    L = prefix + suffix + 1
    torch.manual_seed(0)
    n = 1 + prefix + B * (suffix + 1)
    kb = torch.randn(n, Hkv, D).to(torch.bfloat16)
    vb = torch.randn(n, Hkv, D).to(torch.bfloat16)
    q = torch.randn(B, Hq, D).to(torch.bfloat16)
    r2t = torch.zeros(B, L, dtype=torch.int32)
    for b in range(B):
        r2t[b, :prefix] = torch.arange(1, 1 + prefix)
        r2t[b, prefix:] = torch.arange(1 + prefix + b * (suffix + 1), 1 + prefix + (b + 1) * (suffix + 1))
    seq = torch.full((B,), L, dtype=torch.int64)
    shapes = [(cfg.hidden, (cfg.q_heads + 2 * cfg.kv_heads) * D), (cfg.q_heads * D, cfg.hidden),
              (cfg.hidden, 2 * cfg.inter), (cfg.inter, cfg.hidden)]
    path, kind = RL.ref_cpu_ops_path()
    t_layer = None
    if path is not None:
        try:
            opsr = RL.load_ref_cpu_ops()
            o = torch.zeros(B, Hq, D, dtype=torch.bfloat16)
            loc = r2t[:, L - 1].to(torch.int64)
            key, val = kb[loc].clone(), vb[loc].clone()
            logits = torch.empty(B, Hq, 8, D + 1)
            ws = [opsr.convert_weight_packed((torch.randn(N, K) * 0.02).to(torch.bfloat16)) for K, N in shapes]
            xs = [torch.randn(B, K).to(torch.bfloat16) for K, _ in shapes]

            def layer():
                opsr.decode_attention_cpu(q, kb, vb, o, key, val, loc, logits, r2t, torch.arange(B), seq,
                                          1.0 / math.sqrt(D), 0.0)
                for w, x in zip(ws, xs):
                    opsr.weight_packed_linear(x, w, None, True)

            for _ in range(2):
                layer()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                layer()
                ts.append(time.perf_counter() - t0)
            t_layer = sorted(ts)[len(ts) // 2]
            kind_s = "reference"
            sample = (f"1 of {cfg.layers} decoder layers (decode_attention_cpu + 4 weight_packed_linear bf16; the "
                      f"reference has no int4 CPU kernel), median of 5, x{cfg.layers}; lm_head excluded; "
                      f"reference build '{kind}'")
        except Exception as e:  # library present but unusable here: fall through to the port
            sys.stderr.write(f"[bench] reference CPU library unusable ({e}); using the oracle port\n")
            t_layer = None
    if t_layer is None:
        ptr = torch.zeros(B + 1, dtype=torch.int32)
        ptr[1:] = torch.cumsum(seq, 0)
        ind = torch.cat([r2t[b] for b in range(B)])
        t0 = time.perf_counter()
        oracle.decode_attention(q, kb, vb, ptr, ind, torch.full((B,), 8), 1.0 / math.sqrt(D))
        t_attn = time.perf_counter() - t0
        K, N = shapes[1]
        x = torch.randn(B, K).to(torch.bfloat16)
        w = (torch.randn(K, N) * 0.02).to(torch.bfloat16)
        t0 = time.perf_counter()
        oracle.half_matmul(x, w)
        t_mm = time.perf_counter() - t0
        params = sum(k * n for k, n in shapes)
        t_layer = t_attn + t_mm * params / (K * N)
        kind_s = "port"
        sample = (f"C oracle: decode attention of 1 layer + one {K}x{N} bf16 matmul scaled to the layer's "
                  f"{params / 1e6:.0f}M weights, x{cfg.layers}; lm_head excluded")
    return {"value": round(batch / (t_layer * cfg.layers), 2), "unit": "tokens/s", "cores": threads, "kind": kind_s,
            "sample": sample}


def measure_decode(dec, args, dev, world, share_prefix, attn):
    """Set the batch up, capture one decode step into a graph, time K replays.  Returns (elapsed, graph_used, mode)."""
    from sglang_security_amd.benchlib import dist_helpers, timed_steps

    dec.setup_shared_prefix_batch(args.prefix, args.suffix, share=share_prefix)
    mode = "per_request"
    dec.enable_shared_prefix(False)
    from sglang_security_amd import ops

    c = dec.cfg
    pays = ops.shared_prefix_pays(c.kv_heads, c.head_dim, c.head_dim, dec.k_pool[0].element_size())
    if share_prefix and (attn == "shared_prefix" or (attn == "auto" and pays)) and dec.enable_shared_prefix(True):
        mode = "shared_prefix"
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(dev)
    barrier, mx = dist_helpers(world, dev)
    with torch.cuda.stream(stream):
        n_eager = min(2, args.warmup)
        for _ in range(n_eager):
            dec.step()
        stream.synchronize()
        graph = None
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                dec.step()
        run = graph.replay if graph is not None else dec.step
        elapsed = timed_steps(run, args.steps, args.warmup - n_eager, torch.cuda.synchronize, barrier, mx)
    return elapsed, graph is not None, mode, stream


def tp_report(dec, group, args, dev, world, rank, stream, step_ms):
    """What the tensor-parallel run actually did: the ranks that took part (rank, device, process), which path each
    all-reduce call took (peer-to-peer kernel on a registered / unregistered input, or the group's RCCL collective),
    and the all-reduce's share of the step, measured twice: events around every row-parallel sum of eager steps, and
    the replayed step against the same graph captured with the sums replaced by local copies."""
    import socket

    import torch.distributed as dist
    from sglang_security_amd.benchlib import dist_helpers, timed_steps

    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "device": torch.cuda.current_device(), "pid": os.getpid(),
                                  "host": socket.gethostname()})
    barrier, mx = dist_helpers(world, dev)
    pairs, orig = [], dec._tp_sum

    def timed_sum():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        orig()
        b.record(stream)
        pairs.append((a, b))

    before = dict(group.ar_calls)
    eager = []
    with torch.cuda.stream(stream):
        dec._tp_sum = timed_sum
        for _ in range(3):
            pairs.clear()
            barrier()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record(stream)
            dec.step()
            s1.record(stream)
            stream.synchronize()
            dec.check_tp_health()
            eager.append((sum(a.elapsed_time(b) for a, b in pairs), s0.elapsed_time(s1), len(pairs)))
        dec._tp_sum = orig
        calls = {k: (group.ar_calls[k] - before[k]) // 3 for k in before}
        ar_ms, step_eager_ms, n_calls = sorted(eager)[1]
        # the same step with every row-parallel sum replaced by a copy of this rank's product
        local_ms = None
        if not args.no_graph:
            dec._tp_sum = lambda: dec.hidden.copy_(dec.tp_buf)
            dec.step()
            stream.synchronize()
            dec.check_tp_health()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                dec.step()
            dec._tp_sum = orig
            local_ms = timed_steps(g.replay, args.steps, 2, torch.cuda.synchronize, barrier, mx) / args.steps * 1e3
    nbytes = dec.tp_buf.numel() * dec.tp_buf.element_size()
    out = {"ranks_seen": sorted(seen, key=lambda r: r["rank"]), "devices": len({(r["host"], r["device"]) for r in seen}),
           "allreduce_calls_per_step": n_calls, "allreduce_bytes_per_call": nbytes, "allreduce_path_per_step": calls,
           "eager_events": {"allreduce_ms_per_step": round(ar_ms, 4), "step_ms": round(step_eager_ms, 4),
                            "allreduce_share": round(ar_ms / step_eager_ms, 4),
                            "allreduce_us_per_call": round(ar_ms * 1e3 / max(n_calls, 1), 2)}}
    if local_ms is not None:
        out["graph_replay"] = {"step_ms": round(step_ms, 4), "step_ms_sums_replaced_by_copies": round(local_ms, 4),
                               "allreduce_share": round(max(0.0, 1.0 - local_ms / step_ms), 4)}
    return out


def gemm_roofline(dec, cfg, B, stream):
    """The dominant kernel (fused AWQ GEMM), timed with events on the launch stream: each projection of all layers
    is launched back to back between one event pair (the device is parked first so that the host has enqueued the
    whole sweep before the first launch runs; every launch reads another layer's weights, so they come from HBM)."""
    per = {}
    with torch.cuda.stream(stream):
        for name in ("qkv", "o", "gate_up", "down"):
            ts = []
            for _ in range(3):
                torch.cuda._sleep(int(20e-3 * 2.0e9))
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                dec.gemm_sweep(name)
                b.record(stream)
                stream.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / cfg.layers)
            per[name] = sorted(ts)[1]
    avg_us = sum(per.values()) / len(per)
    # algorithmic bytes per launch: one layer's weights (+ scales and zeros) + the 16-bit input and output activations
    # of its four projections, / 4 launches
    shapes = ((cfg.hidden, dec.q_size + 2 * dec.kv_size), (dec.q_size, cfg.hidden), (cfg.hidden, 2 * cfg.inter),
              (cfg.inter, cfg.hidden))
    act = 2 * B * sum(k + n for k, n in shapes)
    bytes_per_launch = (dec.layer_weight_bytes() + act) / 4.0
    ach = bytes_per_launch / (avg_us * 1e-6) / 1e9
    kernel = {"awq": f"awq_gemm_wide_kernel (fused int4 dequant + MFMA GEMM, M={B}; 4 launches per layer)",
              "fp8": f"fp8_scaled_mm kernel (e4m3 x e4m3 MFMA GEMM with per-token and per-channel scales, M={B}; 4 per layer)",
              "bf16": f"dense_gemm_nt kernel (bf16 MFMA GEMM, M={B}; 4 per layer)"}[cfg.quant]
    return {"bound": "hbm", "kernel": kernel,
            "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
            "traffic": pmc_traffic(cfg, B),
            "avg_launch_us": round(avg_us, 2), "bytes_per_launch": int(bytes_per_launch),
            "per_projection_us": {k: round(v, 2) for k, v in per.items()}}


def pmc_traffic(cfg, B):
    """HBM bytes per launch of the dominant kernel from the hardware counters.  Counters cannot be read inside this
    process, so the figure comes from a committed collection (profiles/*_pmc_traffic.json) and is reported only when
    that file says it was collected on the kernel source this tree holds (sha256 of awq.hip) for this workload;
    otherwise null."""
    import glob
    import hashlib

    if cfg.quant != "awq":
        return None
    try:
        src = os.path.join(ROOT, "sglang_security_amd", "csrc", "awq.hip")
        digest = hashlib.sha256(open(src, "rb").read()).hexdigest()
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            rec = json.load(open(path))
            if rec.get("awq_hip_sha256") == digest and rec.get("hidden") == cfg.hidden and rec.get("batch") == B:
                return rec.get("awq_gemm_hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def deepseek_dense_workload(args, dev, world):
    """BASELINE configuration 5's layer type: DeepSeek-V3 dense layers (absorbed MLA + block fp8) on a shared-prefix
    batch.  One step = the three layers on `batch` new tokens; the line says so (this is not a whole-model rate)."""
    from sglang_security_amd import ops
    from sglang_security_amd.benchlib import aggregate, dist_helpers, timed_steps
    from sglang_security_amd.mla_runner import DeepseekV3Config, DeepseekV3DenseDecoder

    cfg = DeepseekV3Config.tiny() if args.tiny else DeepseekV3Config.deepseek_v3()
    B = args.batch
    per_req = args.suffix + args.steps + args.warmup + 10
    dec = DeepseekV3DenseDecoder(cfg, B, 1 + args.prefix + B * per_req, args.prefix + per_req, device=dev,
                                 max_kv_splits=args.kv_splits)
    dec.cu_count = torch.cuda.get_device_properties(dev).multi_processor_count
    dec.setup_shared_prefix_batch(args.prefix, args.suffix)
    x_in = dec.hidden.clone()
    stream = torch.cuda.Stream(dev)
    barrier, mx = dist_helpers(world, dev)
    with torch.cuda.stream(stream):
        for _ in range(2):
            dec.step(x_in)
        stream.synchronize()
        graph = None
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                dec.step(x_in)
        run = graph.replay if graph is not None else (lambda: dec.step(x_in))
        elapsed = timed_steps(run, args.steps, max(args.warmup - 2, 0), torch.cuda.synchronize, barrier, mx)
        # dominant kernel: the block-fp8 GEMM, every projection of every layer launched back to back between events
        per = {}
        for name, K in (("qkv_a", cfg.hidden), ("q_b", cfg.q_lora_rank), ("o", cfg.heads * cfg.v_head_dim),
                        ("gate_up", cfg.hidden), ("down", cfg.inter)):
            xq, xs = dec._quant(torch.randn(B, K, device=dev).to(torch.bfloat16))
            ts = []
            for _ in range(3):
                torch.cuda._sleep(int(5e-3 * 2.0e9))
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for L in dec.layers:  # the product the step runs (streaming layout where the layer holds one)
                    w = L[name]
                    if w.wpacked is not None:
                        ops.fp8_blockwise_scaled_mm_packed(xq, w.wpacked, xs, w.scale.t(), torch.bfloat16, dec.workspace)
                    else:
                        ops.fp8_blockwise_scaled_mm(xq, w.weight.t(), xs, w.scale.t(), torch.bfloat16, dec.workspace)
                b.record(stream)
                stream.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / cfg.layers)
            per[name] = sorted(ts)[1]
    agg = aggregate(B, args.steps, world, elapsed)
    L0 = dec.layers[0]
    gemm_bytes = sum(L0[k].bytes() + B * (L0[k].K + 2 * L0[k].N) for k in per)   # weights + fp8 in + bf16 out
    avg_us = sum(per.values()) / len(per)
    ach = gemm_bytes / len(per) / (avg_us * 1e-6) / 1e9
    kv_tok = dec.lat * 2 * cfg.layers
    uniq = dec.layer_weight_bytes() * cfg.layers + (args.prefix + B * (args.suffix + args.warmup + args.steps // 2)) * kv_tok
    return {
        "metric": f"decode tokens/s through DeepSeek-V3's {cfg.layers} dense layers (absorbed MLA + block fp8), batch {B}/GPU, "
                  f"{args.prefix}-token shared prefix",
        "value": round(agg["value"], 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(agg["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp8", "data": "synthetic",
        "config": {"workload": f"DeepSeek-V3 dense decoder layers x{cfg.layers} (hidden {cfg.hidden}, {cfg.heads} heads, latent "
                               f"KV 512 + 64, MLP {cfg.inter}; e4m3 weights with 128 x 128 block scales, per-group e4m3 "
                               f"activations, bf16 latent KV), batch {B}/GPU, prefix {args.prefix} shared + {args.suffix} "
                               "private tokens, random-init weights; a layer stack, not a whole model",
                   "attention": "per_request (absorbed MLA, 576 / 512)", "graph": graph is not None,
                   "parallelism": f"replicas x{world}", "layers": cfg.layers},
        "step_bytes_unique": int(uniq), "step_hbm_frac_of_8TBps": round(uniq / (elapsed / args.steps) / 8e12, 4),
        "roofline": {"bound": "hbm", "kernel": f"fp8 block-scaled GEMM (M={B}; 5 launches per layer)", "achieved": round(ach, 1),
                     "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                     "avg_launch_us": round(avg_us, 2), "bytes_per_launch": int(gemm_bytes / len(per)),
                     "per_projection_us": {k: round(v, 2) for k, v in per.items()}},
    }


def extend_ttft(cfg, B, prefix, suffix, dev):
    """Prefix-hit speedup of the extend (prefill) attention of one layer: all prefix + suffix tokens new, against
    only the suffix new over a cached prefix."""
    from sglang_security_amd import ops

    Hq, Hkv, D = cfg.q_heads, cfg.kv_heads, cfg.head_dim
    g = torch.Generator(device=dev).manual_seed(0)
    dt = torch.bfloat16
    kb = torch.randn(1 + prefix, Hkv, D, device=dev, generator=g).to(dt)
    vb = torch.randn(1 + prefix, Hkv, D, device=dev, generator=g).to(dt)

    def run(ext, pre):
        T = B * ext
        q = torch.randn(T, Hq, D, device=dev, generator=g).to(dt)
        k = torch.randn(T, Hkv, D, device=dev, generator=g).to(dt)
        v = torch.randn(T, Hkv, D, device=dev, generator=g).to(dt)
        o = torch.empty_like(q)
        qo = (torch.arange(B + 1, device=dev) * ext).int()
        kp = (torch.arange(B + 1, device=dev) * pre).int()
        ki = (torch.arange(pre, device=dev, dtype=torch.int32) + 1).repeat(B) if pre else torch.zeros(1, dtype=torch.int32, device=dev)
        f = lambda: ops.extend_attention_fwd(q, k, v, o, kb, vb, qo, kp, ki, None, True, None, ext, 1.0 / math.sqrt(D))
        for _ in range(2):
            f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 5

    miss = run(prefix + suffix, 0)
    hit = run(suffix, prefix)
    return {"extend_attention_ms_no_hit": round(miss, 3), "extend_attention_ms_hit": round(hit, 3),
            "extend_attention_speedup": round(miss / hit, 2)}


def in_batch_prefix_attention(cfg, dev, n_prefix=10, per_prefix=32, prefix=1024, suffix=128, layers=8):
    """Decode attention of the reference's in-batch-prefix workload (benchmark/bench_in_batch_prefix/
    bench_in_batch_prefix.py:112-116: 10 prefixes x 32 requests, 1024 + 128 tokens), the requests of the ten groups
    interleaved in the batch, through the attention backend exactly as python/sglang drives it under graphs:
    init_cuda_graph_state, capture, init_forward_metadata_replay_cuda_graph, replay.  The backend is told nothing about
    the groups; with prefix sharing on it finds them on the device.  Time per layer launch, sharing on against off."""
    import types

    from sglang_security_amd.attention import SglkAttnBackend
    from sglang_security_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool, TokenToKVPoolAllocator

    B = n_prefix * per_prefix
    Hq, Hkv, D = cfg.q_heads, cfg.kv_heads, cfg.head_dim
    slots = 1 + n_prefix * prefix + B * (suffix + 1) + 64
    dt = torch.bfloat16
    r2t = ReqToTokenPool(B + 1, prefix + suffix + 8, dev)
    kv = MHATokenToKVPool(slots, 1, dt, Hkv, D, layers, dev)
    alloc = TokenToKVPoolAllocator(slots, dt, dev, kv)
    g = torch.Generator(device=dev).manual_seed(0)
    for l in range(layers):
        kv.k_buffer[l].normal_(generator=g)
        kv.v_buffer[l].normal_(generator=g)
    prefixes = [alloc.alloc(prefix) for _ in range(n_prefix)]
    L = prefix + suffix
    for b in range(B):
        r2t.req_to_token[b, :prefix] = prefixes[b % n_prefix].int()
        r2t.req_to_token[b, prefix:L] = alloc.alloc(suffix).int()
    mcfg = types.SimpleNamespace(num_attention_heads=Hq, head_dim=D, get_num_kv_heads=lambda tp: Hkv)
    mr = types.SimpleNamespace(device=dev, req_to_token_pool=r2t, token_to_kv_pool=kv, model_config=mcfg, tp_size=1,
                               server_args=types.SimpleNamespace(triton_attention_num_kv_splits=8))
    rpi = torch.arange(B, device=dev, dtype=torch.int32)
    lens = torch.full((B,), L, device=dev, dtype=torch.int32)
    q = torch.randn(B, Hq * D, device=dev, generator=g).to(dt)
    mode = types.SimpleNamespace(is_decode_or_idle=lambda: True)
    fb = types.SimpleNamespace(forward_mode=mode, batch_size=B, req_pool_indices=rpi, seq_lens=lens, seq_lens_sum=B * L,
                               out_cache_loc=None, token_to_kv_pool=kv)
    lyr = [types.SimpleNamespace(layer_id=i, tp_q_head_num=Hq, tp_k_head_num=Hkv, tp_v_head_num=Hkv, qk_head_dim=D,
                                 v_head_dim=D, scaling=1.0 / math.sqrt(D), logit_cap=0.0) for i in range(layers)]
    out = {}
    res = {}
    for name, share in (("shared", True), ("per_request", False)):
        be = SglkAttnBackend(mr, share_prefix=share)
        be.init_cuda_graph_state(B)
        be.init_forward_metadata_capture_cuda_graph(B, B, rpi, lens)
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            for ly in lyr[:2]:
                be.forward(q, None, None, ly, fb, save_kv_cache=False)
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for ly in lyr:
                    res[name] = be.forward(q, None, None, ly, fb, save_kv_cache=False)
            ts = []
            for _ in range(5):
                be.init_forward_metadata_replay_cuda_graph(B, rpi, lens, B * L)
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                gr.replay()
                b_.record(st)
                st.synchronize()
                ts.append(a.elapsed_time(b_) * 1e3 / layers)
        out[f"attention_us_{name}"] = round(sorted(ts)[len(ts) // 2], 2)
    out["attention_speedup"] = round(out["attention_us_per_request"] / out["attention_us_shared"], 2)
    out["max_abs_diff"] = round(float((res["shared"].float() - res["per_request"].float()).abs().max()), 5)
    out["shape"] = f"{n_prefix} prefixes x {per_prefix} requests, {prefix} + {suffix} tokens, groups interleaved, graph replay"
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    if args.dist_backend == "gloo":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from sglang_security_amd.benchlib import aggregate
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    if args.model == "deepseek-v3-dense":
        out = deepseek_dense_workload(args, dev, world)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            import torch.distributed as dist

            dist.destroy_process_group()
        return
    if args.tiny:
        cfg = LlamaConfig.tiny(args.quant)
    elif args.model == "llama3-70b-tp8-rank":
        cfg = LlamaConfig.llama3_70b_tp8_rank(args.quant)
    elif args.model == "qwen2-0.5b":
        cfg = LlamaConfig.qwen2_0_5b(args.quant)
    else:
        cfg = LlamaConfig.llama3_8b(args.quant)
    B = args.batch
    total_steps = args.steps + args.warmup + 8
    per_req = args.suffix + total_steps + 2
    want_hit = not args.no_prefix_hit and world == 1 and args.prefix > 0
    # pool sized for the variant without sharing too (every request holds its own copy of the prefix)
    slots = 1 + (B * (args.prefix + per_req) if want_hit else args.prefix + B * per_req)
    max_ctx = args.prefix + per_req
    dec = LlamaDecoder(cfg, B, slots, max_ctx, device=dev, max_kv_splits=args.kv_splits)
    dec.cu_count = torch.cuda.get_device_properties(dev).multi_processor_count
    tp = args.tp and world > 1
    fused = (not args.no_fusion) and dec.enable_fusion(True)
    collective = None
    if tp:
        from sglang_security_amd.distributed import parallel_state

        # peer-to-peer kernels where they apply (2/4/6/8 ranks, < 16 MiB), the device group's collectives (RCCL under
        # "nccl") everywhere else
        group = parallel_state.initialize_model_parallel(world, backend=args.dist_backend, local_rank=local)
        dec.enable_tp(group)
        p2p = group.ca_comm is not None and not group.ca_comm.disabled
        collective = "peer-to-peer kernels (RCCL for what they decline)" if p2p else f"{args.dist_backend} collectives"
        if not p2p and args.dist_backend != "nccl":
            args.no_graph = True  # host-staged collectives cannot be captured

    elapsed, graphed, attn_mode, stream = measure_decode(dec, args, dev, world, True, args.attn)
    dec.check_tp_health()
    agg = aggregate(B, args.steps, 1 if tp else world, elapsed)
    roof = gemm_roofline(dec, cfg, B, stream)

    wname = {"awq": "AWQ int4", "bf16": "bf16", "fp8": "fp8 w8a8"}[cfg.quant]
    wdesc = {"awq": "awq (int4 weights, G=128; bf16 activations and KV)",
             "bf16": "bf16 (bf16 weights, activations and KV)",
             "fp8": "fp8 (e4m3 weights with per-channel scales, per-token e4m3 activations; bf16 KV)"}[cfg.quant]
    mname = ("Llama-3-8B" if args.model == "llama3-8b" else "Qwen2-0.5B" if args.model == "qwen2-0.5b"
             else f"Llama-3-70B TP={world}" if tp else "Llama-3-70B TP=8 rank shard (no all-reduce)")
    out = {
        "metric": f"decode tokens/s (whole job), {mname} {wname}, batch {B}/GPU, {args.prefix}-token shared prefix",
        "value": round(agg["value"], 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(agg["ms_per_step"], 4), "higher_is_better": True, "scaling": "strong" if tp else "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{mname} {wdesc} decode, batch {B}/GPU, "
                               f"prefix {args.prefix} shared + {args.suffix} private tokens, seq_len "
                               f"{args.prefix + args.suffix}+step, greedy, random-init weights",
                   "attention": attn_mode, "layer_fusion": bool(fused), "graph": graphed, "parallelism": f"tp{world}" if tp else f"replicas x{world}", "layers": cfg.layers},
    }
    if collective:
        out["config"]["collective"] = collective
        out["tensor_parallel"] = tp_report(dec, group, args, dev, world, rank, stream, agg["ms_per_step"])
        dec.check_tp_health()
    # fraction of the HBM roofline for the whole step (unique bytes: weights + lm_head + KV with the prefix once)
    if not args.tiny:
        kv_tok = 2 * cfg.kv_heads * cfg.head_dim * 2 * cfg.layers
        w_bytes = dec.layer_weight_bytes() * cfg.layers + dec.lm_head.bytes()
        uniq = w_bytes + (args.prefix + B * (args.suffix + args.warmup + args.steps // 2)) * kv_tok
        out["step_bytes_unique"] = int(uniq)
        out["step_hbm_frac_of_8TBps"] = round(uniq / (elapsed / args.steps) / 8e12, 4)
    if roof:
        out["roofline"] = roof
    if want_hit and not args.tiny:
        # the same batch with the prefix NOT shared in the pool (every request walks its own 1152 slots)
        e2, _, _, _ = measure_decode(dec, args, dev, world, False, "per_request")
        hit = {"decode_ms_no_hit": round(e2 / args.steps * 1e3, 4), "decode_ms_hit": round(elapsed / args.steps * 1e3, 4),
               "decode_speedup": round(e2 / elapsed, 3)}
        try:
            hit.update(extend_ttft(cfg, B, args.prefix, args.suffix, dev))
        except Exception as e:  # the headline number must survive a failure here
            hit["extend_error"] = str(e)[:200]
        try:
            hit["in_batch_prefix"] = in_batch_prefix_attention(cfg, dev)
        except Exception as e:
            hit["in_batch_prefix_error"] = str(e)[:200]
        if cfg.quant in ("awq", "bf16"):
            # whole-model time to first token through the radix cache: cold, prefix cached by another user (public
            # nodes), and the same with SafeKV holding the nodes private (the fork's timing side-channel experiment)
            try:
                from sglang_security_amd.prefill_runner import ttft_experiment

                hit["ttft"] = ttft_experiment(dec, args.prefix, args.suffix)
            except Exception as e:
                hit["ttft_error"] = str(e)[:200]
        out["prefix_hit"] = hit
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.tiny:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, B, args.prefix, args.suffix)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
