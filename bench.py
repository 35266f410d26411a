#!/usr/bin/env python3
"""Decode-throughput benchmark of the sglk hot path.

Workload (BASELINE.json configs[2] on the bench_in_batch_prefix shape): Llama-3-8B, AWQ int4 (G=128) weights,
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
    ap.add_argument("--quant", default="awq", choices=["awq", "bf16"])
    ap.add_argument("--attn", default="auto", choices=["auto", "per_request", "shared_prefix"])
    ap.add_argument("--tiny", action="store_true", help="2-layer toy model (plumbing check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    from sglang_security_amd import ops
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = LlamaConfig.tiny(args.quant) if args.tiny else LlamaConfig.llama3_8b(args.quant)
    B = args.batch
    total_steps = args.steps + args.warmup + 8
    per_req = args.suffix + total_steps + 2
    slots = 1 + args.prefix + B * per_req
    max_ctx = args.prefix + per_req
    dec = LlamaDecoder(cfg, B, slots, max_ctx, device=dev)
    dec.cu_count = torch.cuda.get_device_properties(dev).multi_processor_count
    dec.setup_shared_prefix_batch(args.prefix, args.suffix)
    attn_mode = "per_request"
    if args.attn in ("auto", "shared_prefix") and hasattr(dec, "enable_shared_prefix"):
        if dec.enable_shared_prefix():
            attn_mode = "shared_prefix"
    torch.cuda.synchronize()

    # ---- warm-up (eager), then capture one step into a graph ------------------------------------------
    stream = torch.cuda.Stream(dev)
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
        for _ in range(args.warmup - n_eager):
            run()
        stream.synchronize()

        def barrier():
            if world > 1:
                import torch.distributed as dist

                dist.barrier()

        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist

            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- dominant kernel: the fused AWQ GEMM, timed per launch with events on the launch stream -------
        roof = None
        if cfg.quant == "awq":
            evs = []

            def timer(name, fn):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                fn()
                b.record(stream)
                evs.append((name, a, b))

            dec.gemm_timer = timer
            for _ in range(2):
                dec.step()
            stream.synchronize()
            dec.gemm_timer = None
            ms = [a.elapsed_time(b) for _, a, b in evs]
            avg_s = sum(ms) / len(ms) * 1e-3
            # algorithmic bytes per launch: one layer's int4 weights + scales + zeros + bf16 in/out
            # activations of its four projections, / 4 launches
            act = 2 * B * sum(k + n for k, n in ((cfg.hidden, dec.q_size + 2 * dec.kv_size), (dec.q_size, cfg.hidden),
                                                 (cfg.hidden, 2 * cfg.inter), (cfg.inter, cfg.hidden)))
            bytes_per_launch = (dec.layer_weight_bytes() + act) / 4.0
            ach = bytes_per_launch / avg_s / 1e9
            roof = {"bound": "hbm", "kernel": "awq_gemm_kernel (+xsum, split-K reduce)", "achieved": round(ach, 1),
                    "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                    "avg_launch_us": round(avg_s * 1e6, 2), "bytes_per_launch": int(bytes_per_launch)}

    tokens = B * args.steps * world
    value = tokens / elapsed
    out = {
        "metric": "decode tokens/s (whole job), Llama-3-8B AWQ int4, batch 64/GPU, 1024-token shared prefix",
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16 activations x int4 weights (fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"Llama-3-8B {cfg.quant} decode, batch {B}/GPU, prefix {args.prefix} shared + "
                               f"{args.suffix} private tokens, seq_len {args.prefix + args.suffix}+step, greedy",
                   "attention": attn_mode, "graph": graph is not None, "parallelism": f"replicas x{world}",
                   "layers": cfg.layers},
    }
    # fraction of the HBM roofline for the whole step (unique bytes: weights + lm_head + KV with the prefix once)
    if not args.tiny:
        kv_tok = 2 * cfg.kv_heads * cfg.head_dim * 2 * cfg.layers
        w_bytes = dec.layer_weight_bytes() * cfg.layers + dec.lm_head.bytes()
        uniq = w_bytes + (args.prefix + B * (args.suffix + args.warmup + args.steps // 2)) * kv_tok
        out["step_bytes_unique"] = int(uniq)
        out["step_hbm_frac_of_8TBps"] = round(uniq / (elapsed / args.steps) / 8e12, 4)
    if roof:
        out["roofline"] = roof
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
