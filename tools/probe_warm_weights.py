"""Does a projection run faster when its weights were read shortly before (memory-side cache) than from HBM?

Times each projection of the decode step two ways with events on one stream, launches queued behind a parked device:
`cold` = the projection of every layer in turn (each launch reads other weights), `warm` = one layer's projection as
many times (the weights were read by the previous launch: the best a prefetch by an earlier kernel of the step could
leave behind).
"""
import argparse
import json

import torch

from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder


def timed(stream, fn, n):
    ts = []
    for _ in range(3):
        torch.cuda._sleep(int(20e-3 * 2.0e9))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        fn()
        b.record(stream)
        stream.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / n)
    return sorted(ts)[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quant", default="awq")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = LlamaConfig.llama3_8b(args.quant)
    B = args.batch
    dec = LlamaDecoder(cfg, B, 1 + 1024 + B * 256, 1024 + 256, device=dev)
    dec.cu_count = torch.cuda.get_device_properties(dev).multi_processor_count
    dec.enable_fusion(True)
    dec.setup_shared_prefix_batch(1024, 128, share=True)
    dec.enable_shared_prefix(True)
    stream = torch.cuda.Stream(dev)
    out = {}
    with torch.cuda.stream(stream):
        dec.step()
        stream.synchronize()
        layers = dec.layers
        for name in ("qkv", "o", "gate_up", "down"):
            cold = timed(stream, lambda: dec.gemm_sweep(name), len(layers))
            dec.layers = [layers[3]] * len(layers)
            warm = timed(stream, lambda: dec.gemm_sweep(name), len(layers))
            dec.layers = layers
            out[name] = {"cold_us": round(cold, 2), "warm_us": round(warm, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
