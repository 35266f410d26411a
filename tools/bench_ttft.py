"""Time to first token with and without a radix-cache hit, with the cached nodes public and SafeKV-private
(Llama-3-8B AWQ, 1024-token shared prefix + 128 private tokens, one request at a time).  Prints one JSON object."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder  # noqa: E402
from sglang_security_amd.prefill_runner import ttft_experiment  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prefix", type=int, default=1024)
ap.add_argument("--suffix", type=int, default=128)
ap.add_argument("--quant", default="awq")
ap.add_argument("--tiny", action="store_true")
ap.add_argument("--resident-bf16", action="store_true", help="keep a dequantised copy of the weights for the prefill GEMMs")
args = ap.parse_args()
cfg = LlamaConfig.tiny(args.quant) if args.tiny else LlamaConfig.llama3_8b(args.quant)
dec = LlamaDecoder(cfg, 1, 2, 16, device="cuda")
print(json.dumps(ttft_experiment(dec, args.prefix, args.suffix, resident_bf16=args.resident_bf16)))
