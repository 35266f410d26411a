"""Where the wall time of a prefix-hit request goes on the host side (development aid): cProfile over a few
serve_first_token calls with the prefix in the cache."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder  # noqa: E402
from sglang_security_amd.prefill_runner import PrefixCacheHarness  # noqa: E402

cfg = LlamaConfig.llama3_8b("awq")
dec = LlamaDecoder(cfg, 1, 2, 16, device="cuda")
h = PrefixCacheHarness(dec, 16384, 1024 + 128 + 8, None)
g = torch.Generator().manual_seed(0)
rand = lambda n: torch.randint(0, cfg.vocab, (n,), generator=g).tolist()
shared = rand(1024)
h.serve_first_token(shared + rand(128), "a")
for _ in range(3):
    h.serve_first_token(shared + rand(128), "b")
torch.cuda.synchronize()
ts = []
for _ in range(5):
    toks = shared + rand(128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.serve_first_token(toks, "b")
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("hit ms:", [round(t, 3) for t in ts])
# phases
toks = shared + rand(128)
torch.cuda.synchronize()
t0 = time.perf_counter()
prefix, node = h.tree.match_prefix(toks[:-1], user_id="b")
t1 = time.perf_counter()
print(f"match_prefix {1e3 * (t1 - t0):.3f} ms, {int(prefix.numel())} cached")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    h.serve_first_token(shared + rand(128), "b")
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
