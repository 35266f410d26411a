import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle
from sglang_security_amd import ops
from test_gpu_ops import _awq_case
DEV="cuda"
M,K,N=64,4096,6144
x, qw, sc, qz = _awq_case(M, K, N, 128, torch.bfloat16, seed=N)
want = oracle.awq_gemm(x, qw, sc, qz).float()
packed = ops.awq_repack(qw.to(DEV)); scd, qzd = sc.to(DEV), qz.to(DEV)
os.environ["SGLK_AWQ_WIDE"]="2"
ys=[ops.awq_gemm(x.to(DEV), packed, scd, qzd).float().cpu() for _ in range(4)]
print("sps", os.environ.get("SGLK_AWQ_WIDE_SPS"), "err vs oracle", [round((y-want).abs().max().item(),3) for y in ys])
d=(ys[0]-want).abs()
idx=(d>0.1).nonzero()
print("  bad count", idx.shape[0], "rows", sorted(set(idx[:,0].tolist()))[:20], "cols", sorted(set((idx[:,1]//16).tolist()))[:40])
