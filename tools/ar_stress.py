"""Stress of the peer-to-peer collectives with the ranks sharing one GPU: eager launches of a step-like sequence
(a matrix product, a 1 MiB two-stage all-reduce, ... then a 2 MB per rank all-gather), reporting any stuck barrier."""
import json, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sglang_security_amd.distributed.custom_all_reduce import CustomAllreduce

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
comm = CustomAllreduce(dist.group.WORLD, torch.device("cuda:0"), max_size=8 << 20, timeout_ms=int(os.environ.get("TMO", "8000")))
x = torch.ones(64, 8192, dtype=torch.bfloat16, device="cuda")
out = torch.empty_like(x)
comm.register_buffer(x)
lg = torch.ones(64, 16032, dtype=torch.bfloat16, device="cuda")
comm.register_buffer(lg)
a = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")
calls = int(os.environ.get("CALLS", "160"))
rep = {"rank": rank, "iters": []}
try:
    for it in range(int(os.environ.get("ITERS", "6"))):
        dist.barrier()
        t0 = time.time()
        for c in range(calls):
            b = a @ a
            comm.all_reduce_reg(x, out)
        g = comm.all_gather_reg(lg)
        torch.cuda.synchronize()
        rep["iters"].append(round((time.time() - t0) * 1e3, 1))
        comm.check_health()
        ok = bool((out == float(world)).all()) and bool((g == 1).all())
        if not ok:
            rep["bad"] = it
            break
except Exception as e:  # noqa: BLE001
    rep["error"] = str(e)[:200]
rep["stuck"] = comm.stuck_blocks()
print(json.dumps(rep), flush=True)
os._exit(0)
