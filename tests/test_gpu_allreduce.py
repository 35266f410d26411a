"""GPU: the peer-to-peer all-reduce through the C ABI with two and four processes sharing the one GPU of the
test box (IPC mappings between processes work the same on one device as across xGMI; what cannot be seen here is
link bandwidth).  Expected values come from the C oracle (fp32 sum in rank order, one rounding:
custom_all_reduce_hip.cuh:260-282), compared bit for bit; all ranks must hold identical results."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["SGLK_ROOT"])
import oracle
from sglang_security_amd.distributed.custom_all_reduce import CustomAllreduce

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
comm = CustomAllreduce(dist.group.WORLD, torch.device("cuda:0"), max_size=4 << 20)
assert not comm.disabled
fails = []
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    for numel in (8, 4096, 40 * 1024 + 8, 1 << 20):
        for algo in (1, 2):
            for rep in range(3):
                inputs = []
                for r in range(world):
                    g = torch.Generator().manual_seed(1000 * r + numel % 977 + rep)
                    inputs.append((torch.randn(numel, generator=g) * 4).to(dtype))
                want = oracle.allreduce_sum(inputs, dtype)
                x = inputs[rank].cuda()
                out = comm.all_reduce_unreg(x, algo=algo)
                torch.cuda.synchronize()
                if not torch.equal(out.cpu().view(torch.uint8), want.view(torch.uint8)):
                    fails.append((str(dtype), numel, algo, rep, float((out.cpu().float() - want.float()).abs().max())))
# registered input + default algorithm choice + graph capture of the launch
reg = torch.zeros(65536, dtype=torch.bfloat16, device="cuda")
comm.register_buffer(reg)
out = torch.empty_like(reg)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    comm.all_reduce_reg(reg, out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        comm.all_reduce_reg(reg, out)
for rep in range(3):
    reg.fill_(float(rank + 1 + rep))
    torch.cuda.synchronize()
    dist.barrier()
    g.replay()
    torch.cuda.synchronize()
    expect = float(sum(r + 1 + rep for r in range(world)))
    if not bool((out.float() == expect).all()):
        fails.append(("graph", rep, float(out.float().mean())))
    dist.barrier()
assert comm.custom_all_reduce(torch.zeros(3, device="cuda")) is None  # 12 bytes: not a multiple of 16
stuck = comm.stuck_blocks()
dist.barrier()
comm.close()
print(json.dumps({"rank": rank, "fails": fails[:5], "nfail": len(fails), "stuck": stuck}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_allreduce_processes_on_one_gpu(world, tmp_path):
    script = tmp_path / "ar_worker.py"
    script.write_text(WORKER)
    procs = []
    port = 29500 + world
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SGLK_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=150)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    import json

    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
        rep = json.loads(o.strip().splitlines()[-1])
        assert rep["stuck"] == 0, rep
        assert rep["nfail"] == 0, rep


TP_WORKER = r'''
import os, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["SGLK_ROOT"])
from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder
from sglang_security_amd.distributed.custom_all_reduce import CustomAllreduce

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
comm = CustomAllreduce(dist.group.WORLD, torch.device("cuda:0"), max_size=4 << 20)
assert not comm.disabled
fails = []
for quant in ("bf16", "fp8"):
    cfg = LlamaConfig.tiny(quant)
    dec = LlamaDecoder(cfg, batch=6, pool_slots=1 + 64 + 6 * 24, max_ctx=96, device="cuda", seed=0)   # same shard on every rank
    dec.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
    dec.enable_shared_prefix(True)
    dec.enable_tp(comm)
    # row-parallel projection: identical shards, so the sum over ranks is exactly world x the local product
    x = torch.randn(6, dec.q_size, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).to(torch.bfloat16)
    dec._row_parallel("o", dec.layers[0]["o"], x)
    torch.cuda.synchronize()
    want = (dec.tp_buf.float() * world).to(torch.bfloat16)
    if not torch.equal(dec.hidden.view(torch.int16), want.view(torch.int16)):
        fails.append((quant, "row_parallel"))
    dist.barrier()
    # vocab-parallel sampling: the highest rank holds the best logit
    dec.logits.zero_()
    best_local = 7 + 3 * rank                      # one peak per rank, at a different place and height
    dec.logits[:, best_local] = 1.0 + rank
    best_local = 7 + 3 * (world - 1)               # the winner's
    dec._sample()
    torch.cuda.synchronize()
    if dec.next_ids.tolist() != [best_local + (world - 1) * cfg.vocab] * 6:
        fails.append((quant, "sample", dec.next_ids.tolist()))
    if dec.input_ids.tolist() != [best_local] * 6:
        fails.append((quant, "fold", dec.input_ids.tolist()))
    dist.barrier()
    # whole steps, eager and as a graph; every rank must hold the same tokens and the same activations
    for _ in range(2):
        dec.step()
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            dec.step()
    for _ in range(2):
        dist.barrier()
        g.replay()
        torch.cuda.synchronize()
    mine = torch.cat([dec.next_ids.float().cpu(), dec.hidden.float().cpu().flatten()])
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if not all(torch.equal(gathered[0], t) for t in gathered):
        fails.append((quant, "ranks differ"))
    if not torch.isfinite(mine).all():
        fails.append((quant, "non-finite"))
    dist.barrier()
stuck = comm.stuck_blocks()
dist.barrier()
comm.close()
print(json.dumps({"rank": rank, "fails": fails[:5], "nfail": len(fails), "stuck": stuck}))
dist.destroy_process_group()
'''


def test_tensor_parallel_decode_step_two_processes(tmp_path):
    """The decode step with the row-parallel projections summed by the peer-to-peer all-reduce and the vocab-parallel
    greedy sample, two ranks sharing the one GPU (decode_runner.enable_tp; llama.py:72-86,145-160)."""
    world = 2
    script = tmp_path / "tp_worker.py"
    script.write_text(TP_WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
                   SGLK_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=200)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    import json

    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
        rep = json.loads(o.strip().splitlines()[-1])
        assert rep["stuck"] == 0, rep
        assert rep["nfail"] == 0, rep
