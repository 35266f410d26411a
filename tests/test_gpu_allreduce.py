"""GPU: the peer-to-peer all-reduce through the C ABI with two, four, six and eight processes sharing the one GPU of
the test box (IPC mappings between processes work the same on one device as across xGMI; what cannot be seen here is
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
sizes = (8, 4096, 40 * 1024 + 8, 1 << 20) if world <= 4 else (8, 40 * 1024 + 8, 1 << 19)
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    for numel in sizes:
        for algo in (1, 2):
            for rep in range(3 if world <= 4 else 2):
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
# the reference's graph path (custom_all_reduce.py:335-348, 388-393, 475-498): inputs nobody registered, captured inside
# capture(); leaving the context exchanges their handles (register_graph_buffers) and only then may the graph run
ins = [(torch.randn(3, 8192, generator=torch.Generator().manual_seed(500 + r)) * 4).to(torch.bfloat16) for r in range(world)]
xg = torch.empty(3, 8192, dtype=torch.bfloat16, device="cuda")
g2 = torch.cuda.CUDAGraph()
with comm.capture():
    with torch.cuda.stream(s):
        w = comm.custom_all_reduce(xg)          # warm-up before the capture: mimics the allocation only
        assert w is not None and w.shape == xg.shape
        torch.cuda.synchronize()
        with torch.cuda.graph(g2, stream=s):
            o1 = comm.custom_all_reduce(xg)
            mid = o1 * 1                        # an input that lives in the graph's own memory pool
            o2 = comm.custom_all_reduce(mid)
assert int(comm._ops.get_graph_buffer_ipc_meta(comm._ptr)[0].numel()) == 0   # both inputs are registered now
dist.barrier()
for rep in range(2):
    scale = float(rep + 1)
    xg.copy_((ins[rank] * scale).to(torch.bfloat16))
    torch.cuda.synchronize()
    dist.barrier()
    g2.replay()
    torch.cuda.synchronize()
    want1 = oracle.allreduce_sum([(t * scale).to(torch.bfloat16).reshape(-1) for t in ins], torch.bfloat16)
    want2 = oracle.allreduce_sum([want1] * world, torch.bfloat16)
    if not torch.equal(o1.cpu().reshape(-1).view(torch.uint8), want1.view(torch.uint8)):
        fails.append(("capture", rep, 1))
    if not torch.equal(o2.cpu().reshape(-1).view(torch.uint8), want2.view(torch.uint8)):
        fails.append(("capture", rep, 2))
    dist.barrier()
# an unregistered input outside a capture is an error, as in the reference (custom_all_reduce_hip.cuh:528-531)
try:
    comm.all_reduce_reg(torch.zeros(64, dtype=torch.bfloat16, device="cuda"))
    fails.append(("unregistered input accepted",))
except RuntimeError:
    pass
# the reference binding's argument checks (custom_all_reduce.hip:18-27)
try:
    comm._ops.init_custom_ar(comm.meta, comm.rank_data, [b"x" * 64] * 3, [0, 0, 0], 0, True)
    fails.append(("odd world accepted",))
except Exception as e:
    if "Odd num gpus" not in str(e):
        fails.append(("odd world", str(e)[:80]))
# all-gather of registered inputs, eager and default algorithm all-reduce of the same buffer afterwards
part = torch.full((3, 520), float(rank), dtype=torch.bfloat16, device="cuda")
comm.register_buffer(part)
for rep in range(2):
    part.fill_(float(rank + 10 * rep))
    torch.cuda.synchronize()
    dist.barrier()
    got = comm.all_gather_reg(part)
    torch.cuda.synchronize()
    want = torch.stack([torch.full((3, 520), float(r + 10 * rep)) for r in range(world)])
    if not torch.equal(got.float().cpu(), want):
        fails.append(("all_gather", rep))
    dist.barrier()
comm.check_health()
stuck = comm.stuck_blocks()
dist.barrier()
comm.close()
print(json.dumps({"rank": rank, "fails": fails[:5], "nfail": len(fails), "stuck": stuck}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4, 6, 8])
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
            o, e = p.communicate(timeout=240)
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
from sglang_security_amd.distributed import parallel_state

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
# the device group is gloo here (RCCL refuses two ranks on one GPU); it carries what the peer-to-peer kernels decline
group = parallel_state.initialize_model_parallel(world, backend="gloo", local_rank=0, max_size=4 << 20)
assert group.ca_comm is not None and not group.ca_comm.disabled, group.ca_error
fails = []
B = 6
for quant, fuse in (("bf16", False), ("fp8", False), ("awq", False), ("awq", True)):
    cfg = LlamaConfig.tiny(quant)
    dec = LlamaDecoder(cfg, batch=B, pool_slots=1 + 64 + B * 24, max_ctx=96, device="cuda", seed=0)   # same shard on every rank
    dec.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
    dec.enable_shared_prefix(True)
    if fuse:
        assert dec.enable_fusion(True)
    dec.enable_tp(group)
    tag = quant + ("+fused" if fuse else "")
    # row-parallel projection: identical shards, so the sum over ranks is exactly world x the local product
    x = torch.randn(B, dec.q_size, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).to(torch.bfloat16)
    dec._row_parallel("o", dec.layers[0]["o"], x)
    torch.cuda.synchronize()
    want = (dec.tp_buf.float() * world).to(torch.bfloat16)
    if not torch.equal(dec.hidden.view(torch.int16), want.view(torch.int16)):
        fails.append((tag, "row_parallel"))
    dist.barrier()
    # vocab-parallel sampling: every rank's logits are gathered; the highest rank holds the best one
    dec.logits.zero_()
    dec.logits[:, 7 + 3 * rank] = 1.0 + rank      # one peak per rank, at a different place and height
    best_local = 7 + 3 * (world - 1)              # the winner's
    dec._sample()
    torch.cuda.synchronize()
    if dec.next_ids.tolist() != [best_local + (world - 1) * cfg.vocab] * B:
        fails.append((tag, "sample", dec.next_ids.tolist()))
    if dec.input_ids.tolist() != [best_local] * B:
        fails.append((tag, "fold", dec.input_ids.tolist()))
    full = dec.logits_full.float().cpu()
    for r in range(world):
        if not bool((full[:, r * cfg.vocab + 7 + 3 * r] == 1.0 + r).all()) or float(full[:, r * cfg.vocab:(r + 1) * cfg.vocab].sum()) != B * (1.0 + r):
            fails.append((tag, "gathered logits", r))
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
    dec.check_tp_health()
    mine = torch.cat([dec.next_ids.float().cpu(), dec.hidden.float().cpu().flatten()])
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if not all(torch.equal(gathered[0], t) for t in gathered):
        fails.append((tag, "ranks differ"))
    if not torch.isfinite(mine).all():
        fails.append((tag, "non-finite"))
    dist.barrier()
    if fuse:
        # the fused layer path under TP against the plain op sequence under TP, from the same state
        ref = LlamaDecoder(cfg, batch=B, pool_slots=1 + 64 + B * 24, max_ctx=96, device="cuda", seed=0)
        ref.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
        ref.enable_shared_prefix(True)
        ref.enable_tp(group)
        dec.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
        for _ in range(2):
            dec.input_ids.copy_(ref.input_ids)
            ref.step()
            dec.step()
            torch.cuda.synchronize()
            a, b = ref.logits.float(), dec.logits.float()
            if float((a - b).abs().max()) > 0.05 * max(1.0, float(a.abs().max())):
                fails.append((tag, "fused vs plain", float((a - b).abs().max()), float(a.abs().max())))
        dist.barrier()
# the fallback: what the peer-to-peer kernels decline goes through the device group
odd = torch.full((3,), float(rank + 1), device="cuda")              # 12 bytes: not a multiple of 16
out = group.all_reduce(odd)
if out.tolist() != [float(sum(range(1, world + 1)))] * 3:
    fails.append(("fallback all_reduce", out.tolist()))
big = torch.full((5 << 20,), 1.0, dtype=torch.bfloat16, device="cuda")   # above this group's max_size
out = group.all_reduce(big)
if not bool((out == float(world)).all()):
    fails.append(("fallback big",))
ag = group.all_gather(torch.full((2, 3), float(rank), device="cuda"), dim=-1)
if tuple(ag.shape) != (2, 3 * world) or ag[0].tolist() != [float(r) for r in range(world) for _ in range(3)]:
    fails.append(("all_gather", ag.tolist()))
stuck = group.ca_comm.stuck_blocks()
dist.barrier()
parallel_state.destroy_model_parallel()
print(json.dumps({"rank": rank, "fails": fails[:5], "nfail": len(fails), "stuck": stuck}))
dist.destroy_process_group()
'''


def _run_workers(script, world, port, timeout):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SGLK_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


@pytest.mark.parametrize("world", [2, 6, 8])
def test_tensor_parallel_decode_step(world, tmp_path):
    """The decode step (bf16, fp8, AWQ plain and AWQ fused layer path) with the row-parallel projections summed by the
    peer-to-peer all-reduce and the vocab-parallel logits gathered from every rank, the ranks sharing the one GPU
    (decode_runner.enable_tp; llama.py:72-86,145-160), then the device-group fallback for tensors the kernels decline
    (parallel_state.py:436-461,489-540)."""
    import json

    script = tmp_path / "tp_worker.py"
    script.write_text(TP_WORKER)
    for rc, o, e in _run_workers(script, world, 29520 + world, 300):
        assert rc == 0, e[-2000:]
        rep = json.loads(o.strip().splitlines()[-1])
        assert rep["stuck"] == 0, rep
        assert rep["nfail"] == 0, rep


STUCK_WORKER = r'''
import os, sys, json, time
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["SGLK_ROOT"])
from sglang_security_amd._lib import SglkError
from sglang_security_amd.distributed.custom_all_reduce import CustomAllreduce

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
comm = CustomAllreduce(dist.group.WORLD, torch.device("cuda:0"), max_size=1 << 20, timeout_ms=300)
x = torch.ones(4096, dtype=torch.bfloat16, device="cuda")
out = comm.all_reduce_unreg(x)                       # everybody: fine
torch.cuda.synchronize()
comm.check_health()
ok_first = bool((out == float(world)).all())
dist.barrier()
rep = {"rank": rank, "ok_first": ok_first}
if rank == 0:
    t0 = time.time()
    out = comm.all_reduce_unreg(x)                   # rank 1 never comes
    torch.cuda.synchronize()
    rep["waited_s"] = round(time.time() - t0, 2)
    rep["all_nan"] = bool(torch.isnan(out.float()).all())
    try:
        comm.check_health()
        rep["raised"] = False
    except SglkError:
        rep["raised"] = True
    rep["disabled"] = comm.disabled
    rep["declines"] = comm.custom_all_reduce(x) is None
    try:
        comm.all_reduce_unreg(x)                     # a failed communicator takes no more work
        rep["rejected"] = False
    except SglkError as e:
        rep["rejected"] = "time limit" in str(e)
dist.barrier()
try:
    comm.close()
    rep["close_raised"] = False
except SglkError:
    rep["close_raised"] = True
print(json.dumps(rep))
dist.destroy_process_group()
'''


def test_missing_peer_is_loud(tmp_path):
    """A rank whose peer never reaches the barrier gets NaN instead of a sum within the time limit, check_health()
    raises, the communicator disables itself and refuses further launches."""
    import json

    script = tmp_path / "stuck_worker.py"
    script.write_text(STUCK_WORKER)
    reps = {}
    for rc, o, e in _run_workers(script, 2, 29531, 120):
        assert rc == 0, e[-2000:]
        rep = json.loads(o.strip().splitlines()[-1])
        reps[rep["rank"]] = rep
    r0 = reps[0]
    assert r0["ok_first"] and reps[1]["ok_first"]
    assert r0["all_nan"] and r0["raised"] and r0["disabled"] and r0["declines"] and r0["rejected"], r0
    assert 0.2 <= r0["waited_s"] < 20, r0
    assert not r0["close_raised"]      # already reported through check_health()
    assert not reps[1]["close_raised"]
