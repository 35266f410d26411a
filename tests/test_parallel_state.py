"""CPU, world_size 2 over gloo: the collective dispatch of distributed.parallel_state.GroupCoordinator when no
peer-to-peer communicator exists -- every call goes through the device group, as the reference does when its custom
all-reduce declines (parallel_state.py:436-461; all_gather layout :520-540)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["SGLK_ROOT"])
import torch, torch.distributed as dist
from sglang_security_amd.distributed import parallel_state as ps
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = ps.initialize_model_parallel(world, backend="gloo", local_rank=0)
rep = {"rank": rank, "p2p": g.ca_comm is not None, "tp_rank": ps.get_tensor_model_parallel_rank(),
       "tp_world": ps.get_tensor_model_parallel_world_size()}
x = torch.arange(6, dtype=torch.float32).reshape(2, 3) + 10 * rank
rep["sum"] = ps.tensor_model_parallel_all_reduce(x.clone()).tolist()
rep["gather_last"] = ps.tensor_model_parallel_all_gather(x, dim=-1).tolist()
rep["gather_first"] = g.all_gather(x, dim=0).tolist()
out = torch.empty(2, 3)
g.all_reduce_into(x, out)
rep["into"] = out.tolist()
rep["x_untouched"] = x.tolist()
g.check_health()
g.barrier()
ps.destroy_model_parallel()
try:
    ps.get_tp_group()
    rep["gone"] = False
except Exception:
    rep["gone"] = True
print(json.dumps(rep))
dist.destroy_process_group()
'''


def test_group_coordinator_falls_back_to_the_device_group(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", SGLK_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-1500:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    x0 = [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
    x1 = [[10.0, 11.0, 12.0], [13.0, 14.0, 15.0]]
    total = [[a + b for a, b in zip(r0, r1)] for r0, r1 in zip(x0, x1)]
    for rep in outs:
        assert not rep["p2p"] and rep["tp_world"] == 2 and rep["tp_rank"] == rep["rank"]
        assert rep["sum"] == total and rep["into"] == total
        assert rep["x_untouched"] == (x0 if rep["rank"] == 0 else x1)
        assert rep["gather_last"] == [x0[0] + x1[0], x0[1] + x1[1]]
        assert rep["gather_first"] == x0 + x1
        assert rep["gone"]
