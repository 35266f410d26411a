"""CPU, world_size 2 over gloo: the multi-rank timing path of bench.py (barriers on both sides, max over ranks,
whole-job aggregate).  The decode step itself needs a GPU; here each rank runs a stand-in step of a known duration."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["SGLK_ROOT"])
import torch.distributed as dist
from sglang_security_amd.benchlib import aggregate, dist_helpers, timed_steps
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
barrier, mx = dist_helpers(world)
calls = [0]
def run():
    calls[0] += 1
    time.sleep(0.01 if rank == 0 else 0.03)      # rank 1 is three times slower
elapsed = timed_steps(run, steps=10, warmup=3, sync=lambda: None, barrier=barrier, max_over_ranks=mx)
out = aggregate(64, 10, world, elapsed)
out.update(rank=rank, calls=calls[0], elapsed=elapsed)
print(json.dumps(out))
dist.destroy_process_group()
'''


def test_two_rank_timing(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", SGLK_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-1500:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    a, b = outs
    assert a["calls"] == b["calls"] == 13                      # 3 warm-up + exactly 10 timed
    assert abs(a["elapsed"] - b["elapsed"]) < 1e-9              # both report the max over ranks
    assert 0.29 < a["elapsed"] < 0.6                            # the slow rank's 10 x 30 ms, not the fast rank's 100 ms
    assert abs(a["value"] - 64 * 10 * 2 / a["elapsed"]) < 1e-6  # whole-job aggregate over both ranks
    assert abs(a["ms_per_step"] - a["elapsed"] * 100) < 1e-6


def test_single_process_has_no_collective():
    sys.path.insert(0, ROOT)
    from sglang_security_amd.benchlib import dist_helpers, timed_steps

    assert dist_helpers(1) == (None, None)
    n = [0]
    t = timed_steps(lambda: n.__setitem__(0, n[0] + 1), steps=5, warmup=2, sync=lambda: None)
    assert n[0] == 7 and t >= 0
