"""Timing harness shared by bench.py and its CPU test: W untimed warm-up steps, then exactly K steps bracketed by
barrier + device synchronise on both sides, the maximum over ranks, and the whole-job aggregate."""
from __future__ import annotations

import time
from typing import Callable, Optional


def timed_steps(run: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None],
                barrier: Optional[Callable[[], None]] = None,
                max_over_ranks: Optional[Callable[[float], float]] = None) -> float:
    """Returns the elapsed seconds of `steps` calls of `run` (max over ranks when a reducer is given)."""
    for _ in range(warmup):
        run()
    sync()
    if barrier:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    sync()
    if barrier:
        barrier()
    elapsed = time.perf_counter() - t0
    if max_over_ranks:
        elapsed = max_over_ranks(elapsed)
    return elapsed


def aggregate(units_per_step_per_rank: int, steps: int, world: int, elapsed: float) -> dict:
    """Whole-job throughput: every rank processed units_per_step_per_rank * steps units in `elapsed` seconds."""
    return {"value": units_per_step_per_rank * steps * world / elapsed, "ms_per_step": elapsed / steps * 1e3}


def dist_helpers(world: int, device=None):
    """(barrier, max_over_ranks) on the default process group, or (None, None) for a single process."""
    if world <= 1:
        return None, None
    import torch
    import torch.distributed as dist

    def barrier():
        dist.barrier()

    def max_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    return barrier, max_over_ranks
