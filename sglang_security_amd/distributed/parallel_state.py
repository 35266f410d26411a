"""The tensor collectives of a model-parallel group: peer-to-peer kernels when they apply, RCCL otherwise.

Mirror of the tensor-collective part of GroupCoordinator
(python/sglang/srt/distributed/parallel_state.py:166-315 constructor, :398-461 all_reduce, :463-540
reduce_scatter / all_gather_into_tensor / all_gather, :893-900 barrier, :930-946 destroy): the same dispatch
order -- world size 1 returns the input, the custom all-reduce when `should_custom_ar` accepts the tensor, else
`torch.distributed.all_reduce` on the device group, which on ROCm with backend "nccl" is RCCL over xGMI.  The
object / tensor-dict broadcast helpers of the reference class are control plane and are not mirrored.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .._lib import SglkError
from .custom_all_reduce import CustomAllreduce


class GroupCoordinator:
    def __init__(self, group_ranks: List[List[int]], local_rank: int, torch_distributed_backend: str = "nccl",
                 use_custom_allreduce: bool = True, group_name: Optional[str] = None,
                 max_size: int = CustomAllreduce._MAX_CAR_SIZE, full_xgmi: bool = True,
                 timeout_ms: Optional[int] = None):
        self.unique_name = group_name or "anonymous"
        self.rank = dist.get_rank()
        self.local_rank = local_rank
        self.device_group = None
        self.cpu_group = None
        for ranks in group_ranks:
            device_group = dist.new_group(ranks, backend=torch_distributed_backend)
            cpu_group = dist.new_group(ranks, backend="gloo")  # handle exchange and host-side coordination
            if self.rank in ranks:
                self.ranks = list(ranks)
                self.world_size = len(ranks)
                self.rank_in_group = self.ranks.index(self.rank)
                self.device_group = device_group
                self.cpu_group = cpu_group
        if self.cpu_group is None or self.device_group is None:
            raise SglkError(f"rank {self.rank} is in none of the groups {group_ranks}")
        self.device = torch.device(f"cuda:{local_rank}") if torch.cuda.is_available() else torch.device("cpu")
        self.use_custom_allreduce = use_custom_allreduce
        # which path each all_reduce_into call took (a benchmark reads the difference over a step)
        self.ar_calls = {"peer_to_peer_registered": 0, "peer_to_peer_unregistered": 0, "rccl": 0}
        self.ca_comm: Optional[CustomAllreduce] = None
        self.ca_error: Optional[str] = None
        if use_custom_allreduce and self.world_size > 1 and self.device.type == "cuda":
            try:  # parallel_state.py:268-280: a failed set-up leaves the RCCL path
                self.ca_comm = CustomAllreduce(self.cpu_group, self.device, max_size=max_size, full_xgmi=full_xgmi,
                                               timeout_ms=timeout_ms)
            except Exception as e:  # noqa: BLE001
                self.ca_error = str(e)

    # ---- all-reduce ---------------------------------------------------------------------------------
    def _custom(self, input_: torch.Tensor) -> bool:
        ca = self.ca_comm
        return ca is not None and not ca.disabled and ca.should_custom_ar(input_)

    def all_reduce(self, input_: torch.Tensor) -> torch.Tensor:
        """parallel_state.py:398-446: out of place through the peer-to-peer kernels, in place through RCCL."""
        if self.world_size == 1:
            return input_
        if self._custom(input_):
            out = self.ca_comm.custom_all_reduce(input_)
            assert out is not None
            return out
        dist.all_reduce(input_, group=self.device_group)
        return input_

    def all_reduce_into(self, inp: torch.Tensor, out: torch.Tensor, registered: bool = False) -> torch.Tensor:
        """out = sum over ranks of inp, for callers that own both buffers (HIP-graph capture): `registered` says
        that inp went through register_buffer(), which saves the copy into the communicator's own buffer."""
        if self.world_size == 1:
            if out.data_ptr() != inp.data_ptr():
                out.copy_(inp)
            return out
        if self._custom(inp):
            if registered:
                self.ar_calls["peer_to_peer_registered"] += 1
                return self.ca_comm.all_reduce_reg(inp, out)
            self.ar_calls["peer_to_peer_unregistered"] += 1
            return self.ca_comm.all_reduce_unreg(inp, out)
        self.ar_calls["rccl"] += 1
        if out.data_ptr() != inp.data_ptr():
            out.copy_(inp)
        dist.all_reduce(out, group=self.device_group)
        return out

    def register_buffer(self, t: torch.Tensor) -> bool:
        """True when the peer-to-peer kernels can take `t` as a registered input from now on (collective call)."""
        if self.ca_comm is None or self.ca_comm.disabled:
            return False
        self.ca_comm.register_buffer(t)
        return True

    # ---- gather / scatter -----------------------------------------------------------------------------
    def reduce_scatter(self, output: torch.Tensor, input_list: List[torch.Tensor]) -> torch.Tensor:
        dist.reduce_scatter(output, input_list, group=self.device_group)
        return output

    def all_gather_into_tensor(self, output: torch.Tensor, input: torch.Tensor, registered: bool = False) -> None:
        """parallel_state.py:472-487.  With `registered` the peer-to-peer gather kernel runs (graph-capturable, no
        RCCL call); otherwise RCCL's all-gather on the device group."""
        ca = self.ca_comm
        if registered and ca is not None and not ca.disabled and ca.should_custom_ag(input):
            ca.all_gather_reg(input, output.view((self.world_size,) + tuple(input.shape)))
            return
        dist.all_gather_into_tensor(output, input, group=self.device_group)

    def all_gather(self, input_: torch.Tensor, dim: int = -1, tensor_list: Optional[List[torch.Tensor]] = None,
                   registered: bool = False) -> torch.Tensor:
        """parallel_state.py:489-540: concatenation of every rank's tensor along `dim`."""
        if self.world_size == 1:
            return input_
        if tensor_list is not None:
            return dist.all_gather(tensor_list, input_, group=self.device_group)
        if not -input_.dim() <= dim < input_.dim():
            raise SglkError(f"Invalid dim ({dim}) for input tensor with shape {tuple(input_.size())}")
        if dim < 0:
            dim += input_.dim()
        size = tuple(input_.size())
        out = torch.empty((size[0] * self.world_size,) + size[1:], dtype=input_.dtype, device=input_.device)
        self.all_gather_into_tensor(out, input_.contiguous() if not registered else input_, registered)
        out = out.reshape((self.world_size,) + size).movedim(0, dim)
        return out.reshape(size[:dim] + (self.world_size * size[dim],) + size[dim + 1:])

    # ---- life cycle ------------------------------------------------------------------------------------
    def barrier(self) -> None:
        dist.barrier(group=self.cpu_group)

    def check_health(self) -> None:
        """After a synchronise: raises when the peer-to-peer kernels timed out waiting for a peer (outputs since then
        are NaN-filled).  The communicator is disabled from then on, so later calls are routed to the device group's
        collective -- which helps when the peer was only late, not when it is gone: a collective over a group with a
        dead rank blocks like any other.  Treat the error as fatal for the step and rebuild the group."""
        if self.ca_comm is not None:
            self.ca_comm.check_health()

    def destroy(self) -> None:
        ca, self.ca_comm = self.ca_comm, None
        try:
            if ca is not None:
                ca.close()
        finally:
            for g in (self.device_group, self.cpu_group):
                if g is not None:
                    dist.destroy_process_group(g)
            self.device_group = self.cpu_group = None


_TP: Optional[GroupCoordinator] = None


def init_model_parallel_group(group_ranks: List[List[int]], local_rank: int, backend: str,
                              use_custom_allreduce: bool = True, group_name: Optional[str] = None,
                              **kw) -> GroupCoordinator:
    """parallel_state.py:969-993."""
    return GroupCoordinator(group_ranks, local_rank, backend, use_custom_allreduce, group_name, **kw)


def initialize_model_parallel(tensor_model_parallel_size: int = 1, backend: Optional[str] = None,
                              local_rank: Optional[int] = None, **kw) -> GroupCoordinator:
    """The tensor-parallel groups of parallel_state.py:1102-1178: consecutive ranks form a group."""
    global _TP
    world = dist.get_world_size()
    if world % tensor_model_parallel_size:
        raise SglkError(f"world size {world} is not a multiple of the tensor-parallel size {tensor_model_parallel_size}")
    backend = backend or dist.get_backend()
    groups = [list(range(i, i + tensor_model_parallel_size)) for i in range(0, world, tensor_model_parallel_size)]
    if local_rank is None:
        local_rank = torch.cuda.current_device() if torch.cuda.is_available() else 0
    _TP = init_model_parallel_group(groups, local_rank, backend, group_name="tp", **kw)
    return _TP


def get_tp_group() -> GroupCoordinator:
    if _TP is None:
        raise SglkError("tensor model parallel group is not initialized")
    return _TP


def get_tensor_model_parallel_world_size() -> int:
    return get_tp_group().world_size


def get_tensor_model_parallel_rank() -> int:
    return get_tp_group().rank_in_group


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """communication_op.py:11-13."""
    return get_tp_group().all_reduce(input_)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """communication_op.py:16-20."""
    return get_tp_group().all_gather(input_, dim)


def destroy_model_parallel() -> None:
    global _TP
    if _TP is not None:
        _TP.destroy()
    _TP = None
