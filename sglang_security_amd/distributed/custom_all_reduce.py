"""Peer-to-peer all-reduce communicator over the sglk_ar_* entry points.

Mirror of CustomAllreduce (python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:146-510):
same constructor arguments and the same methods (capture, should_custom_ar, all_reduce_reg, all_reduce_unreg,
custom_all_reduce, register_buffer, register_graph_buffers, close), built like the reference's HIP branch on the
`sgl_kernel.allreduce` functions.  Handles are exchanged through the given CPU (gloo) process group; the library
then maps the peers' signal blocks and input buffers with hipIpcOpenMemHandle.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.distributed as dist

from contextlib import contextmanager

from .._lib import SglkError, check, dtype_code, lib


def is_weak_contiguous(inp: torch.Tensor) -> bool:
    """custom_all_reduce.py:139-143."""
    return inp.is_contiguous() or (
        inp.untyped_storage().nbytes() - inp.storage_offset() * inp.element_size() == inp.numel() * inp.element_size())


class CustomAllreduce:
    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]
    _MAX_CAR_SIZE = 2 * 8192 * 1024  # the ROCm value (custom_all_reduce.py:148-151)

    def __init__(self, group, device, max_size: int = _MAX_CAR_SIZE, full_xgmi: bool = True,
                 timeout_ms: Optional[int] = None) -> None:
        """The HIP branch of the reference constructor (custom_all_reduce.py:282-298), op for op: allocate_meta_buffer
        -> get_meta_buffer_ipc_handle -> handle exchange over the CPU group -> init_custom_ar -> register_buffer of
        the staging buffer.  full_xgmi is the reference's full_nvlink (every pair of GPUs one xGMI hop apart)."""
        self._ptr = 0
        self._IS_CAPTURING = False
        self.disabled = True
        self.failed = False
        self.meta = None
        if dist.get_backend(group) == "nccl":
            raise SglkError("CustomAllreduce needs a CPU (gloo) group for the handle exchange")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        if self.world_size == 1:
            return  # nothing to reduce: stays disabled, as in the reference (:193-195)
        if self.world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        if isinstance(device, int):
            device = torch.device(f"cuda:{device}")
        elif isinstance(device, str):
            device = torch.device(device)
        self.device = device
        self.max_size = max_size
        self.full_xgmi = self.full_nvlink = full_xgmi
        from ..sgl_kernel import allreduce as ops

        self._ops = ops
        torch.cuda.set_device(device)
        # signal block + staging area of the two-stage algorithm: uncached, zero-filled (:283-284)
        self.meta = ops.allocate_meta_buffer(ops.meta_size() + max_size)
        # eager inputs are copied into this pre-registered buffer (:285)
        self.buffer = torch.empty(max_size, dtype=torch.uint8, device=device)
        handle = ops.get_meta_buffer_ipc_handle(self.meta)
        handles, offsets = self._gather_ipc_meta((bytes(handle.numpy().tobytes()), 0))
        # one 64-byte record of peer pointers per registered or captured input (:289-291)
        self.rank_data = torch.empty(8 * 1024 * 1024, dtype=torch.uint8, device=device)
        self._ptr = ops.init_custom_ar(self.meta, self.rank_data, handles, offsets, self.rank, self.full_nvlink)
        if timeout_ms is not None:
            check(lib().sglk_ar_set_timeout_ms(self._ptr, int(timeout_ms)), "ar_set_timeout_ms")
        self.register_buffer(self.buffer)
        self.disabled = False

    # ---- handle exchange (custom_all_reduce.py:335-411) ---------------------------------------------------
    @contextmanager
    def capture(self):
        """Wrap a graph capture: all-reduce launches recorded inside it name inputs that no peer has mapped yet;
        leaving the context exchanges their handles and fills the records in (register_graph_buffers)."""
        try:
            self._IS_CAPTURING = True
            yield
        finally:
            self._IS_CAPTURING = False
            if not self.disabled:
                self.register_graph_buffers()

    def _get_ipc_meta(self, inp: torch.Tensor):
        """(handle of the allocation that holds inp, inp's offset in it), gathered over the ranks.  The reference asks
        for the handle of inp's own address and sends offset 0 (:350-358), which holds for a tensor that starts its
        allocation; a tensor the caching allocator carved out of a larger segment needs the segment's base."""
        L = lib()
        buf = (C.c_char * L.sglk_ipc_handle_bytes())()
        off = C.c_int64(0)
        check(L.sglk_ipc_get_handle(inp.data_ptr(), buf, C.byref(off)), "ipc_get_handle")
        return self._gather_ipc_meta((bytes(buf), int(off.value)))

    def _gather_ipc_meta(self, shard_data):
        """custom_all_reduce.py:360-382: broadcast_object_list per rank over the CPU group (all_gather_object is not
        usable with gloo under inference mode)."""
        all_data = [[None] for _ in range(self.world_size)]
        all_data[self.rank][0] = shard_data
        ranks = sorted(dist.get_process_group_ranks(group=self.group))
        for i, rank in enumerate(ranks):
            dist.broadcast_object_list(all_data[i], src=rank, group=self.group, device="cpu")
        handles = [d[0][0] for d in all_data]
        offsets = [d[0][1] for d in all_data]
        return handles, offsets

    def register_buffer(self, inp: torch.Tensor):
        """custom_all_reduce.py:384-386."""
        handles, offsets = self._get_ipc_meta(inp)
        self._ops.register_buffer(self._ptr, inp, handles, offsets)

    def register_graph_buffers(self):
        """custom_all_reduce.py:388-393 (HIP branch)."""
        handle, offset = self._ops.get_graph_buffer_ipc_meta(self._ptr)
        handles, offsets = self._gather_ipc_meta((bytes(handle.numpy().tobytes()), offset))
        self._ops.register_graph_buffers(self._ptr, handles, offsets)

    # ---- reductions -------------------------------------------------------------------------------
    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        """custom_all_reduce.py:414-436."""
        if self.disabled:
            return False
        inp_size = inp.numel() * inp.element_size()
        if inp_size % 16 != 0 or not is_weak_contiguous(inp):
            return False
        if inp.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            return False
        if self.world_size == 2 or self.full_xgmi:
            return inp_size < self.max_size
        return False

    def all_reduce_reg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None, algo: int = 0):
        if out is None:
            out = torch.empty_like(inp)
        check(lib().sglk_ar_all_reduce(self._ptr, inp.data_ptr(), out.data_ptr(), inp.numel(), dtype_code(inp.dtype), algo,
                                       torch.cuda.current_stream().cuda_stream), "ar_all_reduce")
        return out

    def all_reduce_unreg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None, algo: int = 0):
        if out is None:
            out = torch.empty_like(inp)
        check(lib().sglk_ar_all_reduce_unreg(self._ptr, inp.data_ptr(), self.buffer.data_ptr(), out.data_ptr(),
                                             inp.numel(), dtype_code(inp.dtype), algo,
                                             torch.cuda.current_stream().cuda_stream), "ar_all_reduce_unreg")
        return out

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """custom_all_reduce.py:475-498: None when this communicator does not take the tensor; inside capture() the
        captured launch reads the input where it lies (registered after the capture), the warm-up run before the
        capture only mimics the allocation; outside, the input is copied into the registered staging buffer."""
        if self.disabled or not self.should_custom_ar(input):
            return None
        if self._IS_CAPTURING:
            if torch.cuda.is_current_stream_capturing():
                return self.all_reduce_reg(input)
            return torch.empty_like(input)
        return self.all_reduce_unreg(input)

    def all_gather_reg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Every rank's registered `inp`, stacked in rank order: out is [world_size, *inp.shape]."""
        if out is None:
            out = torch.empty((self.world_size,) + tuple(inp.shape), dtype=inp.dtype, device=inp.device)
        if out.numel() != self.world_size * inp.numel() or out.dtype != inp.dtype or not out.is_contiguous():
            raise SglkError("all_gather_reg: out must be a contiguous [world_size, *inp.shape] tensor of inp's dtype")
        check(lib().sglk_ar_all_gather(self._ptr, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size(),
                                       torch.cuda.current_stream().cuda_stream), "ar_all_gather")
        return out

    def should_custom_ag(self, inp: torch.Tensor) -> bool:
        if self.disabled:
            return False
        size = inp.numel() * inp.element_size()
        return size % 16 == 0 and is_weak_contiguous(inp) and size * self.world_size <= 8 * self.max_size

    # ---- failure handling ---------------------------------------------------------------------------
    def stuck_blocks(self) -> int:
        """1 once a barrier wait on this rank has timed out (a host-pinned word the kernels raise; no synchronise)."""
        return int(lib().sglk_ar_stuck(self._ptr)) if self._ptr else 0

    def stuck_detail(self):
        """After a time-out: [(block, ticket it waited for, the ticket each peer had written for that barrier)], read
        from this rank's signal block (64 tickets, 2 x 64 x 8 peer slots, 64 stuck words: the layout of `Signal` in
        allreduce.hip).  A peer whose entry is below the ticket had not arrived."""
        if self.meta is None:
            return []
        raw = self.meta[: 4 * 1153].cpu().numpy().view("uint32")
        slot, stuck = raw[64:1088].reshape(2, 64, 8), raw[1088:1152]
        return [(b, int(stuck[b]), [int(v) for v in slot[int(stuck[b]) & 1, b, : self.world_size]])
                for b in range(64) if stuck[b]]

    def check_health(self) -> None:
        """Call after the stream has been synchronised (the end of a decode step, before results leave the GPU).
        A peer that never reached a barrier has left NaN-filled outputs behind: this raises SglkError and disables
        the communicator, so that should_custom_ar() declines and the caller's fallback collective takes over."""
        if self._ptr and not self.failed and self.stuck_blocks():
            self.failed = True
            self.disabled = True
        if self.failed:
            raise SglkError(
                f"peer-to-peer all-reduce: rank {self.rank} waited for a peer longer than the time limit; outputs since "
                f"then are NaN-filled and this communicator is disabled (block, ticket, peers' tickets: "
                f"{self.stuck_detail()[:4]})")

    def close(self):
        failed = False
        if self._ptr:
            torch.cuda.synchronize()
            failed = not self.failed and bool(self.stuck_blocks())
            self._ops.dispose(self._ptr)  # also closes the peer mappings this communicator opened
            self._ptr = 0
            self.disabled = True
        self.meta = None  # the tensor owns the uncached allocation
        if failed:
            self.failed = True
            raise SglkError(f"peer-to-peer all-reduce: rank {self.rank} had a barrier time-out that nobody checked for; "
                            "results computed through this communicator are NaN-filled")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
