"""Peer-to-peer all-reduce communicator over the sglk_ar_* entry points.

Mirror of CustomAllreduce (python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:146-510):
same constructor arguments and the same methods (should_custom_ar, all_reduce_reg, all_reduce_unreg,
custom_all_reduce, register_buffer, close).  Handles are exchanged through the given CPU (gloo) process
group; every rank then maps the peers' signal blocks and input buffers with hipIpcOpenMemHandle.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.distributed as dist

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
        self._ptr = 0
        self.disabled = True
        self.failed = False
        self._opened: List[int] = []
        self._meta = 0
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
        self.full_xgmi = full_xgmi
        L = lib()
        torch.cuda.set_device(device)
        total = int(L.sglk_ar_meta_size()) + max_size
        self._meta = L.sglk_alloc_uncached(total)
        if not self._meta:
            check(-1, "alloc_uncached")
        metas = self._exchange(self._meta)
        arr = (C.c_void_p * self.world_size)(*metas)
        self._ptr = L.sglk_ar_init(arr, self.rank, self.world_size, 1 if full_xgmi else 0, max_size)
        if not self._ptr:
            check(-1, "ar_init")
        if timeout_ms is not None:
            check(L.sglk_ar_set_timeout_ms(self._ptr, int(timeout_ms)), "ar_set_timeout_ms")
        # the staging input buffer for tensors that are not registered themselves
        self.buffer = torch.empty(max_size, dtype=torch.uint8, device=device)
        self.register_buffer(self.buffer)
        self.disabled = False

    # ---- handle exchange ------------------------------------------------------------------------
    def _exchange(self, local_ptr: int) -> List[int]:
        """All-gather the IPC handle of local_ptr and open the peers' handles; returns one address per rank."""
        L = lib()
        n = L.sglk_ipc_handle_bytes()
        buf = (C.c_char * n)()
        off = C.c_int64(0)
        check(L.sglk_ipc_get_handle(local_ptr, buf, C.byref(off)), "ipc_get_handle")
        mine = (bytes(buf), int(off.value))
        gathered = [None] * self.world_size
        dist.all_gather_object(gathered, mine, group=self.group)
        ptrs = []
        for r, (h, o) in enumerate(gathered):
            if r == self.rank:
                ptrs.append(local_ptr)
                continue
            base = L.sglk_ipc_open_handle(h)
            if not base:
                check(-1, "ipc_open_handle")
            self._opened.append(base)
            ptrs.append(base + o)
        return ptrs

    def register_buffer(self, inp: torch.Tensor):
        """custom_all_reduce.py:384-386."""
        ptrs = self._exchange(inp.data_ptr())
        arr = (C.c_void_p * self.world_size)(*ptrs)
        check(lib().sglk_ar_register_buffer(self._ptr, inp.data_ptr(), arr), "ar_register_buffer")

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
        """custom_all_reduce.py:475-498: None when this communicator does not take the tensor."""
        if not self.should_custom_ar(input):
            return None
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
                "then are NaN-filled and this communicator is disabled")

    def close(self):
        L = lib()
        failed = False
        if self._ptr:
            torch.cuda.synchronize()
            failed = not self.failed and bool(self.stuck_blocks())
            L.sglk_ar_dispose(self._ptr)
            self._ptr = 0
            self.disabled = True
        for p in self._opened:
            L.sglk_ipc_close(p)
        self._opened = []
        if self._meta:
            L.sglk_free_uncached(self._meta)
            self._meta = 0
        if failed:
            self.failed = True
            raise SglkError(f"peer-to-peer all-reduce: rank {self.rank} had a barrier time-out that nobody checked for; "
                            "results computed through this communicator are NaN-filled")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
