"""Native RCCL exchange behind the C-ABI (``gg_comm_*``, ``csrc/comm.cpp``) -- the opt-in alternative (``GG_NATIVE_COMM=1``) to running
the same three collectives through ``torch.distributed`` (whose "nccl" backend IS RCCL on ROCm).  The communicator's 128-byte unique
id is created on rank 0 and handed to the other ranks through the already-initialised ``torch.distributed`` process group (any
backend), i.e. torch.distributed is used as the launcher's rendezvous channel only.

Collectives run on a dedicated HIP stream ordered behind the compute stream by events, so a gradient bucket's all-reduce overlaps
with the rest of the backward pass; ``wait()`` orders the compute stream behind everything enqueued so far."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib as L


def enabled() -> bool:
    return os.environ.get("GG_NATIVE_COMM", "0") not in ("", "0")


class NativeComm:
    def __init__(self, device: Optional[torch.device] = None):
        L.require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        uid = (C.c_char * 128)()
        if rank == 0:
            L.check(L.lib().gg_comm_unique_id(uid), "gg_comm_unique_id")
        box = [bytes(uid)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        self._h = C.c_void_p()
        L.check(L.lib().gg_comm_create(C.byref(self._h), box[0], rank, world, self.device.index or 0), "gg_comm_create")
        self.rank, self.world = rank, world
        self.stream = torch.cuda.Stream(device=self.device)
        self._pending = False

    def accepts(self, t: torch.Tensor, any_dtype: bool = False) -> bool:
        """Tensors this communicator carries itself (everything else goes through torch.distributed): contiguous device tensors, f32 for the sum."""
        return t.is_cuda and t.is_contiguous() and (any_dtype or t.dtype == torch.float32)

    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))      # the collective sees everything enqueued on the compute stream so far
        self._pending = True

    def allreduce_sum_(self, t: torch.Tensor):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._enter()
        L.check(L.lib().gg_comm_allreduce_sum_f32(self._h, t.data_ptr(), t.numel(), self.stream.cuda_stream), "gg_comm_allreduce_sum_f32")
        t.record_stream(self.stream)

    def broadcast_(self, t: torch.Tensor, root: int = 0):
        assert t.is_cuda and t.is_contiguous()
        self._enter()
        L.check(L.lib().gg_comm_broadcast(self._h, t.data_ptr(), t.numel() * t.element_size(), root, self.stream.cuda_stream), "gg_comm_broadcast")
        t.record_stream(self.stream)

    def wait(self):
        if self._pending:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self._pending = False

    def barrier(self, sync: bool = True):
        self._enter()
        L.check(L.lib().gg_comm_barrier(self._h, self.stream.cuda_stream, int(sync)), "gg_comm_barrier")
        self.wait()

    def close(self):
        if self._h:
            L.lib().gg_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
