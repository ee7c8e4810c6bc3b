"""The C-ABI communicator (include/qst.h: qst_comm_init / qst_allreduce_bucket; csrc/comm.hip) from Python.

torch.distributed is what this build's fit() / QuadrupletTrainer use by default; NativeComm is the same exchange step
through libqst.so's own RCCL binding -- what a caller without torch would use -- and can be passed wherever a process group
is accepted (`QuadrupletTrainer(process_group=NativeComm(...))`, `staged_backward(group=...)`). The reference has no
counterpart: it is single-process (training/main.py:113)."""
import ctypes as C
from typing import Optional

import torch

from . import _lib

ID_BYTES = 128


class _Work:
    """What an asynchronous all-reduce returns: wait() orders the caller's current stream after the exchange."""

    def __init__(self, event: torch.cuda.Event):
        self.event = event

    def wait(self) -> bool:
        torch.cuda.current_stream().wait_event(self.event)
        return True


class NativeComm:
    """One RCCL communicator owned by libqst.so. rank 0 calls NativeComm.unique_id() and ships the 128 bytes to its peers
    (a torch.distributed store, a file, MPI ...); every rank then constructs NativeComm(rank, world, id) with its GPU
    current. Collectives run on a stream of their own, ordered against the compute stream with events."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device: Optional[torch.device] = None):
        if len(unique_id) != ID_BYTES:
            raise ValueError(f"the communicator id is {ID_BYTES} bytes")
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.rank, self.world = int(rank), int(world)
        h = _lib.vp()
        with torch.cuda.device(self.device):
            rc = self.lib.qst_comm_init(self.rank, self.world, C.c_char_p(unique_id), C.byref(h))
            if rc:
                raise _lib.QstError(f"qst_comm_init: {self.lib.qst_strerror(rc).decode()}: {self.last_error()}")
            self.stream = torch.cuda.Stream(device=self.device)
        self.handle = h

    @staticmethod
    def unique_id() -> bytes:
        lib = _lib.load()
        buf = C.create_string_buffer(ID_BYTES)
        rc = lib.qst_comm_unique_id(buf)
        if rc:
            raise _lib.QstError(f"qst_comm_unique_id: {lib.qst_strerror(rc).decode()}: {lib.qst_comm_last_error().decode()}")
        return buf.raw

    def last_error(self) -> str:
        return self.lib.qst_comm_last_error().decode()

    def all_reduce(self, t: torch.Tensor, async_op: bool = False):
        """In-place sum of a contiguous fp32 / bf16 device tensor over all ranks. async_op: returns a work whose wait()
        makes the current stream wait for it; otherwise the current stream already does on return."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        self.stream.wait_stream(torch.cuda.current_stream())          # the gradients this bucket carries are complete
        rc = self.lib.qst_allreduce_bucket(self.handle, t.data_ptr(), t.numel(), 0 if t.dtype == torch.float32 else 1,
                                           self.stream.cuda_stream)
        if rc:
            raise _lib.QstError(f"qst_allreduce_bucket: {self.lib.qst_strerror(rc).decode()}: {self.last_error()}")
        t.record_stream(self.stream)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        work = _Work(ev)
        if not async_op:
            work.wait()
            return None
        return work

    def close(self) -> None:
        if getattr(self, "handle", None) is not None:
            torch.cuda.synchronize(self.device)
            self.lib.qst_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
