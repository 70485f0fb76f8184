"""Direct ctypes binding to RCCL (``librccl.so``).

Why not only ``torch.distributed``: these calls are plain kernel enqueues on
the caller's HIP stream - no work objects, no watchdog bookkeeping - so they
can be recorded inside a hipGraph capture of a decode step, and the
communicator can be bootstrapped from a unique id carried over the
framework's own control plane (here: the group's gloo channel) instead of a
second TCP store.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from ..logger import init_logger

logger = init_logger(__name__)

_NCCL_DT = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6,
            torch.float32: 7, torch.float64: 8, torch.bfloat16: 9}
_NCCL_SUM = 0


class ncclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


_lib: Optional[ctypes.CDLL] = None


def load_rccl() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    last: Optional[BaseException] = None
    cands = [os.environ.get("VDAMD_RCCL_PATH"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"]
    try:       # prefer the copy torch already loaded, so there is one RCCL in the process
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(tl):
            cands.insert(1, tl)
    except Exception:  # noqa: BLE001
        pass
    for c in cands:
        if not c:
            continue
        try:
            lib = ctypes.CDLL(c)
            break
        except OSError as e:
            last = e
    else:
        raise OSError(f"librccl.so not found: {last!r}")
    P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [I]
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(ncclUniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(P), I, ncclUniqueId, I]
    lib.ncclAllReduce.argtypes = [P, P, Z, I, I, P, P]
    lib.ncclAllGather.argtypes = [P, P, Z, I, P, P]
    lib.ncclReduceScatter.argtypes = [P, P, Z, I, I, P, P]
    lib.ncclBroadcast.argtypes = [P, P, Z, I, I, P, P]
    lib.ncclSend.argtypes = [P, Z, I, I, P, P]
    lib.ncclRecv.argtypes = [P, Z, I, I, P, P]
    lib.ncclCommDestroy.argtypes = [P]
    lib.ncclCommAbort.argtypes = [P]
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclAllGather", "ncclReduceScatter",
                 "ncclBroadcast", "ncclSend", "ncclRecv", "ncclCommDestroy", "ncclCommAbort", "ncclGroupStart",
                 "ncclGroupEnd"):
        getattr(lib, name).restype = I
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"RCCL error {rc}: {load_rccl().ncclGetErrorString(rc).decode()}")


class RcclCommunicator:
    """One RCCL communicator over a ``GroupCoordinator``'s ranks."""

    def __init__(self, group=None, rank: Optional[int] = None, world_size: Optional[int] = None,
                 unique_id: Optional[bytes] = None):
        self.lib = load_rccl()
        if group is not None:
            rank, world_size = group.rank_in_group, group.world_size
            uid = ncclUniqueId()
            if rank == 0:
                _check(self.lib.ncclGetUniqueId(ctypes.byref(uid)))
            # the id reaches the other ranks over the control plane (the head relays it) when there is one
            blob = group.bootstrap_all_gather(bytes(uid.internal) if rank == 0 else None, "rccl-id")[0]
        else:
            assert rank is not None and world_size is not None
            if unique_id is None:
                uid = ncclUniqueId()
                _check(self.lib.ncclGetUniqueId(ctypes.byref(uid)))
                blob = bytes(uid.internal)
            else:
                blob = unique_id
        uid = ncclUniqueId()
        ctypes.memmove(ctypes.byref(uid), blob, 128)
        self.rank, self.world_size = rank, world_size
        self.comm = ctypes.c_void_p()
        _check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), world_size, uid, rank))

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def all_reduce(self, x: torch.Tensor) -> None:
        assert x.is_contiguous()
        _check(self.lib.ncclAllReduce(x.data_ptr(), x.data_ptr(), x.numel(), _NCCL_DT[x.dtype], _NCCL_SUM,
                                      self.comm, self._stream()))

    def all_gather(self, out: torch.Tensor, x: torch.Tensor) -> None:
        assert out.is_contiguous() and x.is_contiguous() and out.numel() == x.numel() * self.world_size
        _check(self.lib.ncclAllGather(x.data_ptr(), out.data_ptr(), x.numel(), _NCCL_DT[x.dtype], self.comm,
                                      self._stream()))

    def reduce_scatter(self, out: torch.Tensor, x: torch.Tensor) -> None:
        """``out`` = this rank's ``1/world`` slice (along dim 0) of the sum of ``x``."""
        assert out.is_contiguous() and x.is_contiguous() and x.numel() == out.numel() * self.world_size
        _check(self.lib.ncclReduceScatter(x.data_ptr(), out.data_ptr(), out.numel(), _NCCL_DT[x.dtype],
                                          _NCCL_SUM, self.comm, self._stream()))

    def broadcast(self, x: torch.Tensor, src: int) -> None:
        assert x.is_contiguous()
        _check(self.lib.ncclBroadcast(x.data_ptr(), x.data_ptr(), x.numel(), _NCCL_DT[x.dtype], src, self.comm,
                                      self._stream()))

    def send(self, x: torch.Tensor, dst: int) -> None:
        assert x.is_contiguous()
        _check(self.lib.ncclSend(x.data_ptr(), x.numel(), _NCCL_DT[x.dtype], dst, self.comm, self._stream()))

    def recv(self, x: torch.Tensor, src: int) -> None:
        assert x.is_contiguous()
        _check(self.lib.ncclRecv(x.data_ptr(), x.numel(), _NCCL_DT[x.dtype], src, self.comm, self._stream()))

    def group_start(self) -> None:
        """Open a group: the point-to-point calls up to ``group_end`` are issued as one operation
        (a send and the matching receive of one rank cannot deadlock each other)."""
        _check(self.lib.ncclGroupStart())

    def group_end(self) -> None:
        _check(self.lib.ncclGroupEnd())

    def send_recv(self, send: Optional[torch.Tensor], dst: int, recv: Optional[torch.Tensor], src: int) -> None:
        """One grouped exchange: ``send`` goes to rank ``dst`` while ``recv`` is filled from rank ``src``."""
        self.group_start()
        try:
            if send is not None:
                self.send(send, dst)
            if recv is not None:
                self.recv(recv, src)
        finally:
            self.group_end()

    def close(self, timeout: float = 5.0) -> None:
        """Free the communicator.  RCCL does not return from destroying (or aborting) a
        communicator while a hipGraph that captured one of its calls is alive, and decode
        graphs may well outlive it at shutdown: the call runs on a helper thread that is
        abandoned after ``timeout`` seconds (process exit reclaims what it holds)."""
        comm, self.comm = getattr(self, "comm", None), ctypes.c_void_p()
        if comm is None or not comm.value:
            return
        import threading

        def destroy() -> None:
            try:
                self.lib.ncclCommAbort(comm)
            except Exception:  # noqa: BLE001
                pass

        t = threading.Thread(target=destroy, name="rccl-close", daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            logger.warning("RCCL communicator still referenced by a captured graph: left to process exit")
