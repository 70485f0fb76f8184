"""Python side of the peer-mapped direct all-reduce (``csrc/custom_all_reduce.hip``).

Bring-up: every rank allocates its staging + signal memory, the IPC handles
are all-gathered over the group's CPU (gloo) channel, and each rank maps its
peers.  ``maybe_all_reduce`` takes messages up to ``max_bytes`` (bf16/fp16,
16-byte multiple) and returns ``None`` for anything else so the caller falls
back to RCCL.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ..logger import init_logger

logger = init_logger(__name__)

_DT = {torch.bfloat16: 0, torch.float16: 1}


class CustomAllReduce:
    ONE_SHOT_BYTES = 512 * 1024

    def __init__(self, group, max_bytes: int = 8 << 20):
        from ..ops import hip
        self.lib = hip.load()
        self._bind()
        self.group = group
        self.rank, self.world = group.rank_in_group, group.world_size
        if not 2 <= self.world <= 8:
            raise ValueError("direct all-reduce supports 2..8 ranks")
        self.max_bytes = int(max_bytes)
        hb = int(self.lib.vdamd_car_handle_bytes())
        dh, sh = ctypes.create_string_buffer(hb), ctypes.create_string_buffer(hb)
        self.ctx = self.lib.vdamd_car_create(self.rank, self.world, self.max_bytes, dh, sh)
        if not self.ctx:
            raise RuntimeError("could not allocate IPC staging memory")
        handles = group.all_gather_object((dh.raw, sh.raw))
        data_handles = b"".join(h[0] for h in handles)
        sig_handles = b"".join(h[1] for h in handles)
        rc = self.lib.vdamd_car_open_peers(self.ctx, data_handles, sig_handles)
        if rc != 0:
            self.close()
            raise RuntimeError(f"hipIpcOpenMemHandle failed with {rc}")
        group.barrier()
        self.disabled = False

    def _bind(self) -> None:
        lib = self.lib
        P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        lib.vdamd_car_handle_bytes.restype = I
        lib.vdamd_car_create.argtypes = [I, I, Z, P, P]
        lib.vdamd_car_create.restype = P
        lib.vdamd_car_open_peers.argtypes = [P, ctypes.c_char_p, ctypes.c_char_p]
        lib.vdamd_car_open_peers.restype = I
        lib.vdamd_car_timed_out.argtypes = [P]
        lib.vdamd_car_timed_out.restype = I
        lib.vdamd_car_all_reduce.argtypes = [P, P, P, Z, I, I, Z, P]
        lib.vdamd_car_all_reduce.restype = I
        lib.vdamd_car_destroy.argtypes = [P]
        lib.vdamd_car_destroy.restype = None

    def should_use(self, x: torch.Tensor) -> bool:
        nbytes = x.numel() * x.element_size()
        return (not self.disabled and x.is_cuda and x.dtype in _DT and x.is_contiguous()
                and nbytes % 16 == 0 and 0 < nbytes <= self.max_bytes)

    def all_reduce(self, x: torch.Tensor, mode: int = 0) -> torch.Tensor:
        out = torch.empty_like(x)
        rc = self.lib.vdamd_car_all_reduce(self.ctx, x.data_ptr(), out.data_ptr(),
                                           x.numel() * x.element_size(), _DT[x.dtype], mode,
                                           self.ONE_SHOT_BYTES, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"direct all-reduce launch failed with {rc}")
        return out

    def maybe_all_reduce(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        return self.all_reduce(x) if self.should_use(x) else None

    def timed_out(self) -> bool:
        return bool(self.lib.vdamd_car_timed_out(self.ctx))

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.vdamd_car_destroy(self.ctx)
            self.ctx = None
