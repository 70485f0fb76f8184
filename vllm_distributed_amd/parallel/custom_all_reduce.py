"""Python side of the peer-mapped direct collectives (``csrc/custom_all_reduce.hip``).

Bring-up: every rank allocates its uncached data region and signal block, the
IPC handles are all-gathered over the group's CPU (gloo) channel, and each
rank maps its peers.

Calls alternate between two copies of every buffer; ``parity`` (0/1) is the
copy the next call uses and is tracked on the host.  Every rank issues the
same sequence of calls, so the parities agree without communication.  A model
forward must contain an even number of calls (``end_step`` pads with a
barrier-only call): a captured hipGraph then starts and ends at parity 0 and
can be replayed between eager steps.

Producers write their partial sums straight into the stage area of the next
call (``stage_view``), so no copy precedes the reduction; a tensor that lives
elsewhere is copied in.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)

_DT = {torch.bfloat16: 0, torch.float16: 1}


class _DevicePointer:
    """Zero-copy view of raw device memory for ``torch.as_tensor``."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class CustomAllReduce:
    def __init__(self, group, max_bytes: int = 16 << 20, ranks_per_device: int = 1):
        from ..ops import hip
        self.lib = hip.load()
        self._bind()
        self.group = group
        self.rank, self.world = group.rank_in_group, group.world_size
        if not 2 <= self.world <= 8:
            raise ValueError("direct all-reduce supports 2..8 ranks")
        self.max_bytes = (int(max_bytes) + 255) // 256 * 256
        # messages up to this size are summed by every rank in one pass; larger ones in two.
        # "auto": the bring-up times both forms on this machine's links (parallel/accel.py)
        spec = str(envs.VDAMD_CUSTOM_ALL_REDUCE_ONE_SHOT_BYTES).strip().lower()
        self.auto_thresholds = spec == "auto"
        self.one_shot_bytes = (256 << 10) if self.auto_thresholds else int(spec)
        # largest message the callers hand to this path (<= max_bytes, the capacity)
        self.use_bytes = self.max_bytes
        self.fence_start = str(envs.VDAMD_CUSTOM_ALL_REDUCE_FENCE_START).strip().lower() in ("1", "true", "yes", "on")
        hb = int(self.lib.vdamd_car_handle_bytes())
        dh, sh = ctypes.create_string_buffer(hb), ctypes.create_string_buffer(hb)
        self.ctx = self.lib.vdamd_car_create(self.rank, self.world, self.max_bytes, dh, sh)
        if not self.ctx:
            raise RuntimeError("could not allocate IPC staging memory")
        handles = group.bootstrap_all_gather((dh.raw, sh.raw), "ipc-handles")
        data_handles = b"".join(h[0] for h in handles)
        sig_handles = b"".join(h[1] for h in handles)
        rc = self.lib.vdamd_car_open_peers(self.ctx, data_handles, sig_handles)
        if rc != 0:
            self.close()
            raise RuntimeError(f"hipIpcOpenMemHandle failed with {rc}")
        self.lib.vdamd_car_set_timeout_ms(self.ctx, int(envs.VDAMD_CUSTOM_ALL_REDUCE_TIMEOUT_MS))
        if ranks_per_device > 1:
            # all grids must be resident together for the in-kernel barriers
            self.lib.vdamd_car_set_max_blocks(self.ctx, max(1, 512 // (4 * ranks_per_device)))
        self.parity = 0
        self.calls = 0
        self._stage = []
        for p in (0, 1):
            ptr = self.lib.vdamd_car_stage_ptr(self.ctx, p)
            self._stage.append(torch.as_tensor(_DevicePointer(ptr, self.max_bytes), device=group.device))
        group.barrier()
        self.disabled = False

    def _bind(self) -> None:
        lib = self.lib
        P, I, Z, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float
        lib.vdamd_car_handle_bytes.restype = I
        lib.vdamd_car_create.argtypes = [I, I, Z, P, P]
        lib.vdamd_car_create.restype = P
        lib.vdamd_car_open_peers.argtypes = [P, ctypes.c_char_p, ctypes.c_char_p]
        lib.vdamd_car_open_peers.restype = I
        lib.vdamd_car_stage_ptr.argtypes = [P, I]
        lib.vdamd_car_stage_ptr.restype = P
        lib.vdamd_car_set_max_blocks.argtypes = [P, I]
        lib.vdamd_car_set_max_blocks.restype = None
        lib.vdamd_car_set_timeout_ms.argtypes = [P, ctypes.c_long]
        lib.vdamd_car_set_timeout_ms.restype = None
        lib.vdamd_car_timed_out.argtypes = [P]
        lib.vdamd_car_timed_out.restype = I
        lib.vdamd_car_all_reduce.argtypes = [P, I, P, P, Z, I, I, I, P]
        lib.vdamd_car_all_reduce.restype = I
        lib.vdamd_car_fused_norm.argtypes = [P, I, P, P, P, P, F, I, I, I, I, I, P]
        lib.vdamd_car_fused_norm.restype = I
        lib.vdamd_car_all_gather.argtypes = [P, I, P, P, Z, P]
        lib.vdamd_car_all_gather.restype = I
        lib.vdamd_car_barrier.argtypes = [P, P]
        lib.vdamd_car_barrier.restype = I
        lib.vdamd_car_destroy.argtypes = [P]
        lib.vdamd_car_destroy.restype = None

    # ------------------------------------------------------------- bookkeeping
    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def _advance(self) -> int:
        p = self.parity
        self.parity ^= 1
        self.calls += 1
        return p

    def end_step(self) -> None:
        """Close a forward pass at parity 0 (see the module docstring)."""
        if self.parity:
            self._advance()
            rc = self.lib.vdamd_car_barrier(self.ctx, self._stream())
            if rc != 0:
                raise RuntimeError(f"direct collective barrier failed with {rc}")

    def fits(self, nbytes: int) -> bool:
        return not self.disabled and 0 < nbytes <= self.max_bytes and nbytes % 16 == 0

    def should_use(self, x: torch.Tensor) -> bool:
        nbytes = x.numel() * x.element_size()
        return (x.is_cuda and x.dtype in _DT and x.is_contiguous() and x.data_ptr() % 16 == 0
                and self.fits(nbytes) and nbytes <= self.use_bytes)

    def stage_view(self, shape: Tuple[int, ...], dtype: torch.dtype) -> Optional[torch.Tensor]:
        """A tensor of ``shape`` inside the stage area the NEXT call reads: a producer that
        writes its partial sums here saves the copy.  None if it does not fit."""
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if dtype not in _DT or not self.fits(nbytes):
            return None
        return self._stage[self.parity][:nbytes].view(dtype).view(*shape)

    def _in_ptr(self, x: torch.Tensor, parity: int) -> Optional[int]:
        return None if x.data_ptr() == self._stage[parity].data_ptr() else x.data_ptr()

    # -------------------------------------------------------------- collectives
    def all_reduce(self, x: torch.Tensor, two_shot: Optional[bool] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        nbytes = x.numel() * x.element_size()
        if two_shot is None:
            two_shot = nbytes > self.one_shot_bytes
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device) if out is None else out
        p = self._advance()
        rc = self.lib.vdamd_car_all_reduce(self.ctx, p, self._in_ptr(x, p), out.data_ptr(), nbytes, _DT[x.dtype],
                                           int(two_shot), int(self.fence_start), self._stream())
        if rc != 0:
            raise RuntimeError(f"direct all-reduce launch failed with {rc}")
        return out

    def maybe_all_reduce(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        return self.all_reduce(x) if self.should_use(x) else None

    def shard_rows(self, rows: int) -> Tuple[int, int]:
        """(rows per rank, rows this rank owns) of the two-shot row split."""
        per = -(-rows // self.world)
        lo = min(self.rank * per, rows)
        return per, min(lo + per, rows) - lo

    def all_reduce_add_rms_norm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                sharded_residual: bool) -> torch.Tensor:
        """``normed = RMSNorm(sum_ranks(x) + residual) * weight``; ``residual`` is updated in place
        to the sum.  ``sharded_residual=False``: ``residual`` is ``[rows, H]``, replicated (one-shot).
        ``True``: ``residual`` is this rank's ``[ceil(rows/world), H]`` row shard (two-shot).
        ``normed`` is the full ``[rows, H]`` on every rank either way."""
        rows, h = x.shape
        out = torch.empty((rows, h), dtype=x.dtype, device=x.device)
        p = self._advance()
        rc = self.lib.vdamd_car_fused_norm(self.ctx, p, self._in_ptr(x, p), out.data_ptr(), residual.data_ptr(),
                                           weight.data_ptr(), float(eps), rows, h, _DT[x.dtype],
                                           int(sharded_residual), int(self.fence_start), self._stream())
        if rc != 0:
            raise RuntimeError(f"fused all-reduce + RMSNorm launch failed with {rc}")
        return out

    def all_gather(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Concatenation of every rank's ``x`` along dim 0 (any dtype; 16-byte multiple per rank)."""
        nbytes = x.numel() * x.element_size()
        if out is None:
            out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        p = self._advance()
        rc = self.lib.vdamd_car_all_gather(self.ctx, p, x.data_ptr(), out.data_ptr(), nbytes, self._stream())
        if rc != 0:
            raise RuntimeError(f"direct all-gather launch failed with {rc}")
        return out

    def can_gather(self, x: torch.Tensor) -> bool:
        return (x.is_cuda and x.is_contiguous() and x.data_ptr() % 16 == 0
                and self.fits(x.numel() * x.element_size()))

    def timed_out(self) -> bool:
        """Whether a peer failed to arrive at a barrier within the bound (the sums of that call
        are then wrong).  A read of pinned host memory: cheap enough to ask after every step."""
        return bool(self.ctx) and bool(self.lib.vdamd_car_timed_out(self.ctx))

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self._stage = []
            self.lib.vdamd_car_destroy(self.ctx)
            self.ctx = None
