"""Process groups for the data plane.

Rank layout is ``[DP][PP][TP]`` with TP fastest (reference
``src/launch.py:215,234,278``: ``rank = pp_rank * TP + tp_rank``).  A TP group
is the consecutive ranks of one pipeline stage; a PP group strides by TP.

Collectives go through ``GroupCoordinator`` so the model code never sees which
transport carries them: gloo on CPU, RCCL over xGMI on GPU (through
``torch.distributed`` or the direct ctypes binding in ``rccl.py``), and the
peer-mapped direct all-reduce in ``custom_all_reduce.py`` for small messages.
A group of size 1 short-circuits every call.

The rendezvous is a ``tcp://head:port`` store address generated on the head
and shipped to every worker in its ``init_worker`` kwargs
(reference ``src/launch.py:94,279-288``).
"""
from __future__ import annotations

import os
from datetime import timedelta
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .. import envs
from ..logger import init_logger
from ..utils import tracing

logger = init_logger(__name__)


class CollectiveOrderChecker:
    """Debug aid: every collective bumps a per-group sequence number and tags
    it with an op code; ``verify`` all-gathers (seq, op, numel) over the CPU
    group and raises when ranks disagree - mis-ordered collectives are the
    "data race" of this design."""

    def __init__(self) -> None:
        self.seq = 0
        self.log: List[Tuple[int, str, int]] = []

    def record(self, op: str, numel: int) -> None:
        self.seq += 1
        self.log.append((self.seq, op, numel))
        if len(self.log) > 4096:
            del self.log[:2048]

    def verify(self, group: "GroupCoordinator") -> None:
        if group.world_size == 1 or group.cpu_group is None:
            return
        mine = self.log[-1] if self.log else (0, "", 0)
        gathered: List[Any] = [None] * group.world_size
        dist.all_gather_object(gathered, mine, group=group.cpu_group)
        if any(g != gathered[0] for g in gathered):
            raise RuntimeError(f"collective order mismatch in group {group.name}: {gathered}")


class CommTimer:
    """Time spent in collectives per step (``VDAMD_TIME_COLLECTIVES=1``): GPU
    collectives are bracketed with events on the stream they run on, host
    (gloo) collectives with the wall clock.  Collectives on a side stream that
    overlap compute are counted at their own duration."""
    enabled = False
    _events: List[Tuple["torch.cuda.Event", "torch.cuda.Event"]] = []
    _host_seconds = 0.0

    @classmethod
    def configure(cls) -> None:
        cls.enabled = bool(envs.VDAMD_TIME_COLLECTIVES)

    @classmethod
    def drain(cls) -> float:
        """Seconds spent in the collectives that have completed since the last
        call.  Never waits: events still in flight are counted by a later call."""
        total = cls._host_seconds
        cls._host_seconds = 0.0
        try:
            done = 0
            for a, b in cls._events:
                if not b.query():
                    break
                total += a.elapsed_time(b) * 1e-3
                done += 1
            del cls._events[:done]
        except RuntimeError:      # a measurement must never take a step down
            cls._events = []
        return total


def _timed(fn):
    """Collective wrapper feeding ``CommTimer`` (a no-op unless it is enabled)."""
    import functools
    import time as _time

    @functools.wraps(fn)
    def wrapper(self, x, *args, **kwargs):
        if not CommTimer.enabled or self.world_size == 1:
            return fn(self, x, *args, **kwargs)
        on_gpu = isinstance(x, torch.Tensor) and x.is_cuda and not torch.cuda.is_current_stream_capturing()
        if on_gpu:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(self, x, *args, **kwargs)
            b.record()
            CommTimer._events.append((a, b))
            return out
        t0 = _time.perf_counter()
        out = fn(self, x, *args, **kwargs)
        CommTimer._host_seconds += _time.perf_counter() - t0
        return out
    return wrapper


class GroupCoordinator:
    def __init__(self, name: str, ranks: Sequence[int], global_rank: int,
                 device_group: Optional[Any], cpu_group: Optional[Any],
                 device: torch.device):
        self.name = name
        self.ranks = list(ranks)
        self.world_size = len(self.ranks)
        self.rank = global_rank
        self.rank_in_group = self.ranks.index(global_rank)
        self.device_group = device_group
        self.cpu_group = cpu_group
        self.device = device
        self.checker = CollectiveOrderChecker() if envs.VDAMD_CHECK_COLLECTIVE_ORDER else None
        # optional accelerated paths, attached by the worker on GPU
        self.rccl = None              # parallel.rccl.RcclCommunicator
        self.custom_ar = None         # parallel.custom_all_reduce.CustomAllReduce
        # gloo carrying GPU tensors: stage through the host (gloo itself only
        # implements a few collectives for device tensors)
        self.staged = (device.type == "cuda" and device_group is not None
                       and dist.get_backend(device_group) == "gloo")

    # ---------------------------------------------------------------- helpers
    @property
    def first_rank(self) -> int:
        return self.ranks[0]

    @property
    def last_rank(self) -> int:
        return self.ranks[-1]

    @property
    def is_first_rank(self) -> bool:
        return self.rank_in_group == 0

    @property
    def is_last_rank(self) -> bool:
        return self.rank_in_group == self.world_size - 1

    @property
    def next_rank(self) -> int:
        return self.ranks[(self.rank_in_group + 1) % self.world_size]

    @property
    def prev_rank(self) -> int:
        return self.ranks[(self.rank_in_group - 1) % self.world_size]

    def _note(self, op: str, t: torch.Tensor) -> None:
        if tracing.enabled():
            tracing.mark(f"{self.name}.{op}[{t.numel() * t.element_size()}B]")
        if self.checker is not None:
            self.checker.record(op, t.numel())
            self.checker.verify(self)

    # torch.distributed calls, host-staged when gloo carries device tensors
    def _td_all_reduce(self, x: torch.Tensor) -> None:
        if self.staged and x.is_cuda:
            h = x.cpu()
            dist.all_reduce(h, group=self.device_group)
            x.copy_(h)
        else:
            dist.all_reduce(x, group=self.device_group)

    def _td_all_gather(self, flat: torch.Tensor, x: torch.Tensor) -> None:
        if self.staged and x.is_cuda:
            h = torch.empty(flat.shape, dtype=flat.dtype)
            dist.all_gather_into_tensor(h, x.cpu(), group=self.device_group)
            flat.copy_(h)
        else:
            dist.all_gather_into_tensor(flat, x, group=self.device_group)

    def _td_broadcast(self, x: torch.Tensor, src_global: int) -> None:
        if self.staged and x.is_cuda:
            h = x.cpu()
            dist.broadcast(h, src=src_global, group=self.device_group)
            x.copy_(h)
        else:
            dist.broadcast(x, src=src_global, group=self.device_group)

    # ------------------------------------------------------------ collectives
    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """Sum over the group.  May return a new tensor (direct all-reduce) or
        ``x`` reduced in place."""
        if self.world_size == 1:
            return x
        self._note("all_reduce", x)
        if self.custom_ar is not None:
            out = self.custom_ar.maybe_all_reduce(x)
            if out is not None:
                return out
        if self.rccl is not None:
            self.rccl.all_reduce(x)
            return x
        self._td_all_reduce(x)
        return x

    def fused_all_reduce_norm_ok(self, rows: int, hidden: int, dtype: torch.dtype) -> bool:
        """Whether ``all_reduce_add_rms_norm`` takes a ``[rows, hidden]`` message (peer-mapped path)."""
        car = self.custom_ar
        return (car is not None and envs.VDAMD_FUSED_ALL_REDUCE_NORM and dtype in (torch.bfloat16, torch.float16)
                and hidden % 8 == 0 and hidden <= 16384 and car.fits(rows * hidden * 2)
                and rows * hidden * 2 <= car.use_bytes)

    def all_reduce_add_rms_norm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                sharded_residual: bool) -> torch.Tensor:
        """One kernel for all-reduce + residual add + RMSNorm (see ``CustomAllReduce``)."""
        self._note("all_reduce_norm", x)
        return self.custom_ar.all_reduce_add_rms_norm(x, residual, weight, eps, sharded_residual)

    def stage_view(self, shape, dtype: torch.dtype) -> Optional[torch.Tensor]:
        """Where the producer of the next all-reduce input should write (None: anywhere)."""
        return None if self.custom_ar is None else self.custom_ar.stage_view(tuple(shape), dtype)

    def end_step(self) -> None:
        if self.custom_ar is not None:
            self.custom_ar.end_step()

    def graph_capturable(self) -> bool:
        """Whether this group's collectives are plain stream enqueues a hipGraph capture can
        record: the direct RCCL binding, or the peer-mapped kernels alone (then every message
        of the captured step has to fit them)."""
        return self.world_size == 1 or self.rccl is not None or self.custom_ar is not None

    def all_reduce_(self, x: torch.Tensor) -> torch.Tensor:
        """Strictly in-place sum on the current stream (RCCL / gloo only)."""
        if self.world_size == 1:
            return x
        self._note("all_reduce", x)
        if self.rccl is not None:
            self.rccl.all_reduce(x)
        else:
            self._td_all_reduce(x)
        return x

    def all_gather(self, x: torch.Tensor, dim: int = -1) -> torch.Tensor:
        if self.world_size == 1:
            return x
        self._note("all_gather", x)
        if dim < 0:
            dim += x.dim()
        x = x.contiguous()
        if x.dim() == 0:
            x = x.reshape(1)
        # [world * n0, ...]: ranks concatenated along dim 0 (the layout both
        # RCCL and gloo produce), then moved to the requested dim
        flat = torch.empty((self.world_size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype,
                           device=x.device)
        if self.custom_ar is not None and self.custom_ar.can_gather(x):
            self.custom_ar.all_gather(x, out=flat)
        elif self.rccl is not None:
            self.rccl.all_gather(flat, x)
        else:
            self._td_all_gather(flat, x)
        if dim == 0:
            return flat
        out = flat.view((self.world_size,) + tuple(x.shape)).movedim(0, dim)
        shape = list(x.shape)
        shape[dim] *= self.world_size
        return out.reshape(shape)

    def all_gather_into(self, out: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """Concatenate every rank's ``x`` along dim 0 into the caller's ``out``."""
        if self.world_size == 1:
            out.copy_(x)
            return out
        self._note("all_gather", x)
        if self.custom_ar is not None and self.custom_ar.can_gather(x) and out.is_contiguous():
            self.custom_ar.all_gather(x, out=out)
        elif self.rccl is not None:
            self.rccl.all_gather(out, x)
        else:
            self._td_all_gather(out, x)
        return out

    def reduce_scatter(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sum over the group, keeping rows ``[r*n, (r+1)*n)`` of dim 0 on rank
        ``r`` (``n = x.shape[0] / world``).  gloo has no reduce-scatter: there
        the sum is an all-reduce and the slice is copied out."""
        if self.world_size == 1:
            if out is None:
                return x
            out.copy_(x)
            return out
        assert x.shape[0] % self.world_size == 0 and x.is_contiguous()
        n = x.shape[0] // self.world_size
        if out is None:
            out = torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        self._note("reduce_scatter", x)
        if self.rccl is not None:
            self.rccl.reduce_scatter(out, x)
        elif dist.get_backend(self.device_group) == "nccl":
            dist.reduce_scatter_tensor(out, x, group=self.device_group)
        else:
            full = x.clone()
            self._td_all_reduce(full)
            out.copy_(full[self.rank_in_group * n:(self.rank_in_group + 1) * n])
        return out

    def broadcast(self, x: torch.Tensor, src: int = 0) -> torch.Tensor:
        """``src`` is a rank *within the group*."""
        if self.world_size == 1:
            return x
        self._note("broadcast", x)
        if self.rccl is not None:
            self.rccl.broadcast(x, src)
        else:
            self._td_broadcast(x, self.ranks[src])
        return x

    def send(self, x: torch.Tensor, dst: Optional[int] = None) -> None:
        """Point-to-point to a rank within the group (default: next)."""
        dst = (self.rank_in_group + 1) % self.world_size if dst is None else dst
        self._note("send", x)
        if self.rccl is not None:
            self.rccl.send(x.contiguous(), dst)
        else:
            x = x.cpu() if self.staged else x.contiguous()
            dist.send(x, dst=self.ranks[dst], group=self.device_group)

    def recv(self, shape: Sequence[int], dtype: torch.dtype, src: Optional[int] = None) -> torch.Tensor:
        src = (self.rank_in_group - 1) % self.world_size if src is None else src
        x = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        self._note("recv", x)
        if self.rccl is not None:
            self.rccl.recv(x, src)
        elif self.staged and x.is_cuda:
            h = torch.empty(x.shape, dtype=dtype)
            dist.recv(h, src=self.ranks[src], group=self.device_group)
            x.copy_(h)
        else:
            dist.recv(x, src=self.ranks[src], group=self.device_group)
        return x

    all_reduce, all_reduce_, all_gather = _timed(all_reduce), _timed(all_reduce_), _timed(all_gather)
    reduce_scatter, broadcast, send = _timed(reduce_scatter), _timed(broadcast), _timed(send)

    def bootstrap_all_gather(self, obj: Any, tag: str) -> List[Any]:
        """``all_gather_object`` for data-plane bring-up values (RCCL unique ids, IPC handles): over
        the control plane when this process has one, else over the group's CPU channel."""
        if self.world_size == 1:
            return [obj]
        from . import bootstrap
        if bootstrap.available():
            scope = f"{self.name}:{','.join(map(str, self.ranks))}:{tag}"
            return bootstrap.exchange(scope, self.rank_in_group, self.world_size, obj)
        return self.all_gather_object(obj)

    def broadcast_object(self, obj: Any = None, src: int = 0) -> Any:
        if self.world_size == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=self.ranks[src], group=self.cpu_group)
        return box[0]

    def all_gather_object(self, obj: Any) -> List[Any]:
        if self.world_size == 1:
            return [obj]
        out: List[Any] = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.cpu_group)
        return out

    def barrier(self) -> None:
        if self.world_size > 1:
            dist.barrier(group=self.cpu_group)


class SimulatedShardGroup(GroupCoordinator):
    """One rank standing in for a TP group of ``size`` ranks, for measuring a
    rank's compute at a TP degree on a single GPU (``VDAMD_SIMULATE_TP``).
    Layers shard their weights for rank 0 of ``size``; collectives move no
    data between ranks and only produce tensors of the right shape
    (all-reduce: identity; all-gather: ``size`` copies; reduce-scatter: the
    first ``1/size`` rows), so outputs are *not* the model's outputs."""

    def __init__(self, name: str, size: int, device: torch.device):
        super().__init__(name, [0], 0, None, None, device)
        self.world_size = size

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        return x

    all_reduce_ = all_reduce

    def all_gather(self, x: torch.Tensor, dim: int = -1) -> torch.Tensor:
        return torch.cat([x] * self.world_size, dim=dim)

    def all_gather_into(self, out: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        out.view((self.world_size,) + tuple(x.shape)).copy_(x.unsqueeze(0).expand((self.world_size,) + tuple(x.shape)))
        return out

    def reduce_scatter(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = x.shape[0] // self.world_size
        if out is None:
            return x[:n].clone()
        out.copy_(x[:n])
        return out

    def broadcast(self, x: torch.Tensor, src: int = 0) -> torch.Tensor:
        return x

    def broadcast_object(self, obj: Any = None, src: int = 0) -> Any:
        return obj

    def all_gather_object(self, obj: Any) -> List[Any]:
        return [obj] * self.world_size

    def barrier(self) -> None:
        return None


# ------------------------------------------------------------------ globals
_WORLD: Optional[GroupCoordinator] = None
_TP: Optional[GroupCoordinator] = None
_PP: Optional[GroupCoordinator] = None
_OWNS_PROCESS_GROUP = False


def _single(name: str, device: torch.device) -> GroupCoordinator:
    sim = envs.VDAMD_SIMULATE_TP
    if name == "tp" and sim > 1:
        return SimulatedShardGroup(name, sim, device)
    return GroupCoordinator(name, [0], 0, None, None, device)


def init_distributed_environment(world_size: int, rank: int, init_method: Optional[str],
                                 local_rank: int, backend: str, device: torch.device,
                                 timeout_s: float = 1800.0, force: bool = False) -> None:
    """Join the world.  ``backend`` is ``nccl`` (= RCCL on ROCm) or ``gloo``.
    A world of one skips process-group creation unless ``force`` is set."""
    global _WORLD, _OWNS_PROCESS_GROUP
    if world_size == 1 and not force:
        _WORLD = _single("world", device)
        return
    if not dist.is_initialized():
        kwargs: Dict[str, Any] = dict(backend=backend, world_size=world_size, rank=rank,
                                      timeout=timedelta(seconds=timeout_s))
        if init_method is not None and init_method != "env://":
            kwargs["init_method"] = init_method
        if backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(**kwargs)
        _OWNS_PROCESS_GROUP = True
    ranks = list(range(world_size))
    cpu_group = dist.new_group(ranks, backend="gloo")
    _WORLD = GroupCoordinator("world", ranks, rank, dist.group.WORLD, cpu_group, device)


_GENERATION = 0


def initialize_model_parallel(tp_size: int, pp_size: int) -> None:
    """Build TP and PP groups.  Every rank must call this with the same sizes
    (``new_group`` is collective over the world)."""
    global _TP, _PP, _GENERATION
    _GENERATION += 1
    world = get_world_group()
    ws, rank, device = world.world_size, world.rank, world.device
    if ws == 1:
        _TP, _PP = _single("tp", device), _single("pp", device)
        return
    assert ws % (tp_size * pp_size) == 0, (ws, tp_size, pp_size)
    dp_size = ws // (tp_size * pp_size)
    backend = dist.get_backend()
    _TP = _PP = None
    for dp in range(dp_size):
        for pp in range(pp_size):
            base = (dp * pp_size + pp) * tp_size
            ranks = list(range(base, base + tp_size))
            dev = dist.new_group(ranks, backend=backend) if tp_size > 1 else None
            cpu = dist.new_group(ranks, backend="gloo") if tp_size > 1 else None
            if rank in ranks:
                _TP = GroupCoordinator("tp", ranks, rank, dev, cpu, device)
    for dp in range(dp_size):
        for tp in range(tp_size):
            ranks = [(dp * pp_size + pp) * tp_size + tp for pp in range(pp_size)]
            dev = dist.new_group(ranks, backend=backend) if pp_size > 1 else None
            cpu = dist.new_group(ranks, backend="gloo") if pp_size > 1 else None
            if rank in ranks:
                _PP = GroupCoordinator("pp", ranks, rank, dev, cpu, device)
    assert _TP is not None and _PP is not None


def get_world_group() -> GroupCoordinator:
    assert _WORLD is not None, "distributed environment is not initialised"
    return _WORLD


def get_tp_group() -> GroupCoordinator:
    assert _TP is not None, "model parallel groups are not initialised"
    return _TP


def get_pp_group() -> GroupCoordinator:
    assert _PP is not None, "model parallel groups are not initialised"
    return _PP


def model_parallel_is_initialized() -> bool:
    return _TP is not None and _PP is not None


def get_tensor_model_parallel_world_size() -> int:
    return _TP.world_size if _TP is not None else 1


def get_tensor_model_parallel_rank() -> int:
    return _TP.rank_in_group if _TP is not None else 0


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    return x if _TP is None else _TP.all_reduce(x)


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    return x if _TP is None else _TP.all_gather(x, dim)


def ensure_single_process_groups(device: Optional[torch.device] = None) -> None:
    """World/TP/PP of size 1 for single-process use (tests, TP=PP=1)."""
    global _WORLD, _TP, _PP
    device = device or torch.device("cpu")
    if _WORLD is None:
        _WORLD = _single("world", device)
    if _TP is None:
        _TP = _single("tp", device)
    if _PP is None:
        _PP = _single("pp", device)


def current_generation() -> int:
    """Counts ``initialize_model_parallel`` calls: a worker remembers the value it set the groups up
    under, so that a stale worker being finalised late cannot tear down its successor's groups."""
    return _GENERATION


def destroy_model_parallel(generation: Optional[int] = None) -> None:
    """``generation``: only act if the groups are still the ones set up under that generation."""
    global _WORLD, _TP, _PP, _OWNS_PROCESS_GROUP
    if generation is not None and generation != _GENERATION:
        return
    for g in (_TP, _PP):
        if g is not None:
            if g.custom_ar is not None:
                g.custom_ar.close()
            if g.rccl is not None:
                g.rccl.close()
    _TP = _PP = _WORLD = None
    if _OWNS_PROCESS_GROUP and dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:  # pragma: no cover
            pass
    _OWNS_PROCESS_GROUP = False
