"""Attach the accelerated collective paths to the TP/PP groups.

Each path is verified against ``torch.distributed`` on a small tensor before
it is trusted; a path that fails to initialise, times out or returns a wrong
sum is dropped (on all ranks together) and ``torch.distributed`` over RCCL
keeps carrying that group's traffic.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import envs
from ..config import ParallelConfig
from ..logger import init_logger
from .state import GroupCoordinator, get_pp_group, get_tp_group

logger = init_logger(__name__)


def _all_agree(group: GroupCoordinator, ok: bool) -> bool:
    flags = group.all_gather_object(bool(ok))
    return all(flags)


def _verify_all_reduce(group: GroupCoordinator, fn) -> bool:
    try:
        x = torch.arange(4096, dtype=torch.float32, device=group.device) * (group.rank_in_group + 1)
        x = x.to(torch.bfloat16)
        ref = x.clone()
        group._td_all_reduce(ref)
        out = fn(x.clone())
        torch.cuda.synchronize()
        return out is not None and torch.allclose(out.float(), ref.float(), rtol=2e-2, atol=1e-1)
    except Exception as e:  # noqa: BLE001
        logger.warning("collective self-check raised %r", e)
        return False


def attach_accelerated_collectives(pc: ParallelConfig) -> None:
    backend = envs.VDAMD_COMM_BACKEND
    for group in (get_tp_group(), get_pp_group()):
        if group.world_size == 1:
            continue
        # ---- direct RCCL binding (graph-capturable plain kernel enqueues)
        if backend in ("auto", "rccl") and not group.staged:
            comm = None
            try:
                from .rccl import RcclCommunicator
                comm = RcclCommunicator(group)
            except Exception as e:  # noqa: BLE001
                logger.warning("RCCL direct binding unavailable for %s: %r", group.name, e)
            ok = comm is not None and _verify_all_reduce(
                group, lambda t: (comm.all_reduce(t), t)[1])
            if _all_agree(group, ok):
                group.rccl = comm
                logger.info("%s group: direct RCCL communicator active", group.name)
            elif comm is not None:
                comm.close()
        # ---- peer-mapped direct all-reduce for small messages (TP only)
        if (group.name == "tp" and envs.VDAMD_CUSTOM_ALL_REDUCE
                and not pc.disable_custom_all_reduce):
            car = None
            try:
                from .custom_all_reduce import CustomAllReduce
                car = CustomAllReduce(group, max_bytes=envs.VDAMD_CUSTOM_ALL_REDUCE_MAX_BYTES)
            except Exception as e:  # noqa: BLE001
                logger.warning("direct all-reduce unavailable: %r", e)
            ok = car is not None and _verify_all_reduce(group, car.all_reduce)
            if _all_agree(group, ok):
                group.custom_ar = car
                logger.info("tp group: direct (peer-mapped) all-reduce active")
            elif car is not None:
                car.close()
