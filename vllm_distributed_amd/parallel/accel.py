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


def _verify_all_reduce(group: GroupCoordinator, fn, sizes=(4096, 1 << 19, 3 << 20)) -> bool:
    """Sums of several message sizes (they select different algorithms: the
    direct all-reduce is one-shot up to 512 KiB and two-shot above) against
    ``torch.distributed``."""
    try:
        for n in sizes:
            x = (torch.arange(n, dtype=torch.float32, device=group.device) % 251) * (group.rank_in_group + 1)
            x = (x / 64).to(torch.bfloat16)
            ref = x.clone()
            group._td_all_reduce(ref)
            out = fn(x.clone())
            torch.cuda.synchronize()
            if out is None or not torch.allclose(out.float(), ref.float(), rtol=2e-2, atol=1e-1):
                return False
        return True
    except Exception as e:  # noqa: BLE001
        logger.warning("collective self-check raised %r", e)
        return False


def _verify_gather_scatter(group: GroupCoordinator, comm) -> bool:
    """all-gather and reduce-scatter of the direct binding against their definitions."""
    try:
        w, r = group.world_size, group.rank_in_group
        x = torch.full((256, 64), float(r + 1), dtype=torch.bfloat16, device=group.device)
        out = torch.empty((w * 256, 64), dtype=torch.bfloat16, device=group.device)
        comm.all_gather(out, x)
        want = torch.arange(1, w + 1, device=group.device, dtype=torch.float32).repeat_interleave(256)
        y = (torch.arange(w * 128, device=group.device, dtype=torch.float32) % 16).view(-1, 1).expand(-1, 64)
        y = (y * (r + 1)).to(torch.bfloat16).contiguous()
        rs = torch.empty((128, 64), dtype=torch.bfloat16, device=group.device)
        comm.reduce_scatter(rs, y)
        torch.cuda.synchronize()
        scale = w * (w + 1) / 2
        want_rs = (torch.arange(r * 128, (r + 1) * 128, device=group.device, dtype=torch.float32) % 16) * scale
        return bool(torch.equal(out[:, 0].float(), want)
                    and torch.allclose(rs[:, 0].float(), want_rs, rtol=2e-2, atol=1e-1))
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
                group, lambda t: (comm.all_reduce(t), t)[1]) and _verify_gather_scatter(group, comm)
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
            ok = car is not None and _verify_all_reduce(
                group, car.all_reduce, [n for n in (4096, 1 << 19, 3 << 20) if 2 * n <= car.max_bytes])
            if _all_agree(group, ok):
                group.custom_ar = car
                logger.info("tp group: direct (peer-mapped) all-reduce active")
            elif car is not None:
                car.close()


def reverify_after_failed_capture(group: GroupCoordinator) -> None:
    """A hipGraph capture that aborted may have left a communicator half-way through an
    enqueue.  All ranks of the group re-run the self-checks together and drop (together)
    whichever accelerated path no longer gives the right sum; a path that still works
    keeps carrying the eager steps."""
    if group.custom_ar is not None:
        car = group.custom_ar
        ok = _verify_all_reduce(group, car.all_reduce,
                                [n for n in (4096, 1 << 19) if 2 * n <= car.max_bytes]) and not car.timed_out()
        if not _all_agree(group, ok):
            logger.warning("%s group: direct all-reduce failed its re-check after an aborted capture; dropped",
                           group.name)
            group.custom_ar = None
            car.close()
    if group.rccl is not None:
        comm = group.rccl
        ok = _verify_all_reduce(group, lambda t: (comm.all_reduce(t), t)[1])
        if not _all_agree(group, ok):
            logger.warning("%s group: direct RCCL communicator failed its re-check after an aborted capture; "
                           "torch.distributed carries the traffic", group.name)
            group.rccl = None
