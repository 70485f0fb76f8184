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


def _verify_direct(group: GroupCoordinator, car, rounds: int = 12) -> bool:
    """Every form of the peer-mapped path against ``torch.distributed``, back to back and with
    data that changes from call to call (a reader that sees a peer's previous message, or a
    buffer overwritten while a peer still reads it, shows up as a wrong sum): one-shot and
    two-shot all-reduce from the stage area and from ordinary memory, the fused
    all-reduce + add + RMSNorm with replicated and with row-sharded residual, all-gather."""
    try:
        dev, w, r = group.device, group.world_size, group.rank_in_group
        gen = torch.Generator(device="cpu").manual_seed(1234)            # same stream of shapes on all ranks
        sizes = [n for n in (1024, 24 * 1024, 160 * 1024, 1 << 20) if 2 * n <= car.max_bytes]
        ok = True            # never leave early: every rank must issue the same sequence of collectives
        for it in range(rounds):
            n = sizes[it % len(sizes)]
            base = torch.randn(n, generator=gen).to(dev)
            x = (base * (r + 1) + it).to(torch.bfloat16)
            ref = x.clone()
            group._td_all_reduce(ref)
            staged = car.stage_view((n,), torch.bfloat16)
            src = x
            if it % 2 == 0 and staged is not None:
                staged.copy_(x)
                src = staged
            out = car.all_reduce(src, two_shot=bool((it // 2) % 2))
            if not torch.allclose(out.float(), ref.float(), rtol=2e-2, atol=2e-1):
                logger.warning("direct all-reduce self-check: wrong sum (round %d, %d elements)", it, n)
                ok = False
            # fused add + RMSNorm on [rows, 512]
            rows, h = (5, 512) if it % 3 else (64, 512)
            xs = (torch.randn(rows, h, generator=gen).to(dev) * (r + 1)).to(torch.bfloat16)
            res = torch.randn(rows, h, generator=gen).to(dev).to(torch.bfloat16)
            wt = torch.randn(h, generator=gen).to(dev).to(torch.bfloat16)
            tot = xs.clone()
            group._td_all_reduce(tot)
            want_res = (tot.float() + res.float()).to(torch.bfloat16)
            wf = want_res.float()
            want = wf * torch.rsqrt(wf.pow(2).mean(-1, keepdim=True) + 1e-5) * wt.float()
            sharded = bool(it % 2)
            per = -(-rows // w)
            lo = min(r * per, rows)
            my_res = res[lo:min(lo + per, rows)].clone() if sharded else res.clone()
            if sharded and my_res.shape[0] < per:
                my_res = torch.cat([my_res, torch.zeros(per - my_res.shape[0], h, dtype=res.dtype, device=dev)])
            got = car.all_reduce_add_rms_norm(xs, my_res, wt, 1e-5, sharded)
            got_res = my_res[:min(lo + per, rows) - lo] if sharded else my_res
            want_r = want_res[lo:min(lo + per, rows)] if sharded else want_res
            if not (torch.allclose(got.float(), want, rtol=3e-2, atol=3e-2)
                    and torch.allclose(got_res.float(), want_r.float(), rtol=2e-2, atol=2e-1)):
                logger.warning("fused all-reduce + RMSNorm self-check failed (round %d, sharded=%s)", it, sharded)
                ok = False
            piece = torch.full((3, 8), float(r * 100 + it), dtype=torch.float32, device=dev)
            g = car.all_gather(piece)
            want_g = torch.cat([torch.full((3, 8), float(q * 100 + it), dtype=torch.float32, device=dev)
                                for q in range(w)])
            if not torch.equal(g, want_g):
                logger.warning("direct all-gather self-check failed (round %d)", it)
                ok = False
        car.end_step()
        torch.cuda.synchronize()
        return ok and not car.timed_out()
    except Exception as e:  # noqa: BLE001
        logger.warning("direct collective self-check raised %r", e)
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
        # ---- peer-mapped direct collectives for the messages of a decode step (TP only)
        if (group.name == "tp" and envs.VDAMD_CUSTOM_ALL_REDUCE
                and not pc.disable_custom_all_reduce):
            car = None
            try:
                from .custom_all_reduce import CustomAllReduce
                per_dev = 1
                if envs.VDAMD_SHARE_DEVICE:
                    per_dev = -(-group.world_size // max(1, torch.cuda.device_count()))
                car = CustomAllReduce(group, max_bytes=envs.VDAMD_CUSTOM_ALL_REDUCE_MAX_BYTES,
                                      ranks_per_device=per_dev)
            except Exception as e:  # noqa: BLE001
                logger.warning("direct all-reduce unavailable: %r", e)
            ok = _all_agree(group, car is not None and _verify_direct(group, car))
            auto = str(envs.VDAMD_CUSTOM_ALL_REDUCE_FENCE_START).strip().lower() == "auto"
            if not ok and auto and _all_agree(group, car is not None and not car.timed_out()):
                # once more with the release fence also in the opening barrier of every kernel
                car.fence_start = True
                ok = _all_agree(group, _verify_direct(group, car))
                if ok:
                    logger.warning("tp group: direct collectives need the start-barrier release fence on this system")
            if ok:
                group.custom_ar = car
                logger.info("tp group: direct (peer-mapped) collectives active (up to %d KiB, one-shot up to %d KiB)",
                            car.max_bytes >> 10, car.one_shot_bytes >> 10)
            elif car is not None:
                logger.warning("tp group: direct collectives failed their self-check; RCCL carries the traffic")
                car.close()

def reverify_after_failed_capture(group: GroupCoordinator) -> None:
    """A hipGraph capture that aborted may have left a communicator half-way through an
    enqueue.  All ranks of the group re-run the self-checks together and drop (together)
    whichever accelerated path no longer gives the right sum; a path that still works
    keeps carrying the eager steps."""
    if group.custom_ar is not None:
        car = group.custom_ar
        ok = _verify_direct(group, car, rounds=4)
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
