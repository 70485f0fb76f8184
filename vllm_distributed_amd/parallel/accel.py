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


def _exact(n: int, rank: int, salt: int, device, dtype=torch.bfloat16) -> torch.Tensor:
    """Small integers (exact in bf16, sums over 8 ranks below 256) that differ by rank and by call."""
    x = (torch.arange(n, device=device, dtype=torch.int64) * 7 + salt) % 13 + rank + 1 + salt % 5
    return x.to(dtype)


def _verify_direct(group: GroupCoordinator, car, rounds: int = 12) -> bool:
    """Every form of the peer-mapped path on exact-integer payloads, back to back and with data
    that changes from call to call: a reader that sees a peer's previous message, or a buffer
    overwritten while a peer still reads it, gives a sum that differs by at least one and cannot
    hide inside a tolerance.  One-shot and two-shot all-reduce from the stage area and from ordinary
    memory, the fused all-reduce + add + RMSNorm with replicated and with row-sharded residual
    (residual compared exactly, normalised rows to bf16 rounding), all-gather.  The expected sums
    are computed locally (every rank can form every rank's payload)."""
    try:
        dev, w, r = group.device, group.world_size, group.rank_in_group
        sizes = [n for n in (1024, 24 * 1024, 160 * 1024, 1 << 20) if 2 * n <= car.max_bytes]
        ok = True            # never leave early: every rank must issue the same sequence of collectives
        for it in range(rounds):
            n = sizes[it % len(sizes)]
            parts = [_exact(n, q, it, dev) for q in range(w)]
            ref = torch.stack([p.float() for p in parts]).sum(0).to(torch.bfloat16)
            staged = car.stage_view((n,), torch.bfloat16)
            src = parts[r]
            if it % 2 == 0 and staged is not None:
                staged.copy_(parts[r])
                src = staged
            out = car.all_reduce(src, two_shot=bool((it // 2) % 2))
            if not torch.equal(out, ref):
                logger.warning("direct all-reduce self-check: wrong sum (round %d, %d elements)", it, n)
                ok = False
            # fused add + RMSNorm on [rows, 512]
            rows, h = (5, 512) if it % 3 else (64, 512)
            xs = [_exact(rows * h, q, it + 17, dev).view(rows, h) for q in range(w)]
            res = _exact(rows * h, 0, it + 3, dev).view(rows, h)
            wt = (_exact(h, 0, it, dev).float() / 8).to(torch.bfloat16)
            want_res = (torch.stack([x.float() for x in xs]).sum(0) + res.float()).to(torch.bfloat16)
            wf = want_res.float()
            want = wf * torch.rsqrt(wf.pow(2).mean(-1, keepdim=True) + 1e-5) * wt.float()
            sharded = bool(it % 2)
            per = -(-rows // w)
            lo = min(r * per, rows)
            my_res = res[lo:min(lo + per, rows)].clone() if sharded else res.clone()
            if sharded and my_res.shape[0] < per:
                my_res = torch.cat([my_res, torch.zeros(per - my_res.shape[0], h, dtype=res.dtype, device=dev)])
            got = car.all_reduce_add_rms_norm(xs[r], my_res, wt, 1e-5, sharded)
            got_res = my_res[:min(lo + per, rows) - lo] if sharded else my_res
            want_r = want_res[lo:min(lo + per, rows)] if sharded else want_res
            if not (torch.allclose(got.float(), want, rtol=2e-2, atol=2e-2) and torch.equal(got_res, want_r)):
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


def _calibrate_direct(group: GroupCoordinator, car) -> None:
    """Size thresholds of the peer-mapped path from this machine's own links: both forms of the
    all-reduce (and RCCL's, where the direct binding is up) are timed at 32 KiB .. the path's
    capacity; ``one_shot_bytes`` becomes the largest size up to which one pass over every peer's
    partial sums is not slower than reduce-scatter + gather, ``use_bytes`` the largest size up to
    which the path is not slower than RCCL.  Every rank issues the same calls and the decision is
    taken on the slowest rank's times, so all ranks agree."""
    if not getattr(car, "auto_thresholds", False):
        return
    try:
        dev, w = group.device, group.world_size
        sizes = [s for s in (32 << 10, 64 << 10, 128 << 10, 256 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20,
                             8 << 20, 16 << 20, 32 << 20, 64 << 20) if s <= car.max_bytes]

        def timed(fn, iters: int) -> float:
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            b.synchronize()
            return a.elapsed_time(b) * 1e3 / iters

        rows = []
        for s in sizes:
            n, iters = s // 2, (16 if s <= (1 << 20) else 6)
            x = torch.zeros(n, dtype=torch.bfloat16, device=dev)
            one = timed(lambda: car.all_reduce(car.stage_view((n,), torch.bfloat16), two_shot=False), iters) \
                if s * w <= (64 << 20) else float("inf")
            two = timed(lambda: car.all_reduce(car.stage_view((n,), torch.bfloat16), two_shot=True), iters)
            lib = timed(lambda: group.rccl.all_reduce(x), iters) if group.rccl is not None else float("inf")
            rows.append((one, two, lib))
        car.end_step()
        torch.cuda.synchronize()
        worst = [tuple(max(col) for col in zip(*per_size)) for per_size in zip(*group.all_gather_object(rows))]
        one_shot, use, prefix_one, prefix_use = 0, 0, True, True
        for s, (one, two, lib) in zip(sizes, worst):
            if prefix_one and one <= two * 1.05:
                one_shot = s
            else:
                prefix_one = False
            if prefix_use and min(one, two) <= lib:
                use = s
            else:
                prefix_use = False
        # (below half the smallest size timed there is no measurement, and a noisy first row must not send
        # few-kilobyte decode messages through the two-barrier form)
        car.one_shot_bytes = max(one_shot, 16 << 10)
        car.use_bytes = max(use, 16 << 10) if group.rccl is not None else car.max_bytes
        car.calibration = [dict(bytes=s, one_shot_us=round(o, 1) if o != float("inf") else None, two_shot_us=round(t, 1),
                                rccl_us=round(l, 1) if l != float("inf") else None)
                           for s, (o, t, l) in zip(sizes, worst)]
        if group.rank_in_group == 0:
            logger.info("tp group: direct collectives calibrated: one-shot up to %d KiB, used up to %d KiB %s",
                        car.one_shot_bytes >> 10, car.use_bytes >> 10, car.calibration)
    except Exception as e:  # noqa: BLE001 - the defaults stay
        logger.warning("direct collective calibration raised %r; keeping the default thresholds", e)


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
                _calibrate_direct(group, car)
                ok = _all_agree(group, not car.timed_out())
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
