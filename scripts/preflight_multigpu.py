#!/usr/bin/env python3
"""Preflight of the data plane on a multi-GPU node: latency and bandwidth of every collective the
workers use, per message size, over RCCL and over the peer-mapped (hipIpc) path, and the size
thresholds that follow from the numbers.

    python scripts/preflight_multigpu.py --gpus 8                  # one rank per GPU
    python scripts/preflight_multigpu.py --gpus 4 --share-device   # rehearsal: all ranks on GPU 0
    python scripts/preflight_multigpu.py --gpus 2 --device cpu     # gloo plumbing check only

The parent never touches the GPU: it spawns one fresh child per rank and prints ONE JSON line:

    {"world": 8, "mode": "devices",
     "all_reduce": [{"bytes": 16384, "rccl_all_reduce": {"us": .., "algbw_GBps": .., "busbw_GBps": ..},
                     "torch_all_reduce": {..}, "direct_one_shot": {..}, "direct_two_shot": {..},
                     "rccl_in_graph": {..}, "direct_in_graph": {..}}, ...],
     "all_gather": [...], "reduce_scatter": [...], "fused_norm": [...], "p2p": {...},
     "recommend": {"VDAMD_CUSTOM_ALL_REDUCE_ONE_SHOT_BYTES": .., "VDAMD_CUSTOM_ALL_REDUCE_MAX_BYTES": ..}}

Times are the slowest rank's mean over ``--iters`` back-to-back calls between two device events
(``*_in_graph``: the calls replayed from a hipGraph, the way a decode step issues them).  Bus bandwidth
of an all-reduce is ``2 (W-1)/W x bytes / time``: on the fully connected xGMI mesh (7 links x ~153 GB/s
bidirectional per GPU) the peer-mapped two-shot form should approach 7 links' worth, a ring one link's.
The reference leaves all of this to NCCL defaults (reference ``docker-compose.yml:39-41``).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _sizes(lo: int, hi: int):
    s = lo
    while s <= hi:
        yield s
        s *= 4


# ------------------------------------------------------------------------------- child (one rank)
def child(args) -> None:
    import torch
    import torch.distributed as dist

    from vllm_distributed_amd.parallel import state as pstate

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = os.environ["MODE"]
    if mode == "devices":
        torch.cuda.set_device(rank)
        dev, backend = torch.device(f"cuda:{rank}"), "nccl"
    elif mode == "shared":
        torch.cuda.set_device(0)
        dev, backend = torch.device("cuda:0"), "gloo"
    else:
        dev, backend = torch.device("cpu"), "gloo"
    pstate.init_distributed_environment(world, rank, "tcp://127.0.0.1:" + os.environ["MASTER_PORT"], rank, backend, dev)
    pstate.initialize_model_parallel(world, 1)
    g = pstate.get_tp_group()
    gpu = dev.type == "cuda"

    comm = car = None
    notes = []
    if mode == "devices":
        try:
            from vllm_distributed_amd.parallel.rccl import RcclCommunicator
            comm = RcclCommunicator(g)
        except Exception as e:  # noqa: BLE001
            notes.append(f"rccl binding: {e!r}")
    if gpu:
        try:
            from vllm_distributed_amd.parallel.custom_all_reduce import CustomAllReduce
            car = CustomAllReduce(g, max_bytes=args.max_bytes, ranks_per_device=world if mode == "shared" else 1)
        except Exception as e:  # noqa: BLE001
            notes.append(f"peer-mapped path: {e!r}")
    have = g.all_gather_object((comm is not None, car is not None))
    if not all(h[0] for h in have):
        comm = None
    if not all(h[1] for h in have):
        car = None

    def timed(fn, iters: int) -> float:
        """Mean microseconds per call, slowest rank."""
        for _ in range(3):
            fn()
        if gpu:
            torch.cuda.synchronize()
        g.barrier()
        if gpu:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / iters
        else:
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            us = (time.perf_counter() - t0) * 1e6 / iters
        return max(g.all_gather_object(us))

    def in_graph(fn, iters: int, even: bool = False):
        """The same call as ``per`` nodes of one hipGraph (None if the capture fails)."""
        if not gpu:
            return None
        per = 8
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(per):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(per):
                    fn()
            us = timed(gr.replay, max(2, iters // per)) / per
            del gr
            return us
        except Exception as e:  # noqa: BLE001
            notes.append(f"graph capture: {e!r}")
            return None

    def entry(us, nbytes: int, bus_factor: float):
        if us is None:
            return None
        return {"us": round(us, 2), "algbw_GBps": round(nbytes / us / 1e3, 2),
                "busbw_GBps": round(bus_factor * nbytes / us / 1e3, 2)}

    out = {"world": world, "mode": mode, "all_reduce": [], "all_gather": [], "reduce_scatter": [], "fused_norm": []}
    ar_factor = 2.0 * (world - 1) / world
    for nbytes in _sizes(16 << 10, args.max_bytes):
        iters = max(4, min(args.iters, int(args.iters * (4 << 20) / max(nbytes, 4 << 20)) or 4))
        n = nbytes // 2
        x = torch.ones(n, dtype=torch.bfloat16, device=dev)
        row = {"bytes": nbytes}
        row["torch_all_reduce"] = entry(timed(lambda: g._td_all_reduce(x), iters), nbytes, ar_factor)
        if comm is not None:
            row["rccl_all_reduce"] = entry(timed(lambda: comm.all_reduce(x), iters), nbytes, ar_factor)
            if nbytes <= (4 << 20):
                row["rccl_in_graph"] = entry(in_graph(lambda: comm.all_reduce(x), iters), nbytes, ar_factor)
        if car is not None and car.fits(nbytes):
            def direct(two):
                src = car.stage_view((n,), torch.bfloat16)          # the producer writes in place: no copy
                return car.all_reduce(src, two_shot=two)
            row["direct_one_shot"] = entry(timed(lambda: direct(False), iters), nbytes, ar_factor) \
                if nbytes * world <= (64 << 20) else None
            row["direct_two_shot"] = entry(timed(lambda: direct(True), iters), nbytes, ar_factor)
            if nbytes <= (4 << 20):
                best_two = (row["direct_one_shot"] is None
                            or row["direct_two_shot"]["us"] < row["direct_one_shot"]["us"])
                row["direct_in_graph"] = entry(in_graph(lambda: direct(best_two), iters), nbytes, ar_factor)
            car.end_step()
        out["all_reduce"].append(row)

        # all-gather / reduce-scatter with `nbytes` as the FULL message (each rank holds 1/W of it)
        if n % (8 * world) == 0:
            piece = torch.ones(n // world, dtype=torch.bfloat16, device=dev)
            full = torch.empty(n, dtype=torch.bfloat16, device=dev)
            f = (world - 1) / world
            ag = {"bytes": nbytes, "torch_all_gather": entry(timed(lambda: g._td_all_gather(full, piece), iters), nbytes, f)}
            if comm is not None:
                ag["rccl_all_gather"] = entry(timed(lambda: comm.all_gather(full, piece), iters), nbytes, f)
            if car is not None and car.fits(nbytes // world):
                ag["direct_all_gather"] = entry(timed(lambda: car.all_gather(piece, out=full), iters), nbytes, f)
                car.end_step()
            out["all_gather"].append(ag)
            if comm is not None:
                out["reduce_scatter"].append({"bytes": nbytes, "rccl_reduce_scatter": entry(
                    timed(lambda: comm.reduce_scatter(piece, x), iters), nbytes, f)})

    # fused all-reduce + add + RMSNorm at the row counts of decode steps (hidden 8192)
    if car is not None:
        from vllm_distributed_amd import ops
        h = 8192
        wt = torch.ones(h, dtype=torch.bfloat16, device=dev)
        for rows in (1, 16, 64, 256, 1024):
            nbytes = rows * h * 2
            if not car.fits(nbytes):
                continue
            res = torch.zeros(rows, h, dtype=torch.bfloat16, device=dev)
            per, _ = car.shard_rows(rows)
            res_s = torch.zeros(per, h, dtype=torch.bfloat16, device=dev)

            def fused(sharded):
                src = car.stage_view((rows, h), torch.bfloat16)
                return car.all_reduce_add_rms_norm(src, res_s if sharded else res, wt, 1e-5, sharded)

            def unfused():
                src = car.stage_view((rows, h), torch.bfloat16)
                y = car.all_reduce(src)
                return ops.fused_add_rms_norm(y, res, wt, 1e-5)
            row = {"rows": rows, "bytes": nbytes,
                   "fused_replicated": entry(timed(lambda: fused(False), args.iters), nbytes, ar_factor)
                   if nbytes * world <= (64 << 20) else None,
                   "fused_row_sharded": entry(timed(lambda: fused(True), args.iters), nbytes, ar_factor)}
            try:
                row["all_reduce_then_norm"] = entry(timed(unfused, args.iters), nbytes, ar_factor)
            except Exception as e:  # noqa: BLE001
                notes.append(f"unfused norm: {e!r}")
            car.end_step()
            out["fused_norm"].append(row)

    # point-to-point: one pair alone, then every rank to its ring neighbour at once
    p2p = {}
    nb = min(args.max_bytes, 64 << 20)
    buf, got = torch.ones(nb // 2, dtype=torch.bfloat16, device=dev), torch.empty(nb // 2, dtype=torch.bfloat16, device=dev)
    if comm is not None:
        def pair():
            if rank == 0:
                comm.send(buf, 1)
            elif rank == 1:
                comm.recv(got, 0)
        p2p["rccl_pair_0_to_1"] = entry(timed(pair, 8), nb, 1.0)
        p2p["rccl_ring_neighbours"] = entry(
            timed(lambda: comm.send_recv(buf, (rank + 1) % world, got, (rank - 1) % world), 8), nb, 1.0)
    p2p["bytes"] = nb
    out["p2p"] = p2p

    # thresholds: one-shot while it is not slower than two-shot; the peer-mapped path while it is not slower than RCCL
    one_shot, max_direct, prefix = 0, 0, True
    for row in out["all_reduce"]:
        o, t, r = row.get("direct_one_shot"), row.get("direct_two_shot"), row.get("rccl_all_reduce") or row.get("torch_all_reduce")
        if prefix and o and t and o["us"] <= t["us"]:
            one_shot = row["bytes"]
        elif t:
            prefix = False
        best = min([e["us"] for e in (o, t) if e], default=None)
        if best is not None and r and best <= r["us"]:
            max_direct = row["bytes"]
    out["recommend"] = {"VDAMD_CUSTOM_ALL_REDUCE_ONE_SHOT_BYTES": one_shot,
                        "VDAMD_CUSTOM_ALL_REDUCE_MAX_BYTES": max_direct,
                        "basis": "largest measured size where the form is not slower (eager, slowest rank)"}
    timed_out = bool(car.timed_out()) if car is not None else False
    out["direct_timed_out"] = any(g.all_gather_object(timed_out))
    out["notes"] = sorted({n for ns in g.all_gather_object(notes) for n in ns})
    if rank == 0:
        print(json.dumps(out), flush=True)
    g.barrier()
    if car is not None:
        car.close()
    if comm is not None:
        comm.close()
    pstate.destroy_model_parallel()


# ---------------------------------------------------------------------------------------- parent
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks (default: every visible GPU)")
    ap.add_argument("--share-device", action="store_true", help="all ranks on GPU 0 (peer-mapped path + gloo only)")
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda")
    ap.add_argument("--max-bytes", type=int, default=256 << 20)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--timeout", type=float, default=540.0)
    args = ap.parse_args()
    if os.environ.get("VDAMD_PREFLIGHT_CHILD") == "1":
        child(args)
        return
    import torch
    from vllm_distributed_amd.utils import get_open_port
    ndev = torch.cuda.device_count()                  # no HIP context in the parent
    world = args.gpus or max(2, ndev)
    mode = "cpu" if args.device == "cpu" else "shared" if args.share_device else "devices"
    if mode == "devices" and world > ndev:
        raise SystemExit(f"--gpus {world} but {ndev} visible GPU(s); use --share-device for a rehearsal")
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT=str(port), MODE=mode,
                   VDAMD_PREFLIGHT_CHILD="1", GLOO_SOCKET_IFNAME=os.environ.get("GLOO_SOCKET_IFNAME", "lo"),
                   NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   VLLM_LOGGING_LEVEL=os.environ.get("VLLM_LOGGING_LEVEL", "WARNING"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    try:
        so, _ = procs[0].communicate(timeout=args.timeout)
        for p in procs:
            rc = rc or p.wait(timeout=60)
        lines = [l for l in so.splitlines() if l.startswith("{")]
        if lines:
            print(lines[-1], flush=True)
        else:
            rc = rc or 1
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


if __name__ == "__main__":
    main()
