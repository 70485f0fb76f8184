"""Bodies of the multi-rank data-plane tests, one process per rank.

    RANK=r WORLD_SIZE=n MASTER_PORT=p MODE=devices|shared|cpu python tests/multirank_bodies.py <body>

``MODE`` says where the ranks live:

* ``devices`` - one rank per physical GPU (``cuda:RANK``), ``torch.distributed`` on nccl (= RCCL): the
  configuration the reference runs its workers in (reference ``src/launch.py:149-296``).  Only here do the
  RCCL communicator, the cross-device ``hipIpcOpenMemHandle`` mappings and the pipeline edge communicators
  carry real traffic.
* ``shared``  - every rank on ``cuda:0``, ``torch.distributed`` on gloo with host staging.  RCCL refuses two
  ranks on one device, so the ``rccl`` and ``pipeline`` bodies run over a stand-in with the same methods;
  the peer-mapped collectives run for real (hipIpc works between processes on one device).
* ``cpu``     - gloo on CPU tensors: checks the bodies themselves (payloads, expected values, call order),
  so that the first run on a multi-GPU box does not trip over a mistake in the test.

Every body prints one line ``RESULT {json}`` with ``ok`` and, on failure, ``why``.  Payloads are small
integers, exact in bf16 / fp16 / fp32, so every comparison is ``torch.equal``: a reader that saw a stale or
half-written buffer cannot hide inside a tolerance.
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vllm_distributed_amd.parallel import state as pstate  # noqa: E402

RANK, WORLD = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
MODE = os.environ.get("MODE", "cpu")
FAILS = []


def fail(msg: str) -> None:
    FAILS.append(msg)


def setup(tp: int, pp: int):
    if MODE == "devices":
        torch.cuda.set_device(RANK)
        dev, backend = torch.device(f"cuda:{RANK}"), "nccl"
    elif MODE == "shared":
        torch.cuda.set_device(0)
        dev, backend = torch.device("cuda:0"), "gloo"
    else:
        dev, backend = torch.device("cpu"), "gloo"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    pstate.init_distributed_environment(WORLD, RANK, "tcp://127.0.0.1:" + os.environ["MASTER_PORT"], RANK, backend, dev)
    pstate.initialize_model_parallel(tp, pp)
    return dev


def sync(dev) -> None:
    if dev.type == "cuda":
        torch.cuda.synchronize()


class TorchDistComm:
    """The ``RcclCommunicator`` methods over ``torch.distributed`` (gloo, host-staged for device tensors):
    what the ``rccl`` body talks to where RCCL itself cannot run (CPU, ranks sharing a device)."""

    def __init__(self, group):
        self.g, self.rank, self.world_size = group, group.rank_in_group, group.world_size

    def all_reduce(self, x):
        self.g._td_all_reduce(x)

    def all_gather(self, out, x):
        self.g._td_all_gather(out, x)

    def reduce_scatter(self, out, x):
        full = x.clone()
        self.g._td_all_reduce(full)
        n = out.shape[0]
        out.copy_(full[self.rank * n:(self.rank + 1) * n])

    def broadcast(self, x, src):
        self.g._td_broadcast(x, self.g.ranks[src])

    def send_recv(self, send, dst, recv, src):
        ops, hs, hr = [], None, None
        if send is not None:
            hs = send.cpu() if send.is_cuda else send
            ops.append(dist.P2POp(dist.isend, hs, self.g.ranks[dst], group=self.g.device_group))
        if recv is not None:
            hr = torch.empty(recv.shape, dtype=recv.dtype) if recv.is_cuda else recv
            ops.append(dist.P2POp(dist.irecv, hr, self.g.ranks[src], group=self.g.device_group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if recv is not None and hr is not recv:
            recv.copy_(hr)

    def close(self):
        pass


# ------------------------------------------------------------------ RCCL communicator
def _payload(n: int, dtype, dev, rank: int, salt: int) -> torch.Tensor:
    """Integers in [0, 13) + rank + 1 + salt % 5: sums over 8 ranks stay below 256 (exact in bf16)."""
    x = (torch.arange(n, device=dev, dtype=torch.int64) * 7 + salt) % 13 + rank + 1 + salt % 5
    return x.to(dtype)


def _check_collectives(comm, dev, n: int, dtype, salt: int, tag: str) -> None:
    w, r = comm.world_size, comm.rank
    parts = [_payload(n, dtype, dev, q, salt) for q in range(w)]
    total = torch.stack([p.float() for p in parts]).sum(0).to(dtype)
    x = parts[r].clone()
    comm.all_reduce(x)
    sync(dev)
    if not torch.equal(x, total):
        fail(f"{tag}: all_reduce n={n} {dtype}")
    out = torch.empty(w * n, dtype=dtype, device=dev)
    comm.all_gather(out, parts[r])
    sync(dev)
    if not torch.equal(out, torch.cat(parts)):
        fail(f"{tag}: all_gather n={n} {dtype}")
    if n % w == 0:
        rs = torch.empty(n // w, dtype=dtype, device=dev)
        comm.reduce_scatter(rs, parts[r])
        sync(dev)
        if not torch.equal(rs, total[r * (n // w):(r + 1) * (n // w)]):
            fail(f"{tag}: reduce_scatter n={n} {dtype}")
    b = parts[r].clone()
    comm.broadcast(b, salt % w)
    sync(dev)
    if not torch.equal(b, parts[salt % w]):
        fail(f"{tag}: broadcast n={n} {dtype}")
    got = torch.zeros(n, dtype=dtype, device=dev)
    comm.send_recv(parts[r], (r + 1) % w, got, (r - 1) % w)            # every rank passes its part round the ring
    sync(dev)
    if not torch.equal(got, parts[(r - 1) % w]):
        fail(f"{tag}: grouped send/recv n={n} {dtype}")


def body_rccl() -> dict:
    dev = setup(WORLD, 1)
    g = pstate.get_tp_group()
    real = MODE == "devices"
    if real:
        from vllm_distributed_amd.parallel.rccl import RcclCommunicator
        comm = RcclCommunicator(g)
    else:
        comm = TorchDistComm(g)
    big = (32 << 20) if real else (1 << 16)
    salt = 0
    for n in (8 * WORLD, 8192, 1 << 19, big):
        for dtype in (torch.bfloat16, torch.float32, torch.int32):
            salt += 1
            _check_collectives(comm, dev, n, dtype, salt, "eager")
    replays = 0
    if real:
        # the same calls as nodes of one hipGraph, replayed on changing data
        n, w, r = 1 << 16, WORLD, RANK
        xin = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        ar = torch.zeros_like(xin)
        ag = torch.zeros(w * n, dtype=torch.bfloat16, device=dev)
        rs = torch.zeros(n // w, dtype=torch.bfloat16, device=dev)
        ring = torch.zeros_like(xin)

        def step():
            ar.copy_(xin)
            comm.all_reduce(ar)
            comm.all_gather(ag, xin)
            comm.reduce_scatter(rs, xin)
            comm.send_recv(xin, (r + 1) % w, ring, (r - 1) % w)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        for rep in range(100):
            parts = [_payload(n, torch.bfloat16, dev, q, rep) for q in range(w)]
            xin.copy_(parts[r])
            gr.replay()
            replays += 1
            if rep < 3 or rep % 10 == 9:
                torch.cuda.synchronize()
                total = torch.stack([p.float() for p in parts]).sum(0).to(torch.bfloat16)
                if not (torch.equal(ar, total) and torch.equal(ag, torch.cat(parts))
                        and torch.equal(rs, total[r * (n // w):(r + 1) * (n // w)])
                        and torch.equal(ring, parts[(r - 1) % w])):
                    fail(f"graph replay {rep}")
                    break
        torch.cuda.synchronize()
        del gr
    g.barrier()
    comm.close()
    return dict(real_rccl=real, replays=replays)


# ------------------------------------------------------------- peer-mapped collectives
def body_car() -> dict:
    """Every form of the peer-mapped collectives on exact-integer payloads, eagerly and in one
    hipGraph replayed 100 times; no bounded spin may have run out."""
    if MODE == "cpu":
        return dict(skipped="peer-mapped collectives need device memory")
    dev = setup(WORLD, 1)
    from vllm_distributed_amd.parallel.custom_all_reduce import CustomAllReduce
    g = pstate.get_tp_group()
    car = CustomAllReduce(g, max_bytes=4 << 20, ranks_per_device=WORLD if MODE == "shared" else 1)
    car.fence_start = os.environ.get("FENCE_START") == "1"
    w, r = WORLD, RANK
    salt = 0
    for it in range(3):
        for n, two in ((8, False), (4096, False), (8192 * 7, False), (8192 * 7, True), (1 << 20, True),
                       (2 << 20, None), (1000 * 8, None), (40, True)):
            for dtype in (torch.bfloat16, torch.float16):
                salt += 1
                parts = [_payload(n, dtype, dev, q, salt) for q in range(w)]
                total = torch.stack([p.float() for p in parts]).sum(0).to(dtype)
                src = parts[r]
                if it % 2:
                    src = car.stage_view((n,), dtype)
                    src.copy_(parts[r])
                out = car.all_reduce(src, two_shot=two)
                sync(dev)
                if not torch.equal(out, total):
                    fail(f"all_reduce n={n} two_shot={two} {dtype} it={it}")
    # all-reduce + residual add + RMSNorm: the residual (the sum) must be exact, the normalised rows close
    for it, (rows, h) in enumerate(((1, 512), (5, 512), (64, 1024), (200, 8192), (37, 4096), (3, 8192))):
        for sharded in (False, True):
            salt += 1
            parts = [_payload(rows * h, torch.bfloat16, dev, q, salt).view(rows, h) for q in range(w)]
            res = _payload(rows * h, torch.bfloat16, dev, 0, salt + 3).view(rows, h)
            wt = (_payload(h, torch.bfloat16, dev, 0, salt) / 8).to(torch.bfloat16)
            tot = (torch.stack([p.float() for p in parts]).sum(0) + res.float()).to(torch.bfloat16)
            tf = tot.float()
            want = tf * torch.rsqrt(tf.pow(2).mean(-1, keepdim=True) + 1e-5) * wt.float()
            per, mine = car.shard_rows(rows)
            lo = min(r * per, rows)
            my_res = torch.zeros(per, h, dtype=torch.bfloat16, device=dev) if sharded else res.clone()
            if sharded:
                my_res[:mine] = res[lo:lo + mine]
            src = parts[r]
            if it % 2 == 0:
                src = car.stage_view((rows, h), torch.bfloat16)
                src.copy_(parts[r])
            got = car.all_reduce_add_rms_norm(src, my_res, wt, 1e-5, sharded)
            sync(dev)
            got_res, want_res = (my_res[:mine], tot[lo:lo + mine]) if sharded else (my_res, tot)
            if not torch.equal(got_res, want_res):
                fail(f"fused norm rows={rows} h={h} sharded={sharded}: residual")
            if not torch.allclose(got.float(), want, rtol=2e-2, atol=2e-2):
                fail(f"fused norm rows={rows} h={h} sharded={sharded}: output")
    for dtype, shape in ((torch.float32, (3, 8)), (torch.bfloat16, (40, 16)), (torch.int32, (4,))):
        n = 1
        for d in shape:
            n *= d
        piece = (torch.arange(n, device=dev).reshape(shape) + 1000 * r).to(dtype)
        out = car.all_gather(piece)
        want = torch.cat([(torch.arange(n, device=dev).reshape(shape) + 1000 * q).to(dtype) for q in range(w)])
        if not torch.equal(out, want):
            fail(f"all_gather {dtype}")
    car.end_step()
    if car.parity != 0:
        fail("parity after end_step")

    rows, h = 48, 1024
    xin = torch.zeros(rows, h, device=dev, dtype=torch.bfloat16)
    res_r = torch.zeros(rows, h, device=dev, dtype=torch.bfloat16)
    per, mine = car.shard_rows(rows)
    res_s = torch.zeros(per, h, device=dev, dtype=torch.bfloat16)
    ones = torch.ones(h, device=dev, dtype=torch.bfloat16)

    def step():
        a = car.stage_view((rows, h), torch.bfloat16)
        a.copy_(xin)
        y0 = car.all_reduce(a, two_shot=False)
        b = car.stage_view((rows, h), torch.bfloat16)
        b.copy_(xin)
        y1 = car.all_reduce_add_rms_norm(b, res_r, ones, 1e-5, False)
        y2 = car.all_reduce_add_rms_norm(xin, res_s, ones, 1e-5, True)
        y3 = car.all_gather(y0[:2])
        y4 = car.all_reduce(xin, two_shot=True)
        car.end_step()
        return y0, y1, y2, y3, y4

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        outs = step()
    for rep in range(100):
        parts = [_payload(rows * h, torch.bfloat16, dev, q, rep).view(rows, h) for q in range(w)]
        xin.copy_(parts[r])
        res_r.zero_()
        res_s.zero_()
        gr.replay()
        if rep < 3 or rep % 10 == 9:
            torch.cuda.synchronize()
            tot = torch.stack([p.float() for p in parts]).sum(0).to(torch.bfloat16)
            lo = min(r * per, rows)
            y0, y1, y2, y3, y4 = outs
            tf = tot.float()
            normed = tf * torch.rsqrt(tf.pow(2).mean(-1, keepdim=True) + 1e-5)
            good = (torch.equal(y0, tot) and torch.equal(y4, tot) and torch.equal(res_r, tot)
                    and torch.equal(res_s[:mine], tot[lo:lo + mine])
                    and torch.allclose(y1.float(), normed, rtol=2e-2, atol=2e-2)
                    and torch.allclose(y2.float(), normed, rtol=2e-2, atol=2e-2)
                    and torch.equal(y3, torch.cat([tot[:2]] * w)))
            if not good:
                fail(f"graph replay {rep}")
                break
    torch.cuda.synchronize()
    timed_out = car.timed_out()
    if timed_out:
        fail("a bounded spin ran out (timed_out)")
    del gr
    g.barrier()
    car.close()
    return dict(timed_out=timed_out, fence_start=car.fence_start)


def body_car_timeout() -> dict:
    """A peer that never issues its call: the other ranks' kernels must come back within the bound
    (no GPU hang) and the host-visible timeout word must be set, readable without a stream sync."""
    if MODE == "cpu":
        return dict(skipped="peer-mapped collectives need device memory")
    import time
    os.environ["VDAMD_CUSTOM_ALL_REDUCE_TIMEOUT_MS"] = "200"
    dev = setup(WORLD, 1)
    from vllm_distributed_amd.parallel.custom_all_reduce import CustomAllReduce
    g = pstate.get_tp_group()
    car = CustomAllReduce(g, max_bytes=1 << 20, ranks_per_device=WORLD if MODE == "shared" else 1)
    x = torch.ones(4096, dtype=torch.bfloat16, device=dev)
    car.all_reduce(x)                     # everybody: a healthy call first
    sync(dev)
    if car.timed_out():
        fail("timeout word set after a healthy call")
    g.barrier()
    waited = 0.0
    if RANK != WORLD - 1:                 # the last rank plays the hung device
        t0 = time.time()
        car.all_reduce(x)
        sync(dev)
        waited = time.time() - t0
        if not car.timed_out():
            fail("timeout word not set although a peer never arrived")
        if not 0.15 <= waited < 5.0:
            fail(f"kernel returned after {waited:.2f}s for a 0.2 s bound")
    g.barrier()
    car.close()
    return dict(waited=round(waited, 3))


# ------------------------------------------------------------------- pipeline link
def body_pipeline() -> dict:
    """``PipelineLink`` along a chain of WORLD stages with two batches in flight: stage 0 issues
    batch i+1 before anybody has waited for batch i, a middle stage receives, adds one and forwards."""
    dev = setup(1, WORLD)
    from vllm_distributed_amd.parallel.pipeline import PipelineLink
    pp = pstate.get_pp_group()
    link = PipelineLink(pp, dev)
    if MODE == "devices" and not link.accelerated:
        fail("edge communicators are not active with one rank per device")
    rows, h, batches = 96, 1024, 6
    bufs = [(torch.zeros(rows, h, dtype=torch.bfloat16, device=dev), torch.zeros(rows, h, dtype=torch.bfloat16, device=dev))
            for _ in range(2)]                                     # two static buffer pairs: two batches in flight
    events = [None, None]
    seen = []
    for b in range(batches):
        hid, res = bufs[b % 2]
        if events[b % 2] is not None and dev.type == "cuda":
            torch.cuda.current_stream().wait_event(events[b % 2])       # the transfer two batches ago has read them
        if not pp.is_first_rank:
            link.recv(hid, res)
        else:
            hid.copy_(_payload(rows * h, torch.bfloat16, dev, 0, b).view(rows, h))
            res.copy_(_payload(rows * h, torch.bfloat16, dev, 1, b + 2).view(rows, h))
        if not pp.is_last_rank:
            if not pp.is_first_rank:
                hid.add_(1)                                        # this stage's "layers"
            link.send(hid, res)
            events[b % 2] = link._send_done
        else:
            hops = pp.world_size - 2                               # middle stages each added one to hidden
            want_h = (_payload(rows * h, torch.bfloat16, dev, 0, b).float() + hops).to(torch.bfloat16).view(rows, h)
            want_r = _payload(rows * h, torch.bfloat16, dev, 1, b + 2).view(rows, h)
            sync(dev)
            ok = torch.equal(hid, want_h) and torch.equal(res, want_r)
            seen.append(ok)
            if not ok:
                fail(f"batch {b}: wrong tensors at the last stage")
    sync(dev)
    pp.barrier()
    accelerated = link.accelerated
    link.close()
    return dict(accelerated=accelerated, batches_checked=len(seen))


BODIES = {"rccl": body_rccl, "car": body_car, "car_timeout": body_car_timeout, "pipeline": body_pipeline}


def main() -> None:
    info = BODIES[sys.argv[1]]()
    print("RESULT " + json.dumps(dict(ok=not FAILS, why=FAILS[:8], rank=RANK, mode=MODE, **info)), flush=True)
    pstate.destroy_model_parallel()


if __name__ == "__main__":
    main()
