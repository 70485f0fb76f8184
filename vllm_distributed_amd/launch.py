"""Entry point: ``launch.py serve <model> [engine/server flags]`` on the head,
``launch.py remote <head_ip>`` on every worker node.

Same two roles and the same deployment workflow as the reference
(reference ``src/launch.py:668-675`` role dispatch; ``:510-541`` worker RPC
endpoint; ``:543-632`` remote node supervisor and worker; ``:635-665`` local
worker main; ``:589-594`` GC ticker).
"""
from __future__ import annotations

import asyncio
import concurrent.futures
import gc
import multiprocessing
import multiprocessing.connection
import os
import sys
import time
import traceback
import uuid
from typing import Any, Dict, Optional

import cloudpickle

from . import envs
from .logger import init_logger
from .rpc import (ConnectionTransport, RpcPeer, StreamTransport, client_handshake,
                  prepare_peer_readloop)
from .parallel.bootstrap import install_worker_exchange
from .utils import current_platform, get_open_port, hip_device_count_stateless, run_method

logger = init_logger(__name__)


# --------------------------------------------------------------------------
# worker-side RPC endpoint
# --------------------------------------------------------------------------
class _Finished:
    """A call whose reply was already pickled on the worker thread."""
    __slots__ = ("reply",)

    def __init__(self, reply: Optional[bytes]):
        self.reply = reply


def set_peer_run_worker(peer: RpcPeer, wrapper: Any, rank: int, local_rank: int):
    """Publish ``run_worker(pickled)`` on ``peer``.

    The payload is ``(method, unique_reply_rank, args, kwargs[, t_send])``; ``method``
    is a name or a pickled callable.  Calls execute strictly in arrival order on one
    dedicated thread (the data-plane collectives deadlock otherwise) while the
    event loop stays free to answer health checks.  A step whose sampled tokens
    are still on the device is handed to a second, reply thread that waits for
    them and pickles the output, so the worker thread can launch the next queued
    step meanwhile and the device never idles between steps.  Only the reply rank
    ships a result back; the work is done on every rank.
    """
    executor = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix=f"worker-{rank}")
    replier = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix=f"reply-{rank}")

    def _run(pickled: bytes, t_arrive: float):
        t_dequeue = time.time()
        call = cloudpickle.loads(pickled)
        method, unique_reply_rank, args, kwargs = call[:4]
        t_send = call[4] if len(call) > 4 else None
        try:
            result = run_method(wrapper, method, args or [], kwargs or {})
        except Exception as e:
            logger.error("error while running method %r on rank %d: %s\n%s",
                         method if isinstance(method, str) else "<callable>", rank, e,
                         traceback.format_exc())
            raise
        if method == "load_model":
            runner = getattr(getattr(wrapper, "worker", None), "model_runner", None)
            if runner is not None:
                runner.defer_output = True      # the reply thread waits for the device, not this one
        if not getattr(result, "lazy_step_result", False):
            # nothing to wait for: the reply is pickled here, without a hand-off to the reply thread
            return _Finished(_finish(result, unique_reply_rank, t_send, t_arrive, t_dequeue))
        return result, unique_reply_rank, t_send, t_arrive, t_dequeue

    def _finish(result: Any, unique_reply_rank: Optional[int], t_send: Optional[float], t_arrive: float,
                t_dequeue: float) -> Optional[bytes]:
        if getattr(result, "lazy_step_result", False):
            result = result.wait()
        if unique_reply_rank is not None and unique_reply_rank != rank:
            return None
        timing = getattr(result, "timing", None)
        if isinstance(timing, dict) and t_send is not None:
            # control-plane transit, wall clocks of head and worker (one host, or NTP-close hosts)
            timing["rpc_in"] = max(0.0, t_arrive - t_send)
            timing["queued"] = t_dequeue - t_arrive      # behind the previous call on the worker thread
            timing["t_reply"] = time.time()
        return cloudpickle.dumps(result)

    async def run_worker(pickled: bytes):
        loop = asyncio.get_running_loop()
        done = await loop.run_in_executor(executor, _run, pickled, time.time())
        if isinstance(done, _Finished):
            return done.reply
        return await loop.run_in_executor(replier, _finish, *done)

    peer.params["run_worker"] = run_worker
    return run_worker


def gc_runner_loop(loop: asyncio.AbstractEventLoop, period: float = 10.0) -> None:
    """Collect garbage periodically so dropped RPC proxies are finalised on the
    peer and unpickled step payloads are returned to the OS."""
    def tick() -> None:
        gc.collect()
        loop.call_later(period, tick)
    loop.call_later(period, tick)


# --------------------------------------------------------------------------
# worker spawned by the head for one of its own GPUs
# --------------------------------------------------------------------------
async def worker_async_main(transport, vllm_config, rank: int, local_rank: int) -> None:
    from .worker.worker_base import WorkerWrapper
    loop = asyncio.get_running_loop()
    peer, read_loop = await prepare_peer_readloop(loop, transport, f"worker-{rank}", "head")
    peer.params["print"] = print
    peer.params["local_rank"] = local_rank
    install_worker_exchange(peer, loop)
    wrapper = WorkerWrapper(vllm_config=vllm_config, rpc_rank=rank, local_rank=local_rank)
    set_peer_run_worker(peer, wrapper, rank, local_rank)
    try:
        await read_loop()
    except BaseException:  # noqa: BLE001
        pass
    finally:
        os._exit(0)        # the pipe is gone: nothing to clean up that the OS will not


def worker_main(conn, vllm_config_blob: bytes, rank: int, local_rank: int) -> None:
    """``conn``: the worker's end of a Unix socket pair (served by the event loop itself) or of a
    ``multiprocessing.Pipe`` (reference ``src/launch.py:655-665``)."""
    import socket as _socket
    _die_with_parent()
    sys.setswitchinterval(0.0005)      # a thread that got a frame takes the interpreter within 0.5 ms, not 5
    vllm_config = cloudpickle.loads(vllm_config_blob)
    loop = asyncio.new_event_loop()
    asyncio.set_event_loop(loop)
    gc_runner_loop(loop)

    async def main() -> None:
        if isinstance(conn, _socket.socket):
            reader, writer = await asyncio.open_connection(sock=conn)
            transport = StreamTransport(reader, writer)
        else:
            transport = ConnectionTransport(conn)
        await worker_async_main(transport, vllm_config, rank, local_rank)

    loop.run_until_complete(main())
    loop.close()


# --------------------------------------------------------------------------
# worker node: ``launch.py remote <head_ip>``
# --------------------------------------------------------------------------
async def remote_worker_async_main(server_ip: str, available_devices: int, local_rank: int,
                                   node_id: str, port: Optional[int] = None) -> None:
    from .worker.worker_base import WorkerWrapper
    loop = asyncio.get_running_loop()
    port = port or envs.VLLM_SERVER_PORT
    wrapper: Optional[Any] = None
    graceful = False

    def goodbye() -> None:
        """The head announces an orderly shutdown: the connection loss that follows is not a failure."""
        nonlocal graceful
        graceful = True

    while True:
        try:
            reader, writer = await asyncio.open_connection(server_ip, port)
            transport = StreamTransport(reader, writer)
            await client_handshake(transport, envs.VLLM_RPC_SECRET)
            peer, read_loop = await prepare_peer_readloop(loop, transport, f"{node_id}/{local_rank}", "head")
            peer.params["print"] = print
            peer.params["available_devices"] = available_devices
            peer.params["local_rank"] = local_rank
            peer.params["node_id"] = node_id
            peer.params["open_port"] = get_open_port
            peer.params["goodbye"] = goodbye

            async def create_worker(vllm_config_blob: bytes, rank: int, environ: Dict[str, str]):
                nonlocal wrapper
                if wrapper is not None:
                    raise RuntimeError("Worker already created.")
                os.environ.update(environ)
                install_worker_exchange(peer, loop)
                vllm_config = cloudpickle.loads(vllm_config_blob)
                wrapper = WorkerWrapper(vllm_config=vllm_config, rpc_rank=rank, local_rank=local_rank)
                return set_peer_run_worker(peer, wrapper, rank, local_rank)

            peer.params["create_worker"] = create_worker
            logger.info("registered with head %s:%d as %s/%d", server_ip, port, node_id, local_rank)
            await read_loop()
        except BaseException as e:  # noqa: BLE001
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                raise
            # retry for as long as no worker has been created; once one exists
            # the whole node must tear down and restart (reference :576-586)
            if wrapper is not None:
                if graceful:
                    logger.info("head shut down; worker %s/%d exits", node_id, local_rank)
                    os._exit(0)
                logger.error("running worker lost its head connection (%r), shutting down", e)
                os._exit(1)
            logger.info("connection to %s:%d failed (%r); retrying in %.0fs", server_ip, port, e,
                        envs.VLLM_REMOTE_RETRY_SECONDS)
            await asyncio.sleep(envs.VLLM_REMOTE_RETRY_SECONDS)


def _die_with_parent() -> None:
    """Ask the kernel to SIGTERM this process when its parent dies, so killing a
    node supervisor never leaves orphaned GPU workers registering with the head."""
    try:
        import ctypes
        import signal
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)   # PR_SET_PDEATHSIG
    except Exception:  # noqa: BLE001
        pass


def remote_worker_main(server_ip: str, available_devices: int, local_rank: int, node_id: str,
                       port: Optional[int] = None) -> None:
    _die_with_parent()
    loop = asyncio.new_event_loop()
    asyncio.set_event_loop(loop)
    gc_runner_loop(loop)
    loop.run_until_complete(remote_worker_async_main(server_ip, available_devices, local_rank, node_id, port))
    loop.close()


def remote_main(server_ip: str, port: Optional[int] = None, num_devices: Optional[int] = None,
                device_offset: int = 0) -> None:
    """One process per local GPU; if any of them exits the node is torn down
    with a non-zero status so the supervisor (docker ``restart:
    unless-stopped``) brings it back (reference ``src/launch.py:608-632``)."""
    from .executor.distributed import check_rpc_secret
    check_rpc_secret(envs.VLLM_RPC_SECRET, server_ip, "connect to")     # workers unpickle what the head sends
    if num_devices is None:
        num_devices = hip_device_count_stateless() if current_platform.is_rocm() else 1
        num_devices = max(1, num_devices - device_offset)
    node_id = f"{os.uname().nodename}-{uuid.uuid4().hex[:8]}"
    ctx = multiprocessing.get_context("spawn")
    workers = []
    for i in range(num_devices):
        p = ctx.Process(target=remote_worker_main,
                        args=(server_ip, num_devices, device_offset + i, node_id, port), daemon=True)
        p.start()
        workers.append(p)
    import signal

    def on_term(signum, frame):
        for p in workers:
            p.terminate()
        os._exit(1)

    signal.signal(signal.SIGTERM, on_term)
    signal.signal(signal.SIGINT, on_term)
    multiprocessing.connection.wait([p.sentinel for p in workers])
    logger.error("a worker exited, shutting down node")
    for p in workers:
        p.terminate()
    for p in workers:
        p.join()
    sys.exit(1)


# --------------------------------------------------------------------------
# role dispatch
# --------------------------------------------------------------------------
def main(argv: Optional[list] = None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "remote":
        import argparse
        ap = argparse.ArgumentParser(prog="launch.py remote")
        ap.add_argument("server_ip")
        ap.add_argument("--port", type=int, default=None)
        ap.add_argument("--num-devices", type=int, default=None,
                        help="GPUs this node offers (default: all visible)")
        ap.add_argument("--device-offset", type=int, default=0,
                        help="first local GPU index to offer (head and workers sharing one host)")
        a = ap.parse_args(argv[1:])
        remote_main(a.server_ip, a.port, a.num_devices, a.device_offset)
        return
    from .entrypoints.cli import head_main
    head_main(argv)


if __name__ == "__main__":
    main()
