"""Bootstrap blobs over the control plane.

Bring-up of the data plane needs a few small opaque values to reach every rank of a group before
any collective exists: the RCCL unique id of a communicator, the hipIpc handles of the peer-mapped
buffers, the ids of the pipeline edge communicators.  Where a worker has a control-plane
connection to the head (reference ``src/launch.py:226-296``: the head already drives every worker
through it) those values travel there: each rank hands its blob to the head under a key and gets
every rank's blob back once all have arrived.  Without a control plane (single process, the SPMD
executor, unit tests) the group's CPU (gloo) channel carries them as before.
"""
from __future__ import annotations

import asyncio
import threading
from typing import Any, Callable, Dict, List, Optional

# installed by the worker process: (key, rank, world, payload) -> list of every rank's payload
_EXCHANGE: Optional[Callable[[str, int, int, Any], List[Any]]] = None
_SEQ: Dict[str, int] = {}
_LOCK = threading.Lock()


def set_exchange(fn: Optional[Callable[[str, int, int, Any], List[Any]]]) -> None:
    global _EXCHANGE
    _EXCHANGE = fn
    _SEQ.clear()


def available() -> bool:
    return _EXCHANGE is not None


def exchange(scope: str, rank: int, world: int, payload: Any) -> List[Any]:
    """All-gather of ``payload`` among the ``world`` ranks that call with the same ``scope``, in
    call order per scope (every rank of a group makes the same sequence of calls)."""
    if _EXCHANGE is None:
        raise RuntimeError("no control-plane exchange installed")
    with _LOCK:
        n = _SEQ.get(scope, 0)
        _SEQ[scope] = n + 1
    return _EXCHANGE(f"{scope}#{n}", rank, world, payload)


class HeadRendezvous:
    """Head side: collects the blobs of one key and releases them when all ranks are in."""

    def __init__(self) -> None:
        self._slots: Dict[str, Dict[str, Any]] = {}

    async def exchange(self, key: str, rank: int, world: int, payload: Any, timeout: float = 600.0) -> List[Any]:
        slot = self._slots.get(key)
        if slot is None:
            slot = self._slots[key] = {"vals": {}, "done": asyncio.Event(), "world": int(world), "left": int(world)}
        if int(world) != slot["world"] or not 0 <= int(rank) < slot["world"] or int(rank) in slot["vals"]:
            raise ValueError(f"inconsistent bootstrap exchange for {key!r} (rank {rank} of {world})")
        slot["vals"][int(rank)] = payload
        if len(slot["vals"]) == slot["world"]:
            slot["done"].set()
        try:
            await asyncio.wait_for(slot["done"].wait(), timeout)
        finally:
            slot["left"] -= 1
            if slot["left"] <= 0:
                self._slots.pop(key, None)
        return [slot["vals"][r] for r in range(slot["world"])]


def install_worker_exchange(peer, loop: asyncio.AbstractEventLoop, timeout: float = 600.0) -> None:
    """Worker side: route ``exchange`` through the head's ``bootstrap_exchange`` (called from the
    worker thread while the event loop keeps pumping the connection)."""
    cache: Dict[str, Any] = {}

    async def call(key: str, rank: int, world: int, payload: Any) -> List[Any]:
        fn = cache.get("fn")
        if fn is None:
            fn = cache["fn"] = await peer.get_param("bootstrap_exchange", timeout=30)
        return await fn(key, rank, world, payload)

    def exchange_sync(key: str, rank: int, world: int, payload: Any) -> List[Any]:
        fut = asyncio.run_coroutine_threadsafe(call(key, rank, world, payload), loop)
        return list(fut.result(timeout))

    set_exchange(exchange_sync)
