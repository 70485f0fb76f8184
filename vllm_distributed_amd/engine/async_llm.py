"""Asynchronous engine client for the HTTP server.

The engine (scheduler + executor loop) runs on its own thread so that a slow
step never blocks the event loop and the event loop never delays a step.
Requests enter through a thread-safe inbox; outputs are posted back to
per-request ``asyncio.Queue``s with ``call_soon_threadsafe``.  This is the
``EngineClient`` the reference obtains from
``build_async_engine_client_from_engine_args`` (reference ``src/launch.py:30,
407-410,431``).
"""
from __future__ import annotations

import asyncio
import itertools
import queue
import threading
import time
from typing import AsyncIterator, Dict, Optional, Sequence, Union

from ..config import VllmConfig
from ..logger import init_logger
from ..metrics import EngineMetrics
from ..outputs import RequestOutput
from ..sampling_params import SamplingParams
from .arg_utils import EngineArgs
from .core import EngineDeadError
from .llm_engine import LLMEngine

logger = init_logger(__name__)


class OutputCollector:
    """Hand-off of one request's outputs from the engine thread to its
    consumer coroutine.  When the consumer lags (slow client, busy event
    loop) successive outputs are merged into one - cumulative fields are
    replaced, the deltas concatenated - so the event loop's work per step is
    bounded by the number of consumers that are keeping up, not by the batch."""

    __slots__ = ("_pending", "_event")

    def __init__(self) -> None:
        self._pending = None
        self._event = asyncio.Event()

    def put(self, item) -> None:
        cur = self._pending
        if cur is None or isinstance(item, BaseException) or isinstance(cur, BaseException):
            if not isinstance(cur, BaseException):
                self._pending = item
        else:
            item.new_token_ids = cur.new_token_ids + item.new_token_ids
            item.new_text = cur.new_text + item.new_text
            self._pending = item
        self._event.set()

    async def get(self):
        while self._pending is None:
            await self._event.wait()
            self._event.clear()
        item, self._pending = self._pending, None
        self._event.clear()
        return item


class AsyncLLM:
    def __init__(self, vllm_config: VllmConfig, executor_class=None, log_stats: bool = True,
                 multiprocess: bool = False):
        self.vllm_config = vllm_config
        self.engine = LLMEngine(vllm_config, executor_class, multiprocess=multiprocess)
        self.tokenizer = self.engine.tokenizer
        self.metrics = EngineMetrics(vllm_config.model_config.served_model_name[0])
        self._inbox: "queue.Queue[tuple]" = queue.Queue()
        self._queues: Dict[str, OutputCollector] = {}
        self._loop: Optional[asyncio.AbstractEventLoop] = None
        self._counter = itertools.count()
        self._stop = threading.Event()
        self._dead: Optional[BaseException] = None
        self._log_stats = log_stats
        self._thread = threading.Thread(target=self._run, name="engine-core", daemon=True)
        self._thread.start()

    @classmethod
    def from_engine_args(cls, engine_args: EngineArgs, multiprocess: bool = False) -> "AsyncLLM":
        return cls(engine_args.create_engine_config(), log_stats=not engine_args.disable_log_stats,
                   multiprocess=multiprocess)

    # ------------------------------------------------------------ engine thread
    def _post(self, rid: str, item) -> None:
        self._post_many([(rid, item)])

    def _deliver(self, items) -> None:        # runs on the event loop
        for rid, item in items:
            q = self._queues.get(rid)
            if q is not None:
                q.put(item)

    def _post_many(self, items) -> None:
        """One loop wake-up per engine step, however many requests produced output."""
        loop = self._loop
        if loop is None or not items:
            return
        try:
            loop.call_soon_threadsafe(self._deliver, items)
        except RuntimeError:        # loop closed
            pass

    def _run(self) -> None:
        eng = self.engine
        last_log = time.time()
        try:
            while not self._stop.is_set():
                block = not eng.has_unfinished_requests()
                try:
                    while True:
                        msg = self._inbox.get(timeout=0.2) if block else self._inbox.get_nowait()
                        block = False
                        self._handle(msg)
                except queue.Empty:
                    pass
                if not eng.has_unfinished_requests():
                    continue
                outs = eng.step()
                now = time.time()
                for out in outs:
                    self.metrics.observe_output(out, now)
                self._post_many([(out.request_id, out) for out in outs])
                stats = eng.core.stats()
                self.metrics.observe_scheduler(stats)
                if self._log_stats and now - last_log > 10:
                    last_log = now
                    logger.info("running %d, waiting %d, KV usage %.1f%%, prefix hit %.1f%%, %s",
                                stats.num_running, stats.num_waiting, 100 * stats.kv_cache_usage,
                                100 * stats.prefix_cache_hits / max(1, stats.prefix_cache_queries),
                                {k: round(v, 2) for k, v in eng.core.timer.summary().items()})
        except BaseException as e:  # noqa: BLE001
            logger.error("engine thread died: %r", e)
            self._dead = e
            for rid in list(self._queues):
                self._post(rid, e)

    def _handle(self, msg: tuple) -> None:
        kind = msg[0]
        if kind == "add":
            _, rid, prompt, params, priority, lora = msg
            try:
                self.engine.add_request(rid, prompt, params, priority=priority, lora_request=lora)
                self.metrics.request_arrived()
            except BaseException as e:  # noqa: BLE001 - report to the caller, keep serving
                self._post(rid, e)
        elif kind == "abort":
            self.engine.abort_request(msg[1])
        elif kind == "call":
            _, fn, fut, loop = msg
            try:
                res = fn()
                loop.call_soon_threadsafe(fut.set_result, res)
            except BaseException as e:  # noqa: BLE001
                loop.call_soon_threadsafe(fut.set_exception, e)

    # ---------------------------------------------------------------- client
    @property
    def errored(self) -> bool:
        return self._dead is not None or self.engine.core.dead

    @property
    def is_running(self) -> bool:
        return self._thread.is_alive() and not self.errored

    async def get_vllm_config(self) -> VllmConfig:
        return self.vllm_config

    async def get_tokenizer(self):
        return self.tokenizer

    async def generate(self, prompt: Union[str, Sequence[int], dict], sampling_params: SamplingParams,
                       request_id: Optional[str] = None, priority: int = 0, lora_request=None,
                       ) -> AsyncIterator[RequestOutput]:
        if self.errored:
            raise EngineDeadError(f"engine is dead: {self._dead!r}")
        self._loop = asyncio.get_running_loop()
        rid = request_id or f"cmpl-{next(self._counter)}"
        q = OutputCollector()
        self._queues[rid] = q
        self._inbox.put(("add", rid, prompt, sampling_params, priority, lora_request))
        finished = False
        try:
            while True:
                item = await q.get()
                if isinstance(item, BaseException):
                    finished = True
                    raise item
                yield item
                if item.finished:
                    finished = True
                    return
        finally:
            self._queues.pop(rid, None)
            if not finished:        # client went away
                self._inbox.put(("abort", rid))

    async def abort(self, request_id: str) -> None:
        self._inbox.put(("abort", request_id))

    async def _call(self, fn):
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        self._inbox.put(("call", fn, fut, loop))
        return await fut

    async def check_health(self) -> None:
        if self.errored:
            raise EngineDeadError(f"engine is dead: {self._dead!r}")

    async def start_profile(self) -> None:
        await self._call(lambda: self.engine.core.profile(True))

    async def stop_profile(self) -> None:
        await self._call(lambda: self.engine.core.profile(False))

    async def reset_prefix_cache(self) -> bool:
        return await self._call(self.engine.core.reset_prefix_cache)

    async def add_lora(self, lora_request) -> bool:
        return await self._call(lambda: self.engine.core.add_lora(lora_request))

    async def remove_lora(self, lora_id: int) -> bool:
        return await self._call(lambda: self.engine.core.remove_lora(lora_id))

    async def list_loras(self):
        return await self._call(self.engine.core.list_loras)

    async def pin_lora(self, lora_id: int) -> bool:
        return await self._call(lambda: self.engine.core.pin_lora(lora_id))

    async def sleep(self, level: int = 1) -> None:
        await self._call(lambda: self.engine.core.sleep(level))

    async def wake_up(self, tags=None) -> None:
        await self._call(lambda: self.engine.core.wake_up(tags))

    def shutdown(self) -> None:
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=5)
        self.engine.shutdown()


EngineClient = AsyncLLM
