"""Engine core in its own process.

The HTTP front-end (tokenisation, detokenisation, SSE serialisation for
hundreds of streams) and the engine core (scheduler + executor hot loop) are
both Python; in one process they take turns on the interpreter lock and every
millisecond the event loop spends on JSON is a millisecond the scheduler is
late.  ``MPEngineCoreClient`` gives ``LLMEngine`` the ``EngineCore`` interface
over a ``multiprocessing`` pipe to a child process that owns the real core:
requests go down as pickled ``Request`` objects, every step's
``EngineCoreOutput`` list comes back as one message, and the front-end blocks
in ``recv`` (interpreter lock released) while the core works.

This is the split the reference gets from vLLM's
``build_async_engine_client_from_engine_args`` unless
``--disable-frontend-multiprocessing`` is given (reference
``src/launch.py:407-410``).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import threading
import time
from typing import Any, Dict, Iterable, List, Optional, Set, Type

from ..config import VllmConfig
from ..executor.abstract import Executor
from ..logger import init_logger
from .core import EngineCore, EngineDeadError
from .request import Request, RequestStatus
from .scheduler import EngineCoreOutput, SchedulerStats

logger = init_logger(__name__)

_STATS_EVERY = 16       # steps between stats / timer snapshots sent to the front-end


def _core_main(vllm_config: VllmConfig, executor_class: Optional[Type[Executor]], conn, env: Dict[str, str]) -> None:
    """Child process: own the engine core, serve the pipe until told to stop."""
    os.environ.update(env)
    try:
        from ..launch import _die_with_parent
        _die_with_parent()
    except Exception:  # noqa: BLE001
        pass
    try:
        core = EngineCore(vllm_config, executor_class)
    except BaseException as e:  # noqa: BLE001
        conn.send(("dead", repr(e)))
        raise
    conn.send(("ready", {"num_gpu_blocks": vllm_config.cache_config.num_gpu_blocks,
                         "max_concurrent_batches": core.executor.max_concurrent_batches}))
    steps = 0
    try:
        while True:
            idle = not core.has_unfinished_requests()
            while conn.poll(0.2 if idle else 0):
                msg = conn.recv()
                idle = False
                kind = msg[0]
                if kind == "add":
                    try:
                        core.add_request(msg[1])
                    except EngineDeadError:
                        raise
                    except Exception as e:  # noqa: BLE001 - one bad request must not take the core down
                        logger.warning("request %s rejected: %r", msg[1].request_id, e)
                        conn.send(("reject", msg[1].request_id, repr(e)))
                elif kind == "abort":
                    try:
                        core.abort_requests(msg[1])
                    except Exception as e:  # noqa: BLE001
                        logger.warning("abort of %s failed: %r", msg[1], e)
                elif kind == "finish":
                    try:
                        core.finish_requests(msg[1], RequestStatus.FINISHED_STOPPED)
                    except Exception as e:  # noqa: BLE001
                        logger.warning("finish of %s failed: %r", msg[1], e)
                elif kind == "call":
                    _, call_id, name, args = msg
                    try:
                        conn.send(("ret", call_id, True, getattr(core, name)(*args)))
                    except BaseException as e:  # noqa: BLE001
                        conn.send(("ret", call_id, False, repr(e)))
                elif kind == "shutdown":
                    core.shutdown()
                    conn.send(("bye",))
                    return
            if not core.has_unfinished_requests():
                continue
            outs = core.step()
            steps += 1
            extra = None
            if steps % _STATS_EVERY == 0 or not core.has_unfinished_requests():
                extra = (core.stats(), core.timer.summary())
            conn.send(("out", outs, extra))
    except (EOFError, BrokenPipeError, KeyboardInterrupt):
        pass
    except BaseException as e:  # noqa: BLE001
        logger.error("engine core process failed: %r", e)
        try:
            conn.send(("dead", repr(e)))
        except Exception:  # noqa: BLE001
            pass
        raise
    finally:
        try:
            core.shutdown()
        except Exception:  # noqa: BLE001
            pass


class _TimerView:
    def __init__(self) -> None:
        self.snapshot: Dict[str, float] = {}

    def summary(self) -> Dict[str, float]:
        return dict(self.snapshot)


class MPEngineCoreClient:
    """``EngineCore``'s interface, served by a child process."""

    def __init__(self, vllm_config: VllmConfig, executor_class: Optional[Type[Executor]] = None,
                 startup_timeout: float = 3600.0):
        ctx = mp.get_context("spawn")
        self._conn, child = ctx.Pipe(duplex=True)
        # the child must not think it is a rank of somebody's torchrun
        env = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK") if k in os.environ}
        self._proc = ctx.Process(target=_core_main, args=(vllm_config, executor_class, child, env),
                                 name="engine-core", daemon=False)   # it spawns the local workers
        self._proc.start()
        child.close()
        self._live: Set[str] = set()
        self._lock = threading.Lock()          # serialises senders (event-loop calls vs engine thread)
        self._calls: Dict[int, Any] = {}
        self._next_call = 0
        self._stats = SchedulerStats()
        self.timer = _TimerView()
        self.dead = False
        self.remote = True
        if not self._conn.poll(startup_timeout):
            self._kill()
            raise EngineDeadError("engine core process did not come up")
        msg = self._recv()
        if msg[0] != "ready":
            self._kill()
            raise EngineDeadError(f"engine core process failed to start: {msg[1:]}")
        vllm_config.cache_config.num_gpu_blocks = msg[1]["num_gpu_blocks"]
        self.max_concurrent_batches = msg[1]["max_concurrent_batches"]

    # ---------------------------------------------------------------- plumbing
    def _send(self, msg) -> None:
        if self.dead:
            raise EngineDeadError("engine core process is gone")
        try:
            with self._lock:
                self._conn.send(msg)
        except (BrokenPipeError, OSError) as e:
            self.dead = True
            raise EngineDeadError(f"engine core process is gone: {e!r}") from None

    def _recv(self):
        try:
            return self._conn.recv()
        except (EOFError, OSError) as e:
            self.dead = True
            raise EngineDeadError(f"engine core process exited: {e!r}") from None

    def _kill(self) -> None:
        self.dead = True
        if self._proc.is_alive():
            self._proc.terminate()
            self._proc.join(timeout=5)

    # --------------------------------------------------------------- interface
    def add_request(self, request: Request) -> None:
        self._live.add(request.request_id)
        self._send(("add", request))

    def abort_requests(self, request_ids: Iterable[str]) -> List[str]:
        ids = [r for r in request_ids if r in self._live]
        for r in ids:
            self._live.discard(r)
        if ids:
            self._send(("abort", ids))
        return ids

    def finish_requests(self, request_ids: Iterable[str], status: RequestStatus) -> None:
        ids = list(request_ids)
        for r in ids:
            self._live.discard(r)
        if ids:
            self._send(("finish", ids))

    def has_unfinished_requests(self) -> bool:
        return bool(self._live)

    def get_num_unfinished_requests(self) -> int:
        return len(self._live)

    def step(self, timeout: float = 1.0) -> List[EngineCoreOutput]:
        """The next step's outputs; ``[]`` if none arrived within ``timeout``."""
        if not self._live:
            return []
        deadline = time.monotonic() + timeout
        while True:
            left = deadline - time.monotonic()
            if left <= 0 or not self._conn.poll(left):
                if not self._proc.is_alive():
                    self.dead = True
                    raise EngineDeadError("engine core process exited")
                return []
            msg = self._recv()
            kind = msg[0]
            if kind == "out":
                _, outs, extra = msg
                if extra is not None:
                    self._stats, self.timer.snapshot = extra
                for o in outs:
                    if o.finish_reason is not None:
                        self._live.discard(o.request_id)
                return outs
            if kind == "ret":
                self._calls[msg[1]] = msg[2:]
            elif kind == "reject":        # the core refused the request: it ends with an error, nothing else does
                self._live.discard(msg[1])
                return [EngineCoreOutput(request_id=msg[1], new_token_ids=[], finish_reason="error",
                                         stop_reason=msg[2])]
            elif kind == "dead":
                self.dead = True
                raise EngineDeadError(f"engine core process failed: {msg[1]}")

    def _call(self, name: str, *args, timeout: float = 600.0):
        """Run ``EngineCore.<name>(*args)`` in the child.  Only used while no
        request is in flight or from the thread that also calls ``step``."""
        call_id = self._next_call
        self._next_call += 1
        self._send(("call", call_id, name, args))
        deadline = time.monotonic() + timeout
        while call_id not in self._calls:
            left = deadline - time.monotonic()
            if left <= 0 or not self._conn.poll(left):
                raise TimeoutError(f"engine core call {name} timed out")
            msg = self._recv()
            if msg[0] == "ret":
                self._calls[msg[1]] = msg[2:]
            elif msg[0] == "dead":
                self.dead = True
                raise EngineDeadError(f"engine core process failed: {msg[1]}")
            elif msg[0] == "out" and msg[2] is not None:
                self._stats, self.timer.snapshot = msg[2]
            elif msg[0] == "reject":
                self._live.discard(msg[1])
        ok, value = self._calls.pop(call_id)
        if not ok:
            raise RuntimeError(f"engine core call {name} failed: {value}")
        return value

    def stats(self) -> SchedulerStats:
        return self._stats

    def profile(self, is_start: bool = True) -> None:
        self._call("profile", is_start)

    def reset_prefix_cache(self) -> bool:
        return bool(self._call("reset_prefix_cache"))

    def add_lora(self, lora_request) -> bool:
        return bool(self._call("add_lora", lora_request))

    def remove_lora(self, lora_id: int) -> bool:
        return bool(self._call("remove_lora", lora_id))

    def list_loras(self):
        return self._call("list_loras")

    def pin_lora(self, lora_id: int) -> bool:
        return bool(self._call("pin_lora", lora_id))

    def sleep(self, level: int = 1) -> None:
        self._call("sleep", level)

    def wake_up(self, tags=None) -> None:
        self._call("wake_up", tags)

    def check_health(self) -> None:
        if self.dead or not self._proc.is_alive():
            self.dead = True
            raise EngineDeadError("engine core process is gone")

    def shutdown(self) -> None:
        if not self.dead and self._proc.is_alive():
            try:
                self._send(("shutdown",))
                deadline = time.monotonic() + 20
                while time.monotonic() < deadline and self._conn.poll(max(0.0, deadline - time.monotonic())):
                    if self._recv()[0] == "bye":
                        break
            except Exception:  # noqa: BLE001
                pass
        self.dead = True
        self._proc.join(timeout=10)
        if self._proc.is_alive():
            self._proc.terminate()
            self._proc.join(timeout=5)
        try:
            self._conn.close()
        except Exception:  # noqa: BLE001
            pass
