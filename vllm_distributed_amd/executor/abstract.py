"""Executor base class: what the engine needs from "the set of workers".

Subclasses implement ``_init_executor`` and ``collective_rpc``; everything
else (KV sizing, warm-up, health, profiling, sleep/wake) is expressed as
``collective_rpc`` calls by method name, the same contract the reference's
``CustomExecutor`` fulfils for vLLM's ``Executor`` base
(reference ``src/launch.py:45,60,74-77,298-388``).
"""
from __future__ import annotations

import concurrent.futures
from typing import Any, Callable, List, Optional, Type, Union

from ..config import VllmConfig
from ..engine.types import KVCacheConfig, KVCacheSpec, ModelRunnerOutput, SchedulerOutput
from ..logger import init_logger

logger = init_logger(__name__)

FailureCallback = Callable[[], None]


class LazyFuture:
    """Future-like wrapper for in-process executors: the worker has launched the
    step on the device; ``result()`` waits for it (``obj.get()``) on first use."""

    def __init__(self, obj: Any):
        self._obj = obj

    def result(self, timeout: Optional[float] = None) -> Any:
        obj = self._obj
        if getattr(obj, "lazy_step_result", False):
            self._obj = obj = obj.get()
        return obj

    def done(self) -> bool:
        return not getattr(self._obj, "lazy_step_result", False)


class Executor:
    uses_ray: bool = False

    def __init__(self, vllm_config: VllmConfig):
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.cache_config = vllm_config.cache_config
        self.parallel_config = vllm_config.parallel_config
        self.scheduler_config = vllm_config.scheduler_config
        self.device_config = vllm_config.device_config
        self.is_failed = False
        self.failure_callback: Optional[FailureCallback] = None
        try:
            self._init_executor()
        except BaseException:
            try:                      # do not leave worker processes / RPC threads behind
                self.shutdown()
            except Exception:  # noqa: BLE001
                pass
            raise

    # ------------------------------------------------------------- selection
    @staticmethod
    def get_class(vllm_config: VllmConfig) -> Type["Executor"]:
        backend = vllm_config.parallel_config.distributed_executor_backend
        if isinstance(backend, type) and issubclass(backend, Executor):
            return backend
        if backend is None:
            backend = "uni" if vllm_config.parallel_config.world_size == 1 else "distributed"
        if backend == "uni":
            from .uniproc import UniProcExecutor
            return UniProcExecutor
        if backend == "spmd":
            from .spmd import SpmdExecutor
            return SpmdExecutor
        if backend == "distributed":
            from .distributed import DistributedExecutor
            return DistributedExecutor
        raise ValueError(f"unknown distributed_executor_backend {backend!r}")

    # ------------------------------------------------------------- interface
    def _init_executor(self) -> None:
        raise NotImplementedError

    def collective_rpc(self, method: Union[str, Callable], timeout: Optional[float] = None,
                       args: tuple = (), kwargs: Optional[dict] = None, non_block: bool = False,
                       unique_reply_rank: Optional[int] = None) -> List[Any]:
        raise NotImplementedError

    @property
    def max_concurrent_batches(self) -> int:
        # async scheduling keeps one step in flight while the next is scheduled
        return 2 if self.scheduler_config.async_scheduling else 1

    @property
    def output_rank(self) -> int:
        """First TP rank of the last PP stage (reference ``src/launch.py:304-314``)."""
        pc = self.parallel_config
        return pc.world_size - pc.tensor_parallel_size

    def register_failure_callback(self, callback: FailureCallback) -> None:
        if self.is_failed:
            callback()
        else:
            self.failure_callback = callback

    def execute_model(self, scheduler_output: SchedulerOutput, non_block: bool = False,
                      ) -> Union[ModelRunnerOutput, "concurrent.futures.Future[ModelRunnerOutput]"]:
        (output,) = self.collective_rpc("execute_model", args=(scheduler_output,),
                                        unique_reply_rank=self.output_rank, non_block=non_block)
        return output

    # ------------------------------------------------- engine bring-up calls
    def get_kv_cache_specs(self) -> List[KVCacheSpec]:
        return self.collective_rpc("get_kv_cache_spec")

    def determine_available_memory(self) -> List[int]:
        return self.collective_rpc("determine_available_memory")

    def initialize_from_config(self, kv_cache_configs: List[KVCacheConfig]) -> None:
        # every worker gets the same block count (the scheduler has one block space)
        self.collective_rpc("initialize_from_config", args=(kv_cache_configs[0],))
        self.collective_rpc("compile_or_warm_up_model")

    def check_health(self) -> None:
        self.collective_rpc("check_health", timeout=10)

    def profile(self, is_start: bool = True) -> None:
        self.collective_rpc("profile", args=(is_start,))

    def add_lora(self, lora_request) -> bool:
        return all(self.collective_rpc("add_lora", args=(lora_request,)))

    def remove_lora(self, lora_id: int) -> bool:
        return all(self.collective_rpc("remove_lora", args=(lora_id,)))

    def list_loras(self) -> List[int]:
        return sorted(set().union(*[set(x or ()) for x in self.collective_rpc("list_loras")]))

    def pin_lora(self, lora_id: int) -> bool:
        return all(self.collective_rpc("pin_lora", args=(lora_id,)))

    def sleep(self, level: int = 1) -> None:
        self.collective_rpc("sleep", kwargs=dict(level=level))

    def wake_up(self, tags: Optional[List[str]] = None) -> None:
        self.collective_rpc("wake_up", kwargs=dict(tags=tags))

    def save_sharded_state(self, path: str, pattern: Optional[str] = None,
                           max_size: Optional[int] = None) -> None:
        self.collective_rpc("save_sharded_state", kwargs=dict(path=path, pattern=pattern, max_size=max_size))

    def execute_dummy_batch(self) -> None:
        self.collective_rpc("execute_dummy_batch")

    def shutdown(self) -> None:
        return None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
