"""Single worker in the engine's own process (TP=PP=1)."""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Union

from ..utils import run_method
from ..worker.worker_base import WorkerWrapperBase
from .abstract import Executor, LazyFuture


class UniProcExecutor(Executor):
    def _init_executor(self) -> None:
        assert self.parallel_config.world_size == 1, "UniProcExecutor runs exactly one worker"
        self.driver = WorkerWrapperBase(self.vllm_config, rpc_rank=0)
        kwargs = dict(vllm_config=self.vllm_config, local_rank=0, rank=0,
                      distributed_init_method=None, is_driver_worker=True)
        if self.parallel_config.worker_cls:
            kwargs["worker_cls"] = self.parallel_config.worker_cls
        self.collective_rpc("init_worker", args=([kwargs],))
        self.collective_rpc("init_device")
        self.collective_rpc("load_model")

    def collective_rpc(self, method: Union[str, Callable], timeout: Optional[float] = None,
                       args: tuple = (), kwargs: Optional[dict] = None, non_block: bool = False,
                       unique_reply_rank: Optional[int] = None) -> List[Any]:
        result = run_method(self.driver, method, args, kwargs)
        if method == "load_model":
            self.driver.worker.model_runner.defer_output = self.scheduler_config.async_scheduling
        if non_block:
            return [LazyFuture(result)]
        return [result.get() if getattr(result, "lazy_step_result", False) else result]

    def shutdown(self) -> None:
        drv = getattr(self, "driver", None)
        if drv is not None and drv.worker is not None:
            try:
                drv.worker.shutdown()
            except Exception:
                pass
            drv.worker = None
