"""SPMD executor for externally launched ranks (``torchrun``).

Every rank runs the *whole* engine - scheduler included - on identical inputs
and owns one in-process worker.  Scheduling is deterministic and every TP
rank samples the same token from the all-gathered logits, so no control-plane
traffic is needed at all: the only communication is the data-plane
collectives inside the model.  Pipeline parallelism is not offered here (its
stages see different outputs); use the distributed executor for PP.

Rank and world size come from ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` /
``MASTER_ADDR`` / ``MASTER_PORT``.
"""
from __future__ import annotations

import os
from typing import Any, Callable, List, Optional, Union

from ..utils import run_method
from ..worker.worker_base import WorkerWrapperBase
from .abstract import Executor, LazyFuture


class SpmdExecutor(Executor):
    def _init_executor(self) -> None:
        pc = self.parallel_config
        assert pc.pipeline_parallel_size == 1, "the SPMD executor supports tensor parallelism only"
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        assert world == pc.world_size, (
            f"WORLD_SIZE ({world}) must equal tensor_parallel_size ({pc.tensor_parallel_size})")
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = os.environ.get("MASTER_PORT", "29500")
        init_method = f"tcp://{addr}:{port}" if world > 1 else None
        self.driver = WorkerWrapperBase(self.vllm_config, rpc_rank=0)
        kwargs = dict(vllm_config=self.vllm_config, local_rank=self.local_rank, rank=self.rank,
                      distributed_init_method="env://" if world > 1 and "MASTER_ADDR" in os.environ
                      else init_method, is_driver_worker=self.rank % pc.tensor_parallel_size == 0)
        if self.parallel_config.worker_cls:
            kwargs["worker_cls"] = self.parallel_config.worker_cls
        self.collective_rpc("init_worker", args=([kwargs],))
        self.collective_rpc("init_device")
        self.collective_rpc("load_model")

    @property
    def output_rank(self) -> int:
        return 0

    def collective_rpc(self, method: Union[str, Callable], timeout: Optional[float] = None,
                       args: tuple = (), kwargs: Optional[dict] = None, non_block: bool = False,
                       unique_reply_rank: Optional[int] = None) -> List[Any]:
        result = run_method(self.driver, method, args, kwargs)
        if method == "load_model":
            # every rank needs the sampled tokens to keep its own scheduler in step
            self.driver.worker.model_runner.report_on_all_ranks = True
            self.driver.worker.model_runner.defer_output = self.scheduler_config.async_scheduling
        if non_block:
            return [LazyFuture(result)]
        return [result.get() if getattr(result, "lazy_step_result", False) else result]

    def determine_available_memory(self) -> List[int]:
        """Every rank runs its own scheduler, so all must size the KV cache
        identically: take the minimum over ranks."""
        (mine,) = self.collective_rpc("determine_available_memory")
        from ..parallel.state import get_world_group
        return [min(get_world_group().all_gather_object(int(mine)))]

    def shutdown(self) -> None:
        drv = getattr(self, "driver", None)
        if drv is not None and drv.worker is not None:
            try:
                drv.worker.shutdown()
            except Exception:
                pass
            drv.worker = None
