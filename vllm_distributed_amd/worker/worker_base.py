"""Worker wrapper: the object a worker *process* holds before and after the
real ``Worker`` exists.

The head sends ``init_worker(all_kwargs)`` to every process with the kwargs of
*all* ranks; each wrapper picks its own entry by ``rpc_rank``
(reference ``src/launch.py:279-290,510-520``).  The head does not know which
local device a remote process sits on and passes ``local_rank=-1``; the
wrapper substitutes the value it was created with.  Any other method name is
forwarded to the worker, so new worker methods need no executor changes.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

from ..config import VllmConfig
from ..logger import init_logger

logger = init_logger(__name__)


class WorkerWrapperBase:
    def __init__(self, vllm_config: Optional[VllmConfig], rpc_rank: int = 0):
        self.vllm_config = vllm_config
        self.rpc_rank = rpc_rank
        self.worker: Optional[Any] = None

    def update_environment_variables(self, envs: Dict[str, str]) -> None:
        os.environ.update(envs)

    def init_worker(self, all_kwargs: List[Dict[str, Any]]) -> None:
        kwargs = dict(all_kwargs[self.rpc_rank])
        worker_cls = kwargs.pop("worker_cls", None)
        if worker_cls is None:
            from .worker import Worker
            worker_cls = Worker
        elif isinstance(worker_cls, str):
            import importlib
            mod, _, name = worker_cls.rpartition(".")
            worker_cls = getattr(importlib.import_module(mod), name)
        if kwargs.get("vllm_config") is None:
            kwargs["vllm_config"] = self.vllm_config
        self.vllm_config = kwargs["vllm_config"]
        self.worker = worker_cls(**kwargs)

    def __getattr__(self, name: str) -> Any:
        worker = self.__dict__.get("worker")
        if worker is None:
            raise AttributeError(f"worker is not initialised; cannot call {name!r}")
        return getattr(worker, name)


class WorkerWrapper(WorkerWrapperBase):
    """Remembers the true ``local_rank`` of the process it lives in."""

    def __init__(self, vllm_config: Optional[VllmConfig], rpc_rank: int, local_rank: int):
        super().__init__(vllm_config, rpc_rank)
        self.local_rank = local_rank
        logger.info("WorkerWrapper created for rpc_rank %d, local_rank %d", rpc_rank, local_rank)

    def init_worker(self, all_kwargs: List[Dict[str, Any]]) -> None:
        all_kwargs = list(all_kwargs)
        kw = dict(all_kwargs[self.rpc_rank])
        kw["local_rank"] = self.local_rank
        all_kwargs[self.rpc_rank] = kw
        super().init_worker(all_kwargs)
