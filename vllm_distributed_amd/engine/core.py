"""Engine core: scheduler + executor loop.

``step()`` = schedule -> execute on all workers -> fold sampled tokens back
into the scheduler.  When the executor allows several batches in flight
(pipeline parallelism: one per stage; async scheduling: two - reference
``src/launch.py:298-302``) ``step()`` keeps a queue of futures from the
non-blocking ``execute_model`` path (reference ``src/launch.py:380-381``) and
only blocks on the oldest one when nothing new can be scheduled.
"""
from __future__ import annotations

import time
from collections import deque
from typing import Deque, Dict, Iterable, List, Optional, Tuple, Type

from ..config import VllmConfig
from ..executor.abstract import Executor
from ..logger import init_logger
from ..utils import cdiv
from .request import Request, RequestStatus
from .scheduler import EngineCoreOutput, Scheduler, SchedulerStats
from .types import KVCacheConfig, ModelRunnerOutput, SchedulerOutput

logger = init_logger(__name__)


class EngineDeadError(RuntimeError):
    pass


class StepTimer:
    """Rolling per-step breakdown.

    Head side (host clock): ``schedule``, ``launch`` (handing a step to the executor
    without waiting), ``execute`` (blocked on the oldest step in flight), ``update``.
    Worker side, from the reply of the output rank: ``worker_prepare`` / ``worker_launch``
    (host time building inputs / enqueueing device work), ``worker_forward`` and
    ``worker_sample`` (device time between event stamps), ``worker_device_idle`` (device
    time between the end of the previous step and the start of this one: what host
    and control plane fail to hide), ``rpc_in`` (head send -> arrival at the worker
    process), ``worker_queued`` (waiting behind the previous call on the worker thread;
    healthy when steps are scheduled ahead) and ``rpc_out`` (worker reply -> head future).  With the device saturated,
    forward + sample + device_idle is the step period."""

    WORKER_KEYS = {"prepare": "worker_prepare", "launch": "worker_launch", "forward": "worker_forward",
                   "sample": "worker_sample", "device_idle": "worker_device_idle", "comm": "worker_comm",
                   "rpc_in": "rpc_in", "rpc_out": "rpc_out", "queued": "worker_queued"}

    def __init__(self) -> None:
        self.reset()

    def reset(self) -> None:
        self.n = 0
        self.tot = {"schedule": 0.0, "launch": 0.0, "execute": 0.0, "update": 0.0, "worker_prepare": 0.0,
                    "worker_forward": 0.0, "worker_sample": 0.0, "worker_device_idle": 0.0,
                    "rpc_in": 0.0, "rpc_out": 0.0}

    def add_head(self, **kw: float) -> None:
        for k, v in kw.items():
            self.tot[k] = self.tot.get(k, 0.0) + v

    def add_step(self, out: Optional[ModelRunnerOutput]) -> None:
        """One finished step, with the worker-side timing its output carries."""
        self.n += 1
        wt = (out.timing if out is not None else None) or {}
        for k, name in self.WORKER_KEYS.items():
            if k in wt:
                self.tot[name] = self.tot.get(name, 0.0) + wt[k]

    def summary(self) -> Dict[str, float]:
        n = max(self.n, 1)
        d = {k + "_ms": 1e3 * v / n for k, v in self.tot.items()}
        d["steps"] = self.n
        d["rpc_overhead_ms"] = d["rpc_in_ms"] + d["rpc_out_ms"]
        d["device_period_ms"] = d["worker_forward_ms"] + d["worker_sample_ms"] + d["worker_device_idle_ms"]
        return d


class EngineCore:
    def __init__(self, vllm_config: VllmConfig, executor_class: Optional[Type[Executor]] = None):
        self.vllm_config = vllm_config
        executor_class = executor_class or Executor.get_class(vllm_config)
        t0 = time.time()
        self.executor = executor_class(vllm_config)
        self.dead = False
        self.executor.register_failure_callback(self._on_executor_failure)
        num_blocks = self._initialize_kv_caches()
        pp = vllm_config.parallel_config.pipeline_parallel_size
        kv_connector = None
        if vllm_config.kv_transfer_config is not None:
            from ..executor.kv_transfer import make_scheduler_connector
            kv_connector = make_scheduler_connector(vllm_config)
        self.scheduler = Scheduler(vllm_config.scheduler_config, vllm_config.cache_config, num_blocks,
                                   include_new_token_ids=pp > 1, kv_connector=kv_connector,
                                   max_loras=vllm_config.lora_config.max_loras if vllm_config.lora_config else 0,
                                   speculative_config=vllm_config.speculative_config,
                                   sliding_window=None if kv_connector is not None else self.kv_window,
                                   pipeline_depth=pp)
        self.max_concurrent = max(1, self.executor.max_concurrent_batches)
        self._tune_gc()
        self.batch_queue: Deque[Tuple[object, SchedulerOutput]] = deque()
        self.timer = StepTimer()
        logger.info("engine core ready in %.1fs (%d KV blocks, %d batches in flight)",
                    time.time() - t0, num_blocks, self.max_concurrent)

    # ----------------------------------------------------------------- setup
    @staticmethod
    def _tune_gc() -> None:
        """A step of a large batch allocates tens of thousands of short-lived, acyclic objects (one
        output record per request).  With the default thresholds that is dozens of young-generation
        passes per step and, every few steps, a full pass over every object the process holds: half of
        the scheduler's time at 16k sequences.  What exists now (model config, tokenizer tables) is moved
        out of the collector's sight, and the young generation is sized for a whole step."""
        import gc
        gc.collect()
        gc.freeze()
        gc.set_threshold(200000, 20, 50)

    def _initialize_kv_caches(self) -> int:
        cc = self.vllm_config.cache_config
        specs = self.executor.get_kv_cache_specs()
        avail = self.executor.determine_available_memory()
        if cc.num_gpu_blocks_override is not None:
            num_blocks = int(cc.num_gpu_blocks_override)
        else:
            num_blocks = min(int(a // s.bytes_per_block) for a, s in zip(avail, specs))
            # no point holding more than every sequence at full length
            cap = self.vllm_config.scheduler_config.max_num_seqs * cdiv(
                self.vllm_config.model_config.max_model_len, cc.block_size)
            num_blocks = min(num_blocks, cap)
        need = cdiv(self.vllm_config.model_config.max_model_len, cc.block_size)
        if num_blocks < need:
            raise RuntimeError(
                f"KV cache too small: {num_blocks} blocks of {cc.block_size} tokens, but one "
                f"sequence of max_model_len={self.vllm_config.model_config.max_model_len} needs {need}; "
                f"lower --max-model-len or raise --gpu-memory-utilization")
        cc.num_gpu_blocks = num_blocks
        # every layer of every worker windowed: blocks that slid out of the widest window are dead
        wins = [getattr(s, "sliding_window", None) for s in specs]
        self.kv_window = max(wins) if wins and all(w is not None for w in wins) else None
        self.executor.initialize_from_config([KVCacheConfig(num_blocks=num_blocks)] * len(specs))
        return num_blocks

    def _on_executor_failure(self) -> None:
        logger.error("executor reported a failure: engine is dead")
        self.dead = True

    # -------------------------------------------------------------- requests
    def add_request(self, request: Request) -> None:
        self._check_alive()
        self.scheduler.add_request(request)

    def abort_requests(self, request_ids: Iterable[str]) -> List[str]:
        return self.scheduler.abort_requests(request_ids)

    def finish_requests(self, request_ids: Iterable[str], status: RequestStatus) -> None:
        self.scheduler.finish_requests(request_ids, status)

    def has_unfinished_requests(self) -> bool:
        return self.scheduler.has_unfinished_requests() or bool(self.batch_queue)

    def _check_alive(self) -> None:
        if self.dead or self.executor.is_failed:
            raise EngineDeadError("the executor failed; the engine cannot continue")

    # ------------------------------------------------------------------ step
    def step(self) -> List[EngineCoreOutput]:
        self._check_alive()
        if self.max_concurrent > 1:
            return self._step_with_batch_queue()
        if not self.scheduler.has_requests():
            return []
        t0 = time.perf_counter()
        so = self.scheduler.schedule()
        t1 = time.perf_counter()
        if not self._worth_running(so):
            return []
        out = self.executor.execute_model(so)
        t2 = time.perf_counter()
        res = self.scheduler.update_from_output(so, out)
        t3 = time.perf_counter()
        self.timer.add_head(schedule=t1 - t0, execute=t2 - t1, update=t3 - t2)
        self.timer.add_step(out)
        return res

    @staticmethod
    def _worth_running(so: SchedulerOutput) -> bool:
        """A step with nothing scheduled still goes to the workers when it carries news for them (finished or
        preempted requests) or when KV transfers are in flight: the workers report a finished transfer with
        the output of the next step they run, so an otherwise idle engine polls (a couple of ms apart)."""
        if not so.is_empty or so.finished_req_ids or so.preempted_req_ids:
            return True
        meta = so.kv_connector_metadata
        if meta and (meta.get("pending") or meta.get("load") or meta.get("save")):
            time.sleep(0.002)
            return True
        return False

    def _step_with_batch_queue(self) -> List[EngineCoreOutput]:
        """Fill the pipeline before draining it: schedule a new batch whenever
        a slot is free and something is schedulable; otherwise wait for the
        oldest in-flight batch."""
        if len(self.batch_queue) < self.max_concurrent and self.scheduler.has_requests():
            t0 = time.perf_counter()
            so = self.scheduler.schedule()
            t1 = time.perf_counter()
            if self._worth_running(so):
                fut = self.executor.execute_model(so, non_block=True)
                self.batch_queue.append((fut, so))
                self.timer.add_head(schedule=t1 - t0, launch=time.perf_counter() - t1)
                if len(self.batch_queue) < self.max_concurrent and not so.is_empty:
                    return []            # try to put another batch in flight first
        if not self.batch_queue:
            return []
        fut, so = self.batch_queue.popleft()
        t0 = time.perf_counter()
        out = fut.result()
        t1 = time.perf_counter()
        res = self.scheduler.update_from_output(so, out)
        t2 = time.perf_counter()
        self.timer.add_head(execute=t1 - t0, update=t2 - t1)   # execute: blocked on the oldest step in flight
        self.timer.add_step(out)
        return res

    # ----------------------------------------------------------------- admin
    def stats(self) -> SchedulerStats:
        return self.scheduler.make_stats()

    def profile(self, is_start: bool = True) -> None:
        self.executor.profile(is_start)

    def reset_prefix_cache(self) -> bool:
        return self.scheduler.reset_prefix_cache()

    def add_lora(self, lora_request) -> bool:
        return self.executor.add_lora(lora_request)

    def remove_lora(self, lora_id: int) -> bool:
        for r in self.scheduler.requests.values():
            if not r.finished and r.lora_request is not None and r.lora_request.lora_int_id == lora_id:
                raise RuntimeError(f"LoRA adapter {lora_id} is in use by request {r.request_id}")
        return self.executor.remove_lora(lora_id)

    def list_loras(self):
        return self.executor.list_loras()

    def pin_lora(self, lora_id: int) -> bool:
        return self.executor.pin_lora(lora_id)

    def sleep(self, level: int = 1) -> None:
        self.executor.sleep(level)

    def wake_up(self, tags=None) -> None:
        self.executor.wake_up(tags)

    def check_health(self) -> None:
        self._check_alive()
        self.executor.check_health()

    def shutdown(self) -> None:
        ex = getattr(self, "executor", None)
        if ex is not None:
            ex.shutdown()
