"""Per-GPU worker: binds a device, joins the process groups, owns the model
runner.  Methods are invoked by name through the executor's
``collective_rpc`` (reference ``src/launch.py:290-292,330,387`` and the
additional calls the engine issues: ``get_kv_cache_spec``,
``determine_available_memory``, ``initialize_from_config``,
``compile_or_warm_up_model``, ``execute_dummy_batch``, ``profile``,
``sleep``/``wake_up``, ``save_sharded_state``).
"""
from __future__ import annotations

import gc
import os
from typing import Any, Dict, List, Optional

import torch

from .. import envs
from ..config import VllmConfig
from ..engine.types import KVCacheConfig, KVCacheSpec, ModelRunnerOutput, SchedulerOutput
from ..logger import init_logger
from ..parallel import state as pstate
from .model_runner import ModelRunner

logger = init_logger(__name__)


class Worker:
    def __init__(self, vllm_config: VllmConfig, local_rank: int, rank: int,
                 distributed_init_method: Optional[str], is_driver_worker: bool = False):
        self.vllm_config = vllm_config
        self.parallel_config = vllm_config.parallel_config
        self.local_rank = local_rank
        self.rank = rank
        self.distributed_init_method = distributed_init_method
        self.is_driver_worker = is_driver_worker
        self.device: Optional[torch.device] = None
        self.model_runner: Optional[ModelRunner] = None
        self._profiler = None
        self._fault = self._parse_fault(envs.VDAMD_FAULT_DIE_AT)
        self._stall = self._parse_fault(envs.VDAMD_FAULT_STALL_AT)
        self._steps = 0
        self._sleeping = False

    # ------------------------------------------------------------------ init
    def _parse_fault(self, spec: Optional[str]) -> Optional[int]:
        """Test-only fault hooks ``VDAMD_FAULT_DIE_AT`` / ``VDAMD_FAULT_STALL_AT`` = ``rank:step``."""
        if not spec:
            return None
        try:
            r, s = spec.split(":")
            return int(s) if int(r) == self.rank else None
        except ValueError:
            return None

    def init_device(self) -> None:
        dev_type = self.vllm_config.device_config.device
        if dev_type == "cuda":
            # one process per GPU; local_rank indexes the *visible* devices
            index, backend = self.local_rank, "nccl"      # RCCL on ROCm
            if envs.VDAMD_SHARE_DEVICE:
                index, backend = self.local_rank % torch.cuda.device_count(), "gloo"
            torch.cuda.set_device(index)
            self.device = torch.device(f"cuda:{index}")
            gc.collect()
            torch.cuda.empty_cache()
            self.init_free, self.init_total = torch.cuda.mem_get_info()
        else:
            self.device = torch.device("cpu")
            backend = "gloo"
            torch.set_num_threads(max(1, min(8, (os.cpu_count() or 2) // max(1, self.parallel_config.world_size))))
        pc = self.parallel_config
        self._default_loopback_interfaces()
        if (self.distributed_init_method or "").startswith("tcp://"):
            # a launcher's "the agent hosts the store" flag would stop rank 0 from hosting the
            # store of the address the head generated
            os.environ.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        pstate.init_distributed_environment(pc.world_size, self.rank, self.distributed_init_method,
                                            self.local_rank, backend, self.device)
        pstate.initialize_model_parallel(pc.tensor_parallel_size, pc.pipeline_parallel_size)
        self._pstate_generation = pstate.current_generation()
        if dev_type == "cuda" and pc.world_size > 1:
            self._attach_fast_collectives()
        torch.manual_seed(self.vllm_config.model_config.seed)
        self.model_runner = ModelRunner(self.vllm_config, self.device)

    def _default_loopback_interfaces(self) -> None:
        """Single-host rendezvous over 127.0.0.1: pin gloo / RCCL bootstrap sockets
        to ``lo`` unless the operator chose an interface (a container hostname
        that does not resolve otherwise breaks gloo's address discovery)."""
        addr = self.distributed_init_method or ""
        if addr in ("", "env://"):
            addr = os.environ.get("MASTER_ADDR", "")
        if any(x in addr for x in ("127.0.0.1", "localhost", "[::1]")):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

    def _attach_fast_collectives(self) -> None:
        """Direct RCCL binding and the peer-mapped all-reduce, each verified
        against ``torch.distributed`` before it is trusted."""
        from ..parallel.accel import attach_accelerated_collectives
        try:
            attach_accelerated_collectives(self.parallel_config)
        except Exception as e:  # never fatal: torch.distributed carries the traffic
            logger.warning("accelerated collectives unavailable (%r); using torch.distributed", e)

    def load_model(self) -> None:
        self.model_runner.load_model()

    # --------------------------------------------------------------- KV cache
    def get_kv_cache_spec(self) -> KVCacheSpec:
        return self.model_runner.get_kv_cache_spec()

    def determine_available_memory(self) -> int:
        """Bytes this worker can give to the KV cache."""
        cc = self.vllm_config.cache_config
        if self.device.type != "cuda":
            return int(cc.cpu_kv_bytes)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        self.model_runner.profile_run()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated()
        free, total = torch.cuda.mem_get_info()
        non_torch = max(0, (total - free) - torch.cuda.memory_reserved())
        torch.cuda.empty_cache()
        budget = int(total * cc.gpu_memory_utilization)
        # head-room for hipGraph pools, RCCL channels and the direct all-reduce staging
        reserve = 2 << 30
        avail = budget - peak - non_torch - reserve
        logger.info("memory: total %.1f GiB, peak torch %.1f GiB, non-torch %.1f GiB -> KV %.1f GiB",
                    total / 2**30, peak / 2**30, non_torch / 2**30, max(avail, 0) / 2**30)
        return max(int(avail), 0)

    def initialize_from_config(self, kv_cache_config: KVCacheConfig) -> None:
        self.model_runner.initialize_kv_cache(kv_cache_config.num_blocks)

    def compile_or_warm_up_model(self) -> None:
        self.model_runner.warm_up()

    # ---------------------------------------------------------------- execute
    def execute_model(self, scheduler_output: SchedulerOutput) -> Optional[ModelRunnerOutput]:
        self._steps += 1
        if self._fault is not None and self._steps >= self._fault:
            logger.error("fault injection: rank %d exiting at step %d", self.rank, self._steps)
            os._exit(17)
        if self._stall is not None and self._steps >= self._stall:
            logger.error("fault injection: rank %d stops issuing work at step %d", self.rank, self._steps)
            import time
            while True:                 # a hung device: the peers' collectives never see this rank arrive
                time.sleep(3600)
        out = self.model_runner.execute_model(scheduler_output)
        self._raise_if_collective_timed_out()
        return out

    def _raise_if_collective_timed_out(self) -> None:
        """A peer-mapped collective whose wait for a peer ran out has summed whatever it found: the
        step's tokens are wrong and every later one too.  The flag lives in pinned host memory, so
        asking costs a load; raising here fails the step on the head, which stops the engine."""
        for group in (pstate.get_tp_group(),) if pstate.model_parallel_is_initialized() else ():
            car = group.custom_ar
            if car is not None and car.timed_out():
                raise RuntimeError(f"rank {self.rank}: a peer-mapped collective gave up waiting for a rank of the "
                                   f"{group.name} group (VDAMD_CUSTOM_ALL_REDUCE_TIMEOUT_MS); results are invalid")

    def execute_dummy_batch(self) -> None:
        self.model_runner.profile_run()

    def check_health(self) -> bool:
        self._raise_if_collective_timed_out()
        return True

    def synchronize(self) -> float:
        """Wait for everything this worker has queued on its device; returns the wall clock after."""
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize()
        import time
        return time.time()

    def get_stats(self) -> Dict[str, Any]:
        d: Dict[str, Any] = {"rank": self.rank, "steps": self._steps}
        if self.device is not None and self.device.type == "cuda":
            free, total = torch.cuda.mem_get_info()
            d.update(mem_free=free, mem_total=total, mem_torch=torch.cuda.memory_allocated())
        from .. import ops
        d["attention_calls"] = dict(ops.attention_calls)
        mr = self.model_runner
        if mr is not None:
            d.update(hip_graph_sizes=list(mr.graph_sizes), hip_graph_fallback=mr.graph_fallback_reason,
                     block_size=mr.block_size,
                     tp_forward_paths=dict(getattr(getattr(mr, "model", None), "path_counts", None) or {}))
            tp = getattr(mr, "tp", None)
            if tp is not None:
                car = tp.custom_ar
                if car is not None:
                    d.update(tp_direct_one_shot_bytes=car.one_shot_bytes, tp_direct_use_bytes=car.use_bytes,
                             tp_direct_calibration=getattr(car, "calibration", None),
                             tp_direct_timed_out=car.timed_out())
                d.update(tp_rccl=tp.rccl is not None, tp_direct_all_reduce=tp.custom_ar is not None,
                         tp_backend=(None if tp.device_group is None else
                                     torch.distributed.get_backend(tp.device_group)))
        return d

    # ------------------------------------------------------------- profiling
    def profile(self, is_start: bool = True) -> None:
        """Start/stop a ``torch.profiler`` trace on this worker; the trace is
        written under ``VDAMD_PROFILE_DIR``."""
        if is_start:
            if self._profiler is not None:
                return
            acts = [torch.profiler.ProfilerActivity.CPU]
            if self.device.type == "cuda":
                acts.append(torch.profiler.ProfilerActivity.CUDA)
            out = envs.VDAMD_PROFILE_DIR
            os.makedirs(out, exist_ok=True)
            self._profiler = torch.profiler.profile(
                activities=acts,
                on_trace_ready=torch.profiler.tensorboard_trace_handler(out, f"rank{self.rank}"))
            self._profiler.start()
        elif self._profiler is not None:
            self._profiler.stop()
            self._profiler = None

    # --------------------------------------------------------- sleep / wake
    def sleep(self, level: int = 1) -> None:
        """Release the KV cache (level 1) so the GPU can be lent out."""
        if self._sleeping:
            return
        for attn in self.model_runner.model.attention_layers():
            attn.k_cache = attn.v_cache = None
        self.model_runner.kv_caches = []
        if self.model_runner.draft is not None:
            self.model_runner.draft.release_kv()
        self.model_runner.graphs.clear()
        self.model_runner.graph_ctx.clear()
        self.model_runner.graph_sizes = []
        gc.collect()
        if self.device.type == "cuda":
            torch.cuda.empty_cache()
        self._sleeping = True

    def wake_up(self, tags: Optional[List[str]] = None) -> None:
        if not self._sleeping:
            return
        self.model_runner.initialize_kv_cache(self.model_runner.num_blocks)
        self.model_runner.warm_up()
        self._sleeping = False

    def save_sharded_state(self, path: str, pattern: Optional[str] = None,
                           max_size: Optional[int] = None) -> None:
        """This rank's tensors as it holds them; ``--load-format sharded_state`` reads them back."""
        from ..models.loader import save_sharded_state
        save_sharded_state(self.model_runner.model, path, self.rank, self.vllm_config, pattern)

    # ------------------------------------------------------------------ LoRA
    def _lora(self):
        mgr = self.model_runner.lora_manager
        if mgr is None:
            raise RuntimeError("the engine was started without --enable-lora")
        return mgr

    def add_lora(self, lora_request) -> bool:
        """Make the adapter resident now (requests naming it load it on demand otherwise)."""
        self._lora().add_lora(lora_request, protect=self.model_runner.loras_in_batch())
        return True

    def remove_lora(self, lora_id: int) -> bool:
        # the engine core refuses this while unfinished requests name the adapter
        return self._lora().remove_lora(lora_id)

    def list_loras(self):
        return self._lora().list_loras()

    def pin_lora(self, lora_id: int) -> bool:
        return self._lora().pin_lora(lora_id)

    def shutdown(self) -> None:
        conn = getattr(self.model_runner, "kv_connector", None)
        if conn is not None:
            conn.shutdown()                 # joins the transfer thread, frees device buffers handed out
        self.model_runner = None
        # only the groups this worker set up: an engine built later in the same process owns newer ones
        pstate.destroy_model_parallel(getattr(self, "_pstate_generation", None))
        gc.collect()
