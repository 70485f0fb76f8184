"""Engine arguments: dataclass <-> CLI <-> ``VllmConfig``.

The flag names follow the CLI the reference drives through vLLM's parser
(``.env.server:11``: ``-tp``, ``-pp``, ``--max-model-len``,
``--served-model-name`` ...; reference ``src/launch.py:399``:
``AsyncEngineArgs.from_cli_args``).
"""
from __future__ import annotations

import argparse
import dataclasses
import json
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

from ..logger import init_logger
from ..config import (CacheConfig, CompilationConfig, DeviceConfig, KVTransferConfig, LoRAConfig, ModelConfig,
                      SpeculativeConfig,
                      ObservabilityConfig, ParallelConfig, SchedulerConfig, VllmConfig)

logger = init_logger(__name__)


@dataclass
class EngineArgs:
    model: str = "llama-tiny"
    tokenizer: Optional[str] = None
    served_model_name: Optional[List[str]] = None
    dtype: str = "auto"
    seed: int = 0
    max_model_len: Optional[int] = None
    load_format: str = "auto"
    quantization: Optional[str] = None
    trust_remote_code: bool = False
    revision: Optional[str] = None
    generation_config: str = "auto"
    hf_overrides: Optional[Dict[str, Any]] = None
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1
    data_parallel_size: int = 1
    enable_expert_parallel: bool = False
    distributed_executor_backend: Any = None
    disable_custom_all_reduce: bool = False
    worker_cls: Optional[str] = None
    block_size: Optional[int] = None
    gpu_memory_utilization: float = 0.90
    num_gpu_blocks_override: Optional[int] = None
    enable_prefix_caching: bool = True
    kv_cache_dtype: str = "auto"
    max_num_batched_tokens: Optional[int] = None
    max_num_seqs: int = 256
    enable_chunked_prefill: bool = True
    async_scheduling: bool = False
    scheduling_policy: str = "fcfs"
    long_prefill_token_threshold: int = 0
    device: str = "auto"
    enforce_eager: bool = False
    hip_graph_sizes: Optional[List[int]] = None
    max_capture_size: int = 512
    kv_transfer_config: Optional[Dict[str, Any]] = None
    speculative_config: Optional[Any] = None          # dict or JSON text: {"method": "ngram", ...} or {"model": <draft>, ...}
    disable_log_stats: bool = False
    enable_lora: bool = False
    max_loras: int = 1
    max_lora_rank: int = 16

    # ---------------------------------------------------------------- CLI
    @staticmethod
    def add_cli_args(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
        g = parser.add_argument_group("engine")
        g.add_argument("--model", type=str, default=EngineArgs.model)
        g.add_argument("--tokenizer", type=str, default=None)
        g.add_argument("--served-model-name", nargs="+", default=None)
        g.add_argument("--dtype", type=str, default="auto",
                       choices=["auto", "bfloat16", "float16", "float32", "half", "float", "bf16", "fp16"])
        g.add_argument("--seed", type=int, default=0)
        g.add_argument("--max-model-len", type=int, default=None)
        g.add_argument("--load-format", type=str, default="auto", choices=["auto", "safetensors", "dummy"])
        g.add_argument("--quantization", "-q", type=str, default=None, choices=[None, "awq", "awq-dequant", "fp8", "nvfp4"])
        g.add_argument("--trust-remote-code", action="store_true")
        g.add_argument("--revision", type=str, default=None)
        g.add_argument("--generation-config", type=str, default="auto",
                       help="'auto': default sampling parameters and extra EOS ids come from the model's "
                            "generation_config.json; 'vllm': ignore that file; or a directory holding one")
        g.add_argument("--hf-overrides", type=json.loads, default=None)
        g.add_argument("--tensor-parallel-size", "-tp", type=int, default=1)
        g.add_argument("--pipeline-parallel-size", "-pp", type=int, default=1)
        g.add_argument("--data-parallel-size", "-dp", type=int, default=1)
        g.add_argument("--enable-expert-parallel", action="store_true")
        g.add_argument("--distributed-executor-backend", type=str, default=None)
        g.add_argument("--disable-custom-all-reduce", action="store_true")
        g.add_argument("--block-size", type=int, default=None, choices=[16, 32, 64, 128])
        g.add_argument("--gpu-memory-utilization", type=float, default=0.90)
        g.add_argument("--num-gpu-blocks-override", type=int, default=None)
        g.add_argument("--enable-prefix-caching", action=argparse.BooleanOptionalAction, default=True)
        g.add_argument("--kv-cache-dtype", type=str, default="auto")
        g.add_argument("--max-num-batched-tokens", type=int, default=None)
        g.add_argument("--max-num-seqs", type=int, default=256)
        g.add_argument("--enable-chunked-prefill", action=argparse.BooleanOptionalAction, default=True)
        g.add_argument("--async-scheduling", action="store_true")
        g.add_argument("--scheduling-policy", type=str, default="fcfs", choices=["fcfs", "priority"])
        g.add_argument("--long-prefill-token-threshold", type=int, default=0)
        g.add_argument("--device", type=str, default="auto", choices=["auto", "cuda", "cpu"])
        g.add_argument("--enforce-eager", action="store_true",
                       help="do not capture decode steps in hipGraphs")
        g.add_argument("--hip-graph-sizes", "--cuda-graph-sizes", type=int, nargs="*", default=None)
        g.add_argument("--max-capture-size", type=int, default=512)
        g.add_argument("--kv-transfer-config", type=json.loads, default=None)
        g.add_argument("--speculative-config", type=json.loads, default=None,
                       help='speculative decoding: prompt lookup, e.g. {"method": "ngram", "num_speculative_tokens": 4, '
                            '"prompt_lookup_max": 4, "prompt_lookup_min": 2}, or a draft model with the target\'s '
                            'vocabulary, e.g. {"model": "<name or path>", "num_speculative_tokens": 4}')
        g.add_argument("--disable-log-stats", action="store_true")
        g.add_argument("--enable-lora", action="store_true", help="serve LoRA adapters next to the base model")
        g.add_argument("--max-loras", type=int, default=1, help="adapters resident (and usable in one batch) at once")
        g.add_argument("--max-lora-rank", type=int, default=16)
        g.add_argument("--rope-scaling", type=json.loads, default=None,
                       help='JSON, e.g. {"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 32768}')
        g.add_argument("--rope-theta", type=float, default=None)
        # vLLM options that tune machinery this build does not have (CPU swap space, torch.compile levels,
        # CUDA-graph padding limits, request logging...): accepted so an existing command line keeps
        # working, reported once at start-up, otherwise without effect
        c = parser.add_argument_group("accepted for vLLM command-line compatibility (no effect)")
        for flag in COMPAT_FLAGS_NO_VALUE:
            c.add_argument(flag, action="store_true", dest="compat_" + flag.lstrip("-").replace("-", "_"),
                           help=argparse.SUPPRESS)
        for flag in COMPAT_FLAGS_ONE_VALUE:
            c.add_argument(*flag.split("|"), type=str, default=None,
                           dest="compat_" + flag.split("|")[0].lstrip("-").replace("-", "_"), help=argparse.SUPPRESS)
        return parser

    @classmethod
    def from_cli_args(cls, args: argparse.Namespace) -> "EngineArgs":
        names = [f.name for f in dataclasses.fields(cls)]
        self = cls(**{n: getattr(args, n) for n in names if hasattr(args, n)})
        overrides = dict(self.hf_overrides or {})
        if getattr(args, "rope_scaling", None):
            overrides["rope_scaling"] = args.rope_scaling
        if getattr(args, "rope_theta", None):
            overrides["rope_theta"] = args.rope_theta
        self.hf_overrides = overrides or None
        ignored = sorted("--" + k[len("compat_"):].replace("_", "-") for k, v in vars(args).items()
                         if k.startswith("compat_") and v not in (None, False))
        if ignored:
            logger.warning("vLLM options without effect in this build: %s", " ".join(ignored))
        return self

    # ------------------------------------------------------------- config
    def create_engine_config(self) -> VllmConfig:
        model_config = ModelConfig(
            model=self.model, tokenizer=self.tokenizer, served_model_name=self.served_model_name,
            dtype=self.dtype, seed=self.seed, max_model_len=self.max_model_len,
            load_format=self.load_format, quantization=self.quantization,
            trust_remote_code=self.trust_remote_code, revision=self.revision,
            generation_config=self.generation_config,
            hf_overrides=self.hf_overrides or {})
        device_config = DeviceConfig(self.device)
        block_size = self.block_size or (64 if device_config.device == "cuda" else 16)
        cache_config = CacheConfig(block_size=block_size,
                                   gpu_memory_utilization=self.gpu_memory_utilization,
                                   num_gpu_blocks_override=self.num_gpu_blocks_override,
                                   enable_prefix_caching=self.enable_prefix_caching,
                                   cache_dtype=self.kv_cache_dtype)
        parallel_config = ParallelConfig(
            tensor_parallel_size=self.tensor_parallel_size,
            pipeline_parallel_size=self.pipeline_parallel_size,
            data_parallel_size=self.data_parallel_size,
            enable_expert_parallel=self.enable_expert_parallel,
            distributed_executor_backend=self.distributed_executor_backend,
            disable_custom_all_reduce=self.disable_custom_all_reduce, worker_cls=self.worker_cls)
        if parallel_config.data_parallel_size != 1:
            raise ValueError("data parallelism inside one engine is not supported: world_size must "
                             "equal tensor_parallel_size x pipeline_parallel_size; run one engine per replica")
        scheduler_config = SchedulerConfig(
            max_num_batched_tokens=self.max_num_batched_tokens, max_num_seqs=self.max_num_seqs,
            max_model_len=model_config.max_model_len,
            enable_chunked_prefill=self.enable_chunked_prefill,
            async_scheduling=self.async_scheduling, policy=self.scheduling_policy,
            long_prefill_token_threshold=self.long_prefill_token_threshold)
        compilation_config = CompilationConfig(use_hip_graph=not self.enforce_eager,
                                               capture_sizes=self.hip_graph_sizes,
                                               max_capture_size=self.max_capture_size)
        kvt = KVTransferConfig(**self.kv_transfer_config) if self.kv_transfer_config else None
        lora = LoRAConfig(self.max_loras, self.max_lora_rank) if self.enable_lora else None
        return VllmConfig(model_config=model_config, cache_config=cache_config,
                          parallel_config=parallel_config, scheduler_config=scheduler_config,
                          device_config=device_config, compilation_config=compilation_config,
                          observability_config=ObservabilityConfig(), kv_transfer_config=kvt,
                          lora_config=lora,
                          speculative_config=SpeculativeConfig.from_value(self.speculative_config))


COMPAT_FLAGS_NO_VALUE = (
    "--disable-log-requests", "--enable-log-requests", "--no-enable-log-requests", "--enable-sleep-mode",
    "--disable-sliding-window", "--disable-cascade-attn", "--enable-request-id-headers",
    "--return-tokens-as-token-ids", "--disable-fastapi-docs", "--enable-prompt-tokens-details",
    "--enable-server-load-tracking", "--disable-hybrid-kv-cache-manager", "--enable-reasoning",
    "--disable-async-output-proc", "--multi-step-stream-outputs", "--use-v2-block-manager",
    "--fully-sharded-loras")
COMPAT_FLAGS_ONE_VALUE = (
    "--swap-space", "--cpu-offload-gb", "--download-dir", "--tokenizer-mode", "--max-seq-len-to-capture",
    "--max-parallel-loading-workers", "--compilation-config|-O", "--guided-decoding-backend",
    "--structured-outputs-config", "--response-role", "--root-path",
    "--chat-template-content-format", "--limit-mm-per-prompt", "--tokenizer-revision", "--code-revision",
    "--config-format", "--max-num-partial-prefills", "--max-long-partial-prefills", "--num-scheduler-steps",
    "--preemption-mode",
    "--mm-processor-kwargs", "--override-generation-config", "--logits-processor-pattern", "--otlp-traces-endpoint",
    "--worker-extension-cls", "--model-impl", "--task", "--runner", "--convert", "--tokenizer-pool-size",
    "--num-lookahead-slots", "--scheduler-delay-factor", "--max-cpu-loras",
    "--api-server-count", "--middleware", "--max-log-len")

AsyncEngineArgs = EngineArgs
