"""Environment-variable registry.

Every environment variable the framework reads is declared here once, with a
parser and a default, and is read lazily so tests can monkeypatch
``os.environ``.  ``environment_variables`` is the set of names the head copies
to remote workers (the reference copies ``vllm.envs.environment_variables``,
reference ``src/launch.py:198-208``); worker-specific names are excluded by the
executor (reference ``src/launch.py:64-72``).

Variable names keep the reference's spelling where the meaning is the same
(``VLLM_HOST_IP``, ``VLLM_SERVER_PORT``, ``VLLM_EXECUTE_MODEL_TIMEOUT_SECONDS``,
``VLLM_HTTP_TIMEOUT_KEEP_ALIVE``, ``VLLM_PP_LAYER_PARTITION``,
``VLLM_USE_MODELSCOPE``) so an operator's ``.env`` files carry over.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict


def _bool(name: str, default: bool) -> Callable[[], bool]:
    def get() -> bool:
        v = os.environ.get(name)
        if v is None:
            return default
        return v.strip().lower() in ("1", "true", "yes", "on")
    return get


def _int(name: str, default: int) -> Callable[[], int]:
    return lambda: int(os.environ.get(name, default))


def _float(name: str, default: float) -> Callable[[], float]:
    return lambda: float(os.environ.get(name, default))


def _str(name: str, default: Any = None) -> Callable[[], Any]:
    return lambda: os.environ.get(name, default)


environment_variables: Dict[str, Callable[[], Any]] = {
    # --- control plane ---------------------------------------------------
    # IP other nodes use to reach this node (head: rendezvous address).
    "VLLM_HOST_IP": _str("VLLM_HOST_IP", ""),
    # TCP port of the head's worker-registration RPC server.
    "VLLM_SERVER_PORT": _int("VLLM_SERVER_PORT", 30044),
    # Shared secret for the RPC handshake (HMAC-SHA256 challenge/response).
    "VLLM_RPC_SECRET": _str("VLLM_RPC_SECRET", ""),
    # Address the head's RPC server binds to.  Unset: every interface when a real shared
    # secret is configured, loopback only (workers on this host) when it is not.
    "VLLM_RPC_BIND": _str("VLLM_RPC_BIND", None),
    # How the head talks to the workers it spawns itself: "socket" (a Unix socket pair on the event
    # loops of both sides) or "pipe" (multiprocessing.Pipe with helper threads, the reference's choice).
    "VLLM_LOCAL_WORKER_TRANSPORT": _str("VLLM_LOCAL_WORKER_TRANSPORT", "socket"),
    # Seconds a remote worker waits between reconnect attempts.
    "VLLM_REMOTE_RETRY_SECONDS": _float("VLLM_REMOTE_RETRY_SECONDS", 10.0),
    # Cap on GPUs the head uses for its own (forked) workers; -1 = all.
    "VLLM_HEAD_LOCAL_DEVICES": _int("VLLM_HEAD_LOCAL_DEVICES", -1),
    # Seconds the head waits for enough remote workers to register; 0 = forever.
    "VLLM_WORKER_REGISTRATION_TIMEOUT": _float("VLLM_WORKER_REGISTRATION_TIMEOUT", 0.0),
    # Per-call RPC timeouts.
    "VLLM_EXECUTE_MODEL_TIMEOUT_SECONDS": _float("VLLM_EXECUTE_MODEL_TIMEOUT_SECONDS", 300.0),
    "VLLM_RPC_TIMEOUT": _float("VLLM_RPC_TIMEOUT", 600.0),
    "VLLM_HEARTBEAT_SECONDS": _float("VLLM_HEARTBEAT_SECONDS", 0.0),
    # --- http ------------------------------------------------------------
    "VLLM_HTTP_TIMEOUT_KEEP_ALIVE": _int("VLLM_HTTP_TIMEOUT_KEEP_ALIVE", 5),
    "VLLM_API_KEY": _str("VLLM_API_KEY", None),
    # Register /sleep, /wake_up, /start_profile, /stop_profile, /reset_prefix_cache.
    "VLLM_SERVER_DEV_MODE": _bool("VLLM_SERVER_DEV_MODE", False),
    # Run the worker-registration port without a shared secret on a non-loopback address.
    "VLLM_RPC_INSECURE": _bool("VLLM_RPC_INSECURE", False),
    # --- engine ----------------------------------------------------------
    "VLLM_USE_V1": _bool("VLLM_USE_V1", True),
    "VLLM_PP_LAYER_PARTITION": _str("VLLM_PP_LAYER_PARTITION", None),
    "VLLM_USE_MODELSCOPE": _bool("VLLM_USE_MODELSCOPE", False),
    "VLLM_LOGGING_LEVEL": _str("VLLM_LOGGING_LEVEL", "INFO"),
    "VLLM_CACHE_ROOT": _str("VLLM_CACHE_ROOT", os.path.expanduser("~/.cache/vllm_distributed_amd")),
    # --- device / kernels --------------------------------------------------
    # "hip" | "torch" | "auto": which implementation ops.* dispatch to.
    "VDAMD_OPS_BACKEND": _str("VDAMD_OPS_BACKEND", "auto"),
    # "auto" | "rccl" | "torch": data-plane collective backend on GPU.
    "VDAMD_COMM_BACKEND": _str("VDAMD_COMM_BACKEND", "auto"),
    # Enable the peer-mapped direct all-reduce for small messages.
    "VDAMD_CUSTOM_ALL_REDUCE": _bool("VDAMD_CUSTOM_ALL_REDUCE", True),
    # Largest message the peer-mapped path takes (it owns 4x this much uncached memory per rank);
    # larger ones go to RCCL.
    "VDAMD_CUSTOM_ALL_REDUCE_MAX_BYTES": _int("VDAMD_CUSTOM_ALL_REDUCE_MAX_BYTES", 16 << 20),
    # Messages up to this size are summed by every rank in one pass (one barrier, W x the
    # traffic); larger ones are reduce-scattered and gathered (two barriers, 2 x the traffic).
    # "auto": both forms (and RCCL) are timed at bring-up on the machine's own links and the crossover
    # sizes are taken from that; a number pins the threshold.
    "VDAMD_CUSTOM_ALL_REDUCE_ONE_SHOT_BYTES": _str("VDAMD_CUSTOM_ALL_REDUCE_ONE_SHOT_BYTES", "auto"),
    # Bound of every in-kernel wait for a peer (milliseconds).  A rank that gives up sets a flag the
    # worker reads after every step and in check_health; the executor then fails the engine.
    "VDAMD_CUSTOM_ALL_REDUCE_TIMEOUT_MS": _int("VDAMD_CUSTOM_ALL_REDUCE_TIMEOUT_MS", 10000),
    # System-scope release fence also in the barrier that opens a kernel (the data it
    # publishes was written by an earlier kernel into uncached memory): "0", "1", or
    # "auto" = turned on if the start-up self-check fails without it.
    "VDAMD_CUSTOM_ALL_REDUCE_FENCE_START": _str("VDAMD_CUSTOM_ALL_REDUCE_FENCE_START", "auto"),
    # Fused all-reduce + residual add + RMSNorm in the decoder layers (needs the peer-mapped path).
    "VDAMD_FUSED_ALL_REDUCE_NORM": _bool("VDAMD_FUSED_ALL_REDUCE_NORM", True),
    # Micro-batch overlap of TP all-reduces with compute in decode:
    # "auto" (TP>1 and batch >= MIN_BATCH), "0" (off), "force" (also at TP=1; tests).
    "VDAMD_TP_OVERLAP": _str("VDAMD_TP_OVERLAP", "auto"),
    "VDAMD_TP_OVERLAP_MIN_BATCH": _int("VDAMD_TP_OVERLAP_MIN_BATCH", 512),
    # Sequence-parallel residual stream under TP (reduce-scatter -> norm on a
    # 1/TP row shard -> all-gather instead of all-reduce -> replicated norm):
    # "auto" (TP>1, single pipeline stage, >= MIN_TOKENS rows), "0", "force".
    "VDAMD_SEQUENCE_PARALLEL": _str("VDAMD_SEQUENCE_PARALLEL", "auto"),
    "VDAMD_SP_MIN_TOKENS": _int("VDAMD_SP_MIN_TOKENS", 512),
    # Single-GPU stand-in for one rank of a TP group of this size: weights are
    # sharded for rank 0, collectives only produce shapes.  Timing proxy only -
    # the generated tokens are meaningless.
    "VDAMD_SIMULATE_TP": _int("VDAMD_SIMULATE_TP", 1),
    # Rehearsal of multi-rank GPU paths on fewer devices than ranks: ranks are
    # mapped round-robin onto the visible devices and torch.distributed runs on
    # gloo with host staging (RCCL refuses two ranks on one device).
    "VDAMD_SHARE_DEVICE": _bool("VDAMD_SHARE_DEVICE", False),
    # Short-batch weight-streaming GEMM for long-reduction shapes (0 = library GEMM only).
    "VDAMD_SKINNY_GEMM": _bool("VDAMD_SKINNY_GEMM", True),
    # Large-batch (M >= 512) 256 x 256-tile LDS-DMA GEMM with fused SiLU*mul / residual epilogues
    # (0 = library GEMM + separate activation kernel).
    "VDAMD_BIG_GEMM": _bool("VDAMD_BIG_GEMM", True),
    # The K-split slabs of a row-parallel projection (TP = 1) are summed by the add + RMSNorm that follows it
    # instead of by a reduction kernel of their own (one launch and a 16-bit round trip fewer per projection).
    # Up to 8 rows into a very wide output (gate_up with the SiLU epilogue, LM head): the barrier-free
    # weight-streaming GEMM instead of the library.
    "VDAMD_STREAM_GEMM": _bool("VDAMD_STREAM_GEMM", True),
    # DeepSeek's multi-head latent attention keeps the latent (kv_lora_rank + rope values per token) in the
    # cache and attends with the up-projections absorbed into q and the output; 0 = expand to per-head
    # keys and values in front of an ordinary paged cache (the only form for latent widths the kernel
    # is not built for on the GPU).
    "VDAMD_MLA_LATENT": _bool("VDAMD_MLA_LATENT", True),
    # ... and prompts that start in the step (no cached context) attend in the per-head form on the
    # LDS-staged prefill kernel, through a scratch cache, when the expanded head size is at most 128
    # (not DeepSeek-V2 / V3's 192: measured slower there); 0 = the latent kernel for every row.
    "VDAMD_MLA_EXPAND_FRESH_PREFILL": _bool("VDAMD_MLA_EXPAND_FRESH_PREFILL", True),
    "VDAMD_FUSED_REDUCE_NORM": _bool("VDAMD_FUSED_REDUCE_NORM", True),
    # The rotary + KV-cache kernel sums the K-split partial sums of the q/k/v projection itself (decode-sized batches).
    "VDAMD_FUSED_REDUCE_ROPE": _bool("VDAMD_FUSED_REDUCE_ROPE", True),
    # Form silu(gate) * up inside the weight-streaming down projection instead of a separate activation
    # kernel.  Off by default: at 70B batch 1 the two dependent loads in front of the LDS store cost the
    # GEMM more (about 7 us) than the 4.8 us activation kernel they replace (26.05 vs 25.85 ms per token).
    "VDAMD_FUSE_SILU_GEMM": _bool("VDAMD_FUSE_SILU_GEMM", False),
    # Elements of padding after every row of a dense 16-bit weight on the GPU (0 = none): breaks the
    # power-of-two row stride that puts all rows of a tile on the same HBM channels.
    "VDAMD_WEIGHT_ROW_PAD": _int("VDAMD_WEIGHT_ROW_PAD", 128),
    # W4A16 linears: largest batch that runs the fused 4-bit GEMM (larger ones
    # dequantise + library GEMM); kernel variant override (0/1) for experiments.
    "VDAMD_W4_MAX_FUSED_M": _int("VDAMD_W4_MAX_FUSED_M", 256),
    "VDAMD_W4_VARIANT": _str("VDAMD_W4_VARIANT", None),
    # 4-bit layers also keep a dequantised copy for batches above the fused
    # kernels' range (4.5x the weight memory; no per-call dequantisation).
    "VDAMD_W4_DENSE_SHADOW": _bool("VDAMD_W4_DENSE_SHADOW", False),
    # Time the library's candidate kernels for every decode GEMM shape at warm-up.
    "VDAMD_TUNE_GEMM": _bool("VDAMD_TUNE_GEMM", False),
    "VDAMD_TUNE_GEMM_FILE": _str("VDAMD_TUNE_GEMM_FILE", os.path.expanduser("~/.cache/vllm_distributed_amd/gemm_tuning.csv")),
    # MoE decode: the second expert GEMM forms silu(gate) * up while it loads its operand (no activation launch).
    # Off by default: every workgroup of the GEMM repeats the activation of its rows, which costs more than
    # the launch it saves (Qwen3-Coder-480B AWQ on one MI355X: batch 1 8.07 -> 8.47 ms, batch 16 28.4 -> 34.4 ms).
    "VDAMD_MOE_FUSED_SILU": _bool("VDAMD_MOE_FUSED_SILU", False),
    # MoE: average rows per expert (4-bit experts; twice that for 16-bit ones) up to which a step
    # runs the grouped HIP GEMMs; above it each expert gets its own library GEMM.
    "VDAMD_MOE_GROUPED_MAX_ROWS": _int("VDAMD_MOE_GROUPED_MAX_ROWS", 512),
    # Measure the time spent in collectives per step (event pairs around each call; eager steps only).
    "VDAMD_TIME_COLLECTIVES": _bool("VDAMD_TIME_COLLECTIVES", False),
    # Capture decode steps in hipGraphs.
    "VDAMD_USE_HIP_GRAPH": _bool("VDAMD_USE_HIP_GRAPH", True),
    # Assert identical collective sequence numbers on all ranks (debug).
    "VDAMD_CHECK_COLLECTIVE_ORDER": _bool("VDAMD_CHECK_COLLECTIVE_ORDER", False),
    # Emit roctx ranges / per-step timing.
    "VDAMD_TRACE": _bool("VDAMD_TRACE", False),
    "VDAMD_PROFILE_DIR": _str("VDAMD_PROFILE_DIR", "/tmp/vdamd_profile"),
    # Test-only fault hook: "rank:step" => that rank exits at that step.
    "VDAMD_FAULT_DIE_AT": _str("VDAMD_FAULT_DIE_AT", None),
    # Test-only fault hook: "rank:step" => that rank stops issuing work at that step (a hung device).
    "VDAMD_FAULT_STALL_AT": _str("VDAMD_FAULT_STALL_AT", None),
    # Data-plane NIC selection, honoured by RCCL / gloo.
    "NCCL_SOCKET_IFNAME": _str("NCCL_SOCKET_IFNAME", None),
    "GLOO_SOCKET_IFNAME": _str("GLOO_SOCKET_IFNAME", None),
    "NCCL_DEBUG": _str("NCCL_DEBUG", None),
    "HSA_ENABLE_IPC_MODE_LEGACY": _str("HSA_ENABLE_IPC_MODE_LEGACY", None),
}


def __getattr__(name: str) -> Any:
    try:
        return environment_variables[name]()
    except KeyError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None


def __dir__():
    return list(environment_variables.keys())


def is_set(name: str) -> bool:
    return name in os.environ
