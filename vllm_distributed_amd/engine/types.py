"""Scheduler -> worker and worker -> scheduler message types.

These are the per-step payloads the executor fans out to every worker and
collects from the output rank (reference ``src/launch.py:322-336``; the
reference treats them as opaque picklable objects from vLLM).  Workers keep a
persistent batch, so only *new* requests carry prompts and sampling params;
running requests are described by deltas (new KV block ids, computed-token
counts, and - for pipeline stages that do not sample - the newly sampled
tokens).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Set

from ..sampling_params import SamplingParams


@dataclass
class NewRequestData:
    req_id: str
    prompt_token_ids: List[int]
    sampling_params: SamplingParams
    block_ids: List[int]
    num_computed_tokens: int        # > 0 on a prefix-cache hit
    # tokens generated before a preemption (recompute path) - part of the prompt now
    output_token_ids: List[int] = field(default_factory=list)
    lora_request: Optional[Any] = None
    mm_inputs: Optional[List[Any]] = None     # multimodal.MultiModalInput per image of the prompt


@dataclass
class CachedRequestData:
    """Parallel lists, one entry per already-known request scheduled this step."""
    req_ids: List[str] = field(default_factory=list)
    new_token_ids: List[List[int]] = field(default_factory=list)
    new_block_ids: List[List[int]] = field(default_factory=list)
    num_computed_tokens: List[int] = field(default_factory=list)

    def __len__(self) -> int:
        return len(self.req_ids)


@dataclass
class SchedulerOutput:
    step_id: int = 0
    scheduled_new_reqs: List[NewRequestData] = field(default_factory=list)
    scheduled_cached_reqs: CachedRequestData = field(default_factory=CachedRequestData)
    # req_id -> tokens to run this step, in batch order
    num_scheduled_tokens: Dict[str, int] = field(default_factory=dict)
    total_num_scheduled_tokens: int = 0
    # requests that left the batch since the previous step (finished/aborted/preempted)
    finished_req_ids: Set[str] = field(default_factory=set)
    preempted_req_ids: Set[str] = field(default_factory=set)
    # KV-transfer work attached by the connector (disaggregated prefill)
    # req_id -> draft tokens to verify this step (speculative decoding); the request is scheduled
    # for 1 + len(drafts) tokens
    scheduled_spec_decode_tokens: Optional[Dict[str, List[int]]] = None
    kv_connector_metadata: Optional[Any] = None

    @property
    def is_empty(self) -> bool:
        return self.total_num_scheduled_tokens == 0


@dataclass
class ModelRunnerOutput:
    step_id: int = 0
    req_ids: List[str] = field(default_factory=list)
    # one list per request: [] while a prompt is only partially prefilled
    sampled_token_ids: List[List[int]] = field(default_factory=list)
    # optional: per request, per sampled token, {token_id: logprob}
    logprobs: Optional[List[Optional[List[Dict[int, float]]]]] = None
    # optional: req_id -> (position of the first prompt token covered, [{token_id: logprob}, ...]) for the
    # prompt tokens this step's prefill chunk predicted (requests with ``prompt_logprobs`` only)
    prompt_logprobs: Optional[Dict[str, Any]] = None
    # req_id -> pooled final hidden state (float32 list) for requests with ``pooling`` that finished their prompt
    pooled: Optional[Dict[str, List[float]]] = None
    # KV-transfer completion sets (aggregated over workers by KVOutputAggregator)
    finished_sending: Optional[Set[str]] = None
    finished_recving: Optional[Set[str]] = None
    # worker-side timing in seconds: prepare / forward / sample
    timing: Optional[Dict[str, float]] = None
    # req_id -> tokens the workers' draft model proposes after the request's newest token
    draft_token_ids: Optional[Dict[str, List[int]]] = None


EMPTY_MODEL_RUNNER_OUTPUT = ModelRunnerOutput()


@dataclass
class KVCacheSpec:
    """Per-worker description of the KV cache its layers need."""
    num_layers: int
    num_kv_heads: int
    head_dim: int
    block_size: int
    dtype_bytes: int
    # the largest attention window of this worker's layers when EVERY one of them has one (keys further
    # back are never read again and their blocks can go back to the pool), else None
    sliding_window: Optional[int] = None
    # bytes per block of further caches held over the same blocks (a draft model's layers)
    extra_bytes_per_block: int = 0

    @property
    def bytes_per_block(self) -> int:
        return (2 * self.num_layers * self.block_size * self.num_kv_heads * self.head_dim * self.dtype_bytes
                + self.extra_bytes_per_block)

    @property
    def bytes_per_token(self) -> int:
        return self.bytes_per_block // self.block_size


@dataclass
class KVCacheConfig:
    num_blocks: int
