"""User-facing generation results."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional


@dataclass
class CompletionOutput:
    index: int
    text: str
    token_ids: List[int]
    cumulative_logprob: Optional[float] = None
    logprobs: Optional[List[Dict[int, float]]] = None
    finish_reason: Optional[str] = None      # stop | length | abort
    stop_reason: Optional[object] = None     # the stop string / token id that fired

    @property
    def finished(self) -> bool:
        return self.finish_reason is not None


@dataclass
class RequestMetrics:
    arrival_time: float = 0.0
    first_scheduled_time: Optional[float] = None
    first_token_time: Optional[float] = None
    finished_time: Optional[float] = None
    num_prompt_tokens: int = 0
    num_generation_tokens: int = 0
    num_cached_tokens: int = 0

    @property
    def ttft(self) -> Optional[float]:
        if self.first_token_time is None:
            return None
        return self.first_token_time - self.arrival_time


@dataclass
class RequestOutput:
    request_id: str
    prompt: Optional[str]
    prompt_token_ids: List[int]
    outputs: List[CompletionOutput]
    finished: bool
    metrics: Optional[RequestMetrics] = None
    # tokens produced since the previous RequestOutput for this request
    new_token_ids: List[int] = field(default_factory=list)
    new_text: str = ""
    # one {token_id: logprob} per prompt token (None for the first), when ``prompt_logprobs`` was requested
    prompt_logprobs: Optional[List[Optional[Dict[int, float]]]] = None
    # pooled final hidden state of the prompt (requests with ``SamplingParams.pooling``), once finished
    pooled: Optional[List[float]] = None
