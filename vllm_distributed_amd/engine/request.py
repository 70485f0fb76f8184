"""Engine-side request state."""
from __future__ import annotations

import enum
import time
from typing import List, Optional

from ..outputs import RequestMetrics
from ..sampling_params import SamplingParams


class RequestStatus(enum.IntEnum):
    WAITING = 0
    RUNNING = 1
    PREEMPTED = 2
    FINISHED_STOPPED = 3
    FINISHED_LENGTH = 4
    FINISHED_ABORTED = 5

    @property
    def finished(self) -> bool:
        return self >= RequestStatus.FINISHED_STOPPED

    @property
    def finish_reason(self) -> Optional[str]:
        return {RequestStatus.FINISHED_STOPPED: "stop", RequestStatus.FINISHED_LENGTH: "length",
                RequestStatus.FINISHED_ABORTED: "abort"}.get(self)


class Request:
    __slots__ = ("request_id", "prompt_token_ids", "sampling_params", "eos_token_id",
                 "arrival_time", "priority", "status", "output_token_ids", "all_token_ids",
                 "num_computed_tokens", "num_cached_tokens", "stop_reason", "metrics",
                 "block_hashes", "num_preemptions", "max_tokens", "prompt", "num_output_placeholders",
                 "lora_request", "prompt_logprobs", "spec_token_ids", "plain_stop", "mm_inputs")

    def __init__(self, request_id: str, prompt_token_ids: List[int],
                 sampling_params: SamplingParams, eos_token_id: Optional[int] = None,
                 arrival_time: Optional[float] = None, priority: int = 0,
                 prompt: Optional[str] = None, lora_request=None, mm_inputs=None):
        self.request_id = request_id
        self.lora_request = lora_request
        # images of a vision-language request (multimodal.MultiModalInput: placeholder span + pixels), or None
        self.mm_inputs = mm_inputs or None
        self.prompt_token_ids = list(prompt_token_ids)
        self.sampling_params = sampling_params
        self.eos_token_id = eos_token_id
        self.arrival_time = time.time() if arrival_time is None else arrival_time
        self.priority = priority
        self.prompt = prompt
        self.status = RequestStatus.WAITING
        self.output_token_ids: List[int] = []
        self.all_token_ids: List[int] = list(prompt_token_ids)
        self.num_computed_tokens = 0
        self.num_cached_tokens = -1
        self.stop_reason: Optional[object] = None
        self.block_hashes: List[int] = []
        self.num_preemptions = 0
        # position -> {token_id: logprob}, filled chunk by chunk; emitted once with the first output token
        self.prompt_logprobs: Optional[dict] = {} if sampling_params.prompt_logprobs is not None else None
        # tokens scheduled to be sampled whose values are not known yet (async scheduling)
        self.num_output_placeholders = 0
        # what the workers' draft model proposed after this request's latest token (draft-model speculation)
        self.spec_token_ids: List[int] = []
        self.max_tokens = sampling_params.max_tokens
        # the only ways this request can end are its length limits: no end-of-sequence or stop-token test per
        # sampled token (the scheduler's per-step loops take a short path for such requests)
        self.plain_stop = bool((sampling_params.ignore_eos or eos_token_id is None)
                               and not sampling_params.stop_token_ids)
        self.metrics = RequestMetrics(arrival_time=self.arrival_time,
                                      num_prompt_tokens=len(prompt_token_ids))

    @property
    def num_prompt_tokens(self) -> int:
        return len(self.prompt_token_ids)

    @property
    def num_tokens(self) -> int:
        return len(self.all_token_ids)

    @property
    def num_output_tokens(self) -> int:
        return len(self.output_token_ids)

    def append_output_token(self, token_id: int) -> None:
        self.output_token_ids.append(token_id)
        self.all_token_ids.append(token_id)

    @property
    def finished(self) -> bool:
        return self.status.finished

    def __repr__(self) -> str:
        return (f"Request({self.request_id}, status={self.status.name}, prompt={self.num_prompt_tokens}, "
                f"out={self.num_output_tokens}, computed={self.num_computed_tokens})")
