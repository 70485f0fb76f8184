"""Per-request sampling parameters (OpenAI-compatible subset)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Union


@dataclass
class SamplingParams:
    n: int = 1
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1                 # -1 / 0 = disabled
    min_p: float = 0.0
    presence_penalty: float = 0.0
    frequency_penalty: float = 0.0
    repetition_penalty: float = 1.0
    seed: Optional[int] = None
    max_tokens: Optional[int] = 16
    min_tokens: int = 0
    stop: Union[None, str, List[str]] = None
    stop_token_ids: Optional[List[int]] = None
    ignore_eos: bool = False
    logprobs: Optional[int] = None
    skip_special_tokens: bool = True
    include_stop_str_in_output: bool = False
    detokenize: bool = True

    def __post_init__(self) -> None:
        if self.temperature < 0:
            raise ValueError("temperature must be >= 0")
        if not 0 < self.top_p <= 1:
            raise ValueError("top_p must be in (0, 1]")
        if self.top_k is None or self.top_k == 0:
            self.top_k = -1
        if self.top_k < -1:
            raise ValueError("top_k must be -1 or >= 1")
        if not 0 <= self.min_p <= 1:
            raise ValueError("min_p must be in [0, 1]")
        if self.max_tokens is not None and self.max_tokens < 1:
            raise ValueError("max_tokens must be >= 1")
        if self.n < 1:
            raise ValueError("n must be >= 1")
        if self.repetition_penalty <= 0:
            raise ValueError("repetition_penalty must be > 0")
        if isinstance(self.stop, str):
            self.stop = [self.stop]
        elif self.stop is None:
            self.stop = []
        else:
            self.stop = list(self.stop)
        self.stop_token_ids = list(self.stop_token_ids or [])
        if self.temperature < 1e-5:      # greedy: the other knobs are no-ops
            self.temperature = 0.0

    @property
    def greedy(self) -> bool:
        return self.temperature == 0.0

    @property
    def has_penalties(self) -> bool:
        return (self.presence_penalty != 0.0 or self.frequency_penalty != 0.0
                or self.repetition_penalty != 1.0)

    def clone(self) -> "SamplingParams":
        import copy
        return copy.deepcopy(self)
