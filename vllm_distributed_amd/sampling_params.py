"""Per-request sampling parameters (OpenAI-compatible subset)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Union

# A request-supplied pattern or grammar is compiled by native recursive-descent parsers in the serving
# process: its size is bounded here, its nesting and automaton size inside the compilers.
MAX_CONSTRAINT_BYTES = 64 * 1024


@dataclass
class SamplingParams:
    n: int = 1
    best_of: Optional[int] = None   # sample this many, return the n with the highest cumulative logprob
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
    prompt_logprobs: Optional[int] = None                # logprob of every prompt token (+ this many alternatives)
    skip_special_tokens: bool = True
    include_stop_str_in_output: bool = False
    detokenize: bool = True
    logit_bias: Optional[Dict[int, float]] = None        # token id -> additive bias, clamped to [-100, 100]
    allowed_token_ids: Optional[List[int]] = None        # every other token is masked out
    bad_words: Optional[List[str]] = None                # never generated (tokenised by the engine front-end)
    bad_words_token_ids: Optional[List[List[int]]] = None
    truncate_prompt_tokens: Optional[int] = None         # keep only the last k prompt tokens
    json_object: bool = False                            # constrain the output to one JSON object
    guided_choice: Optional[List[str]] = None            # the output is exactly one of these strings
    # "last" | "mean": return the pooled final hidden state of the prompt (embeddings, scoring)
    # with the request's single output token; the request then ends after that token
    pooling: Optional[str] = None
    guided_regex: Optional[str] = None                   # the output matches this regular expression
    guided_json: Optional[Any] = None                    # the output is JSON satisfying this schema (dict or str)
    guided_grammar: Optional[str] = None                 # the output is a sentence of this EBNF grammar (start rule: root)
    # filled by the engine: the model's EOS id (the only token allowed once a constrained output is complete)
    eos_token_id: Optional[int] = None
    # filled by the engine: EOS + stop token ids, masked until ``min_tokens`` outputs exist
    min_tokens_ban_ids: List[int] = field(default_factory=list)

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
        if self.best_of is not None and self.best_of < self.n:
            raise ValueError("best_of must be >= n")
        if self.seed is not None:
            # clients commonly send -1 for "no seed"; anything else must be usable as a generator seed
            if isinstance(self.seed, bool) or not isinstance(self.seed, int):
                raise ValueError("seed must be an integer")
            if self.seed == -1:
                self.seed = None
            elif not 0 <= self.seed < 2 ** 63:
                raise ValueError("seed must be in [0, 2**63) (or -1 / null for a random seed)")
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
        if self.logit_bias:
            try:                         # JSON object keys arrive as strings
                self.logit_bias = {int(k): min(100.0, max(-100.0, float(v))) for k, v in self.logit_bias.items()}
            except (TypeError, ValueError) as e:
                raise ValueError("logit_bias must map token ids to numbers") from e
            if any(k < 0 for k in self.logit_bias):
                raise ValueError("logit_bias token ids must be >= 0")
        else:
            self.logit_bias = None
        if self.allowed_token_ids is not None:
            self.allowed_token_ids = [int(t) for t in self.allowed_token_ids]
            if not self.allowed_token_ids or min(self.allowed_token_ids) < 0:
                raise ValueError("allowed_token_ids must be a non-empty list of token ids >= 0")
        if isinstance(self.bad_words, str):
            self.bad_words = [self.bad_words]
        if self.bad_words_token_ids is not None:
            self.bad_words_token_ids = [[int(t) for t in w] for w in self.bad_words_token_ids if len(w)]
        if self.guided_choice is not None:
            self.guided_choice = [str(c) for c in self.guided_choice]
            if not self.guided_choice or not all(self.guided_choice):
                raise ValueError("guided_choice must be a non-empty list of non-empty strings")
            if self.json_object:
                raise ValueError("guided_choice and JSON mode exclude each other")
        if self.pooling is not None:
            if self.pooling not in ("last", "mean"):
                raise ValueError("pooling must be 'last' or 'mean'")
            self.max_tokens, self.min_tokens, self.logprobs = 1, 0, None
        if self.guided_json is not None:
            # a schema is served as the regular expression of its compact JSON texts
            if self.guided_regex is not None and not getattr(self, "_lowered", False):
                raise ValueError("guided_json and guided_regex exclude each other")
            from .ops.grammar import json_schema_to_regex
            self.guided_regex = json_schema_to_regex(self.guided_json)
            self.guided_json = None
        if self.guided_regex is not None:
            if self.json_object or self.guided_choice is not None:
                raise ValueError("a request can carry one output constraint")
            if not isinstance(self.guided_regex, str) or not self.guided_regex:
                raise ValueError("guided_regex must be a non-empty string")
            if len(self.guided_regex.encode("utf-8", "surrogatepass")) > MAX_CONSTRAINT_BYTES:
                raise ValueError(f"guided_regex is longer than {MAX_CONSTRAINT_BYTES} bytes")
            from .ops.grammar import compile_regex
            compile_regex(self.guided_regex)          # ValueError for what the automaton cannot express
        if self.guided_grammar is not None:
            if self.json_object or self.guided_choice is not None or self.guided_regex is not None:
                raise ValueError("a request can carry one output constraint")
            if not isinstance(self.guided_grammar, str) or not self.guided_grammar.strip():
                raise ValueError("guided_grammar must be a non-empty string")
            if len(self.guided_grammar.encode("utf-8", "surrogatepass")) > MAX_CONSTRAINT_BYTES:
                raise ValueError(f"guided_grammar is longer than {MAX_CONSTRAINT_BYTES} bytes")
            from .ops.grammar import compile_grammar
            compile_grammar(self.guided_grammar)      # ValueError for syntax errors, left recursion, ...
        if self.truncate_prompt_tokens is not None and self.truncate_prompt_tokens < 1:
            raise ValueError("truncate_prompt_tokens must be >= 1")
        if self.prompt_logprobs is not None and self.prompt_logprobs < 0:
            raise ValueError("prompt_logprobs must be >= 0")
        if self.min_tokens < 0:
            raise ValueError("min_tokens must be >= 0")

    @property
    def greedy(self) -> bool:
        return self.temperature == 0.0

    @property
    def has_penalties(self) -> bool:
        return (self.presence_penalty != 0.0 or self.frequency_penalty != 0.0
                or self.repetition_penalty != 1.0)

    @property
    def has_logit_adjust(self) -> bool:
        """True when logits must be edited before sampling (bias, allow / ban lists, min_tokens)."""
        return bool(self.logit_bias or self.allowed_token_ids is not None or self.bad_words_token_ids
                    or self.json_object or self.guided_choice or self.guided_regex or self.guided_grammar
                    or (self.min_tokens > 0 and self.min_tokens_ban_ids))

    def check_vocab(self, vocab_size: int) -> None:
        ids = list(self.logit_bias or ()) + list(self.allowed_token_ids or ())
        for w in self.bad_words_token_ids or ():
            ids.extend(w)
        if any(t >= vocab_size for t in ids):
            raise ValueError(f"a token id in logit_bias / allowed_token_ids / bad_words is outside the "
                             f"vocabulary of {vocab_size}")

    def clone(self) -> "SamplingParams":
        import copy
        return copy.deepcopy(self)
