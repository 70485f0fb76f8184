"""Beam search on top of the ordinary request path.

Every step asks the engine for ONE more token of every live beam with the top ``2 * width``
log-probabilities, expands, and keeps the best ``width`` continuations - the scheme of the serving stack
the reference launches (its front end drives beam search the same way, through plain one-token
requests, rather than inside the scheduler).  A beam's prefix is the previous step's prompt plus one
token, so with prefix caching on each step only computes that one token.

Two drivers share the bookkeeping: ``beam_search`` steps a synchronous ``LLMEngine``,
``beam_search_async`` runs over ``AsyncLLM.generate`` for the HTTP server.
"""
from __future__ import annotations

import asyncio
import uuid
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from ..sampling_params import SamplingParams


@dataclass
class BeamSearchParams:
    beam_width: int = 4
    max_tokens: int = 16
    length_penalty: float = 1.0
    ignore_eos: bool = False

    def __post_init__(self) -> None:
        if self.beam_width < 1:
            raise ValueError("beam_width must be >= 1")
        if self.max_tokens < 1:
            raise ValueError("max_tokens must be >= 1")


@dataclass
class Beam:
    tokens: List[int] = field(default_factory=list)      # generated so far (without the prompt)
    cum_logprob: float = 0.0
    finished: bool = False                               # ended with the EOS token (not included in tokens)

    def score(self, length_penalty: float) -> float:
        n = len(self.tokens) + (1 if self.finished else 0)
        return self.cum_logprob / (max(n, 1) ** length_penalty)


def _step_params(p: BeamSearchParams, max_logprobs: Optional[int]) -> SamplingParams:
    k = 2 * p.beam_width
    if max_logprobs is not None and max_logprobs >= 0:
        k = min(k, max(max_logprobs, 1))
    # beams are scored under the model's own distribution (no temperature); the sampled token itself is
    # irrelevant - only the step's log-probabilities are read - so greedy keeps the step deterministic
    return SamplingParams(temperature=0.0, max_tokens=1, logprobs=k, ignore_eos=True, detokenize=False)


def _expand(beams: List[Beam], step_logprobs: List[Dict[int, float]], p: BeamSearchParams,
            eos: Optional[int], done: List[Beam]) -> List[Beam]:
    cand: List[Beam] = []
    for b, lps in zip(beams, step_logprobs):
        for tok, lp in lps.items():
            if eos is not None and tok == eos and not p.ignore_eos:
                done.append(Beam(list(b.tokens), b.cum_logprob + lp, True))
            else:
                cand.append(Beam(b.tokens + [tok], b.cum_logprob + lp))
    cand.sort(key=lambda x: -x.cum_logprob)
    return cand[:p.beam_width]


def _final(beams: List[Beam], done: List[Beam], p: BeamSearchParams) -> List[Beam]:
    allb = done + beams
    allb.sort(key=lambda x: -x.score(p.length_penalty))
    return allb[:p.beam_width]


def _enough(beams: List[Beam], done: List[Beam], p: BeamSearchParams) -> bool:
    """No live beam can still beat the ``width`` best finished ones (log-probabilities only fall, and for
    ``length_penalty <= 1`` so does a live beam's best possible score per token...): conservative test on
    the raw cumulative log-probability with the most favourable length."""
    if len(done) < p.beam_width or not beams:
        return not beams
    worst_done = sorted((d.score(p.length_penalty) for d in done), reverse=True)[p.beam_width - 1]
    best_live = max(b.cum_logprob / (max(p.max_tokens, 1) ** p.length_penalty) if p.length_penalty > 0
                    else b.cum_logprob for b in beams)
    return best_live <= worst_done


def beam_search(engine, prompt_ids: Sequence[int], p: BeamSearchParams, lora_request=None) -> List[Beam]:
    """Best ``p.beam_width`` continuations of ``prompt_ids``, best first, on a synchronous engine."""
    eos = engine.eos_token_id
    sp = _step_params(p, None)
    beams, done = [Beam()], []
    room = engine.model_config.max_model_len - len(prompt_ids) - 1
    for _ in range(min(p.max_tokens, max(room, 0))):
        rids = [engine.add_request(None, list(prompt_ids) + b.tokens, sp, lora_request=lora_request) for b in beams]
        got: Dict[str, Dict[int, float]] = {}
        while len(got) < len(rids):
            for out in engine.step():
                if out.finished and out.request_id in rids:
                    got[out.request_id] = dict(out.outputs[0].logprobs[0])
        beams = _expand(beams, [got[r] for r in rids], p, eos, done)
        if _enough(beams, done, p):
            break
    return _final(beams, done, p)


async def beam_search_async(engine, prompt_ids: Sequence[int], p: BeamSearchParams, priority: int = 0,
                            lora_request=None, max_logprobs: Optional[int] = None) -> List[Beam]:
    """The same search over ``AsyncLLM.generate`` (one request per beam and step, run concurrently)."""
    eos = engine.engine.eos_token_id
    sp = _step_params(p, max_logprobs)
    beams, done = [Beam()], []
    room = engine.vllm_config.model_config.max_model_len - len(prompt_ids) - 1
    base = "beam-" + uuid.uuid4().hex

    async def one(ids: List[int], rid: str) -> Dict[int, float]:
        last = None
        async for out in engine.generate(ids, sp.clone(), rid, priority, lora_request):
            last = out
        return dict(last.outputs[0].logprobs[0])

    for step in range(min(p.max_tokens, max(room, 0))):
        res = await asyncio.gather(*[one(list(prompt_ids) + b.tokens, f"{base}-{step}-{i}")
                                     for i, b in enumerate(beams)])
        beams = _expand(beams, list(res), p, eos, done)
        if _enough(beams, done, p):
            break
    return _final(beams, done, p)
