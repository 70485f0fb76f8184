"""Synchronous engine front-end and the offline ``LLM`` class.

``LLMEngine`` = tokenizer + ``EngineCore`` + incremental detokenisation.
It is what the benchmark and the offline API drive directly; the HTTP server
wraps it in ``AsyncLLM``.
"""
from __future__ import annotations

import itertools
import time
from typing import Dict, Iterable, List, Optional, Sequence, Type, Union

from ..config import VllmConfig
from ..executor.abstract import Executor
from ..logger import init_logger
from ..outputs import CompletionOutput, RequestOutput
from ..sampling_params import SamplingParams
from .beam_search import BeamSearchParams
from .beam_search import beam_search as run_beam_search
from ..tokenizer import get_tokenizer
from .arg_utils import EngineArgs
from .core import EngineCore
from .detokenizer import IncrementalDetokenizer
from .request import Request, RequestStatus
from .scheduler import EngineCoreOutput

logger = init_logger(__name__)


class _ReqState:
    __slots__ = ("request", "detok", "prompt", "cumulative_logprob", "logprobs", "finish_reason",
                 "stop_reason", "prompt_logprobs", "pooled", "live")

    def __init__(self, request: Request, detok: Optional[IncrementalDetokenizer], prompt: Optional[str]):
        self.request = request
        self.detok = detok
        self.prompt = prompt
        self.cumulative_logprob: Optional[float] = None
        self.logprobs: Optional[List[Dict[int, float]]] = None
        self.finish_reason: Optional[str] = None
        self.stop_reason = None
        self.prompt_logprobs = None
        self.pooled = None
        # the completion object of a running request that is not detokenised: it shows the request's own
        # (live) token list, so one object serves every step until the request finishes
        self.live: Optional[List[CompletionOutput]] = None


class LLMEngine:
    def __init__(self, vllm_config: VllmConfig, executor_class: Optional[Type[Executor]] = None,
                 tokenizer=None, multiprocess: bool = False):
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.tokenizer = tokenizer or get_tokenizer(self.model_config)
        if multiprocess:        # scheduler + executor loop in a child process (engine/core_client.py)
            from .core_client import MPEngineCoreClient
            self.core = MPEngineCoreClient(vllm_config, executor_class)
        else:
            self.core = EngineCore(vllm_config, executor_class)
        self.remote_core = multiprocess
        self.states: Dict[str, _ReqState] = {}
        self._counter = itertools.count()
        eos = self.model_config.hf_config.get("eos_token_id")
        self.eos_token_id = eos[0] if isinstance(eos, list) else eos
        if getattr(self.tokenizer, "eos_token_id", None) is not None:
            self.eos_token_id = self.tokenizer.eos_token_id
        # further end-of-sequence ids (config.json / generation_config.json lists) act as stop tokens
        self.extra_eos_ids = [t for t in self.model_config.eos_token_ids() if t != self.eos_token_id]

    @classmethod
    def from_engine_args(cls, engine_args: EngineArgs) -> "LLMEngine":
        return cls(engine_args.create_engine_config())

    # -------------------------------------------------------------- requests
    def add_request(self, request_id: Optional[str], prompt: Union[str, Sequence[int], dict],
                    params: Optional[SamplingParams] = None, arrival_time: Optional[float] = None,
                    priority: int = 0, lora_request=None) -> str:
        params = params or SamplingParams()
        if request_id is None:
            request_id = f"req-{next(self._counter)}"
        text: Optional[str] = None
        mm_data = None
        if isinstance(prompt, dict):
            text = prompt.get("prompt")
            ids = prompt.get("prompt_token_ids")
            mm_data = prompt.get("multi_modal_data")
            if ids is None:
                ids = self.tokenizer.encode(text)
        elif isinstance(prompt, str):
            text, ids = prompt, self.tokenizer.encode(prompt)
        else:
            ids = list(prompt)
        ids = list(ids)
        max_len = self.model_config.max_model_len
        params = params.clone()          # the fields filled below are per request
        mm_inputs = self._process_images(ids, mm_data) if mm_data else None
        if mm_inputs is not None:
            ids, mm_inputs = mm_inputs
            if params.truncate_prompt_tokens is not None and params.truncate_prompt_tokens < len(ids):
                raise ValueError("truncate_prompt_tokens would cut through the image of a multimodal prompt")
        if params.truncate_prompt_tokens is not None:
            ids = ids[-params.truncate_prompt_tokens:]
        if params.max_tokens is None:
            params.max_tokens = max(1, max_len - len(ids))
        vocab = self.model_config.vocab_size
        if self.extra_eos_ids and not params.ignore_eos:
            params.stop_token_ids = params.stop_token_ids + [
                t for t in self.extra_eos_ids if t not in params.stop_token_ids]
        self._resolve_token_lists(params)
        params.check_vocab(vocab)
        if any(t < 0 or t >= vocab for t in ids):
            raise ValueError("prompt contains token ids outside the model vocabulary")
        # everything the scheduler would refuse is refused here, in the caller's process and
        # before anything is sent: a bad request must cost one error reply, never the engine
        if not ids:
            raise ValueError("empty prompt")
        if len(ids) >= max_len:
            raise ValueError(f"prompt ({len(ids)} tokens) does not fit max_model_len ({max_len})")
        if request_id in self.states:
            raise ValueError(f"duplicate request id {request_id}")
        if lora_request is not None and self.vllm_config.lora_config is None:
            raise ValueError("the engine was started without --enable-lora")
        req = Request(request_id, ids, params, self.eos_token_id, arrival_time, priority, text, lora_request,
                      mm_inputs)
        detok = IncrementalDetokenizer(self.tokenizer, ids, params) if params.detokenize else None
        self.core.add_request(req)
        self.states[request_id] = _ReqState(req, detok, text)
        return request_id

    def _process_images(self, ids: List[int], mm_data: dict):
        """Pixels for the vision tower and the prompt with every image placeholder widened to the number of
        embeddings one image yields (``multimodal.py``)."""
        from .. import multimodal as mm
        cfg = self.model_config.hf_config
        if cfg.get("mm_model_type") is None:
            raise ValueError(f"{self.model_config.model} does not take images")
        images = mm_data.get("image") if isinstance(mm_data, dict) else None
        if images is None:
            raise ValueError("multi_modal_data: only {\"image\": ...} is supported")
        if not isinstance(images, (list, tuple)):
            images = [images]
        anyres = cfg.get("mm_model_type") == "llava_next"
        if getattr(self, "_image_processor", None) is None:
            from ..config import resolve_model_path
            self._image_processor = mm.ClipImageProcessor.from_model(resolve_model_path(self.model_config.model),
                                                                     cfg["vision_config"])
            if anyres:
                self._image_processor = mm.AnyResImageProcessor(self._image_processor, cfg["image_grid_pinpoints"])
        loaded = [mm.load_image(im) for im in images]
        pixels = [self._image_processor(im) for im in loaded]
        token = int(cfg.get("image_token_index", cfg.get("image_token_id", 32000)))
        if anyres:          # as many embeddings as the image's own aspect leaves of its tile grid
            sizes = [(im.size[1], im.size[0]) for im in loaded]
            return mm.expand_image_placeholders(ids, pixels, token, [mm.anyres_tokens(cfg, s) for s in sizes], sizes)
        return mm.expand_image_placeholders(ids, pixels, token, mm.tokens_per_image(cfg))

    def _resolve_token_lists(self, params: SamplingParams) -> None:
        """Fill the token-id forms of ``bad_words`` and of the tokens ``min_tokens`` holds back."""
        if params.bad_words:
            words = list(params.bad_words_token_ids or [])
            for w in params.bad_words:
                # a word reads differently at the start of a text and after a space: ban both spellings
                for variant in (w.lstrip(), " " + w.lstrip()):
                    try:
                        t = list(self.tokenizer.encode(variant, add_special_tokens=False))
                    except TypeError:
                        t = list(self.tokenizer.encode(variant))
                    if t and t not in words:
                        words.append(t)
            params.bad_words_token_ids = words
        params.eos_token_id = self.eos_token_id
        if params.min_tokens > 0:
            ban = set(params.stop_token_ids)
            if self.eos_token_id is not None:
                ban.add(int(self.eos_token_id))
            params.min_tokens_ban_ids = sorted(ban)

    def abort_request(self, request_ids: Union[str, Iterable[str]]) -> None:
        if isinstance(request_ids, str):
            request_ids = [request_ids]
        request_ids = list(request_ids)
        self.core.abort_requests(request_ids)
        for rid in request_ids:
            self.states.pop(rid, None)

    def has_unfinished_requests(self) -> bool:
        return self.core.has_unfinished_requests()

    def get_num_unfinished_requests(self) -> int:
        if self.remote_core:
            return self.core.get_num_unfinished_requests()
        return self.core.scheduler.get_num_unfinished_requests()

    # ------------------------------------------------------------------ step
    def step(self) -> List[RequestOutput]:
        return self._process(self.core.step())

    def _process(self, core_outputs: List[EngineCoreOutput]) -> List[RequestOutput]:
        outs: List[RequestOutput] = []
        stop_hits: List[str] = []
        states, remote = self.states, self.remote_core
        for co in core_outputs:
            st = states.get(co.request_id)
            if st is None:
                continue
            req = st.request
            if (st.detok is None and co.finish_reason is None and not remote and co.logprobs is None
                    and co.prompt_logprobs is None and co.pooled is None and st.logprobs is None):
                # a plain decode step of a request nobody detokenises: no text, nothing accumulated
                live = st.live
                if live is None:
                    live = st.live = [CompletionOutput(index=0, text="", token_ids=req.output_token_ids,
                                                       cumulative_logprob=st.cumulative_logprob, logprobs=None,
                                                       finish_reason=None, stop_reason=co.stop_reason)]
                outs.append(RequestOutput(co.request_id, st.prompt, req.prompt_token_ids, live, False, req.metrics,
                                          co.new_token_ids, "", st.prompt_logprobs, st.pooled))
                continue
            finish_reason, stop_reason = co.finish_reason, co.stop_reason
            if self.remote_core:        # the scheduler's Request lives in the core process: mirror it
                for t in co.new_token_ids:
                    req.append_output_token(t)
                m = req.metrics
                if co.new_token_ids and m.first_token_time is None:
                    m.first_token_time = time.time()
                m.num_generation_tokens = len(req.output_token_ids)
                m.num_cached_tokens = co.num_cached_tokens
            if st.detok is not None:
                hit = st.detok.append(co.new_token_ids)
                if hit is not None and finish_reason is None:
                    finish_reason, stop_reason = "stop", hit
                    stop_hits.append(co.request_id)
            if co.logprobs:
                st.logprobs = (st.logprobs or []) + co.logprobs
                for tok, lp in zip(co.new_token_ids, co.logprobs):
                    if lp is not None and tok in lp:
                        st.cumulative_logprob = (st.cumulative_logprob or 0.0) + lp[tok]
            if co.prompt_logprobs is not None:
                st.prompt_logprobs = co.prompt_logprobs
            if co.pooled is not None:
                st.pooled = co.pooled
            finished = finish_reason is not None
            new_text = st.detok.take_delta(finished) if st.detok is not None else ""
            # the live list while the request runs (no per-step copy); a snapshot once it is final
            comp = CompletionOutput(index=0, text=st.detok.text if st.detok is not None else "",
                                    token_ids=list(req.output_token_ids) if finished else req.output_token_ids,
                                    cumulative_logprob=st.cumulative_logprob, logprobs=st.logprobs,
                                    finish_reason=finish_reason, stop_reason=stop_reason)
            outs.append(RequestOutput(request_id=co.request_id, prompt=st.prompt,
                                      prompt_token_ids=req.prompt_token_ids, outputs=[comp],
                                      finished=finished, metrics=req.metrics,
                                      new_token_ids=co.new_token_ids, new_text=new_text,
                                      prompt_logprobs=st.prompt_logprobs, pooled=st.pooled))
            if finished:
                if req.metrics.finished_time is None:
                    req.metrics.finished_time = time.time()
                self.states.pop(co.request_id, None)
        if stop_hits:
            self.core.finish_requests(stop_hits, RequestStatus.FINISHED_STOPPED)
        return outs

    def shutdown(self) -> None:
        self.core.shutdown()


class LLM:
    """Offline batch generation: ``LLM(model=...).generate(prompts, sampling_params)``."""

    def __init__(self, model: str = "llama-tiny", **kwargs):
        self.engine = LLMEngine.from_engine_args(EngineArgs(model=model, **kwargs))

    def get_tokenizer(self):
        return self.engine.tokenizer

    def generate(self, prompts: Union[str, Sequence[Union[str, Sequence[int], dict]]],
                 sampling_params: Union[None, SamplingParams, Sequence[SamplingParams]] = None,
                 lora_request=None) -> List[RequestOutput]:
        if isinstance(prompts, (str, dict)) or (prompts and isinstance(prompts[0], int)):
            prompts = [prompts]  # type: ignore[list-item]
        n = len(prompts)
        if sampling_params is None or isinstance(sampling_params, SamplingParams):
            sps = [sampling_params or SamplingParams()] * n
        else:
            sps = list(sampling_params)
        loras = lora_request if isinstance(lora_request, (list, tuple)) else [lora_request] * n
        # ``n`` (or ``best_of``) samples of a prompt are that many requests sharing the prompt (and,
        # through the prefix cache, its KV blocks); a fixed seed is advanced per sample
        groups: List[List[str]] = []
        for p, sp, lr in zip(prompts, sps, loras):
            fan = sp.best_of or sp.n
            if fan == 1:
                groups.append([self.engine.add_request(None, p, sp, lora_request=lr)])
                continue
            members = []
            for j in range(fan):
                one = sp.clone()
                one.n, one.best_of = 1, None
                if one.seed is not None:
                    one.seed = (one.seed + j) % (2 ** 63)
                if sp.best_of is not None and sp.best_of > sp.n and one.logprobs is None:
                    one.logprobs = 0            # the ranking needs each sample's cumulative logprob
                members.append(self.engine.add_request(None, p, one, lora_request=lr))
            groups.append(members)
        final: Dict[str, RequestOutput] = {}
        while self.engine.has_unfinished_requests():
            for out in self.engine.step():
                if out.finished:
                    final[out.request_id] = out
        return [merge_samples([final[i] for i in g], sp) for g, sp in zip(groups, sps)]

    def beam_search(self, prompts, params: Optional[BeamSearchParams] = None, lora_request=None,
                    ) -> List[RequestOutput]:
        """Best ``beam_width`` continuations of every prompt (``outputs`` sorted best first, each with its
        ``cumulative_logprob``)."""
        params = params or BeamSearchParams()
        if isinstance(prompts, (str, dict)) or (prompts and isinstance(prompts[0], int)):
            prompts = [prompts]  # type: ignore[list-item]
        tok = self.engine.tokenizer
        results = []
        for k, p in enumerate(prompts):
            if isinstance(p, dict):
                ids = p.get("prompt_token_ids") or tok.encode(p.get("prompt"))
                text = p.get("prompt")
            elif isinstance(p, str):
                ids, text = tok.encode(p), p
            else:
                ids, text = list(p), None
            beams = run_beam_search(self.engine, ids, params, lora_request)
            outs = [CompletionOutput(index=i, text=tok.decode(b.tokens), token_ids=list(b.tokens),
                                     cumulative_logprob=b.cum_logprob, logprobs=None,
                                     finish_reason="stop" if b.finished else "length", stop_reason=None)
                    for i, b in enumerate(beams)]
            results.append(RequestOutput(request_id=f"beam-{k}", prompt=text, prompt_token_ids=list(ids),
                                         outputs=outs, finished=True, metrics=None, new_token_ids=[],
                                         new_text="", prompt_logprobs=None, pooled=None))
        return results

    def chat(self, messages: List[dict], sampling_params: Optional[SamplingParams] = None,
             tools=None) -> List[RequestOutput]:
        prompt = self.engine.tokenizer.apply_chat_template(messages, tokenize=False,
                                                           add_generation_prompt=True, tools=tools)
        return self.generate([prompt], sampling_params)

    def encode(self, prompts, pooling: str = "last", normalize: bool = False, lora_request=None) -> List[List[float]]:
        """Pooled final hidden state of each prompt ("last" token or "mean" over the prompt)."""
        outs = self.generate(prompts, SamplingParams(pooling=pooling, temperature=0.0, detokenize=False),
                             lora_request)
        vecs = [o.pooled for o in outs]
        if normalize:
            vecs = [l2_normalize(v) for v in vecs]
        return vecs

    def embed(self, prompts, lora_request=None) -> List[List[float]]:
        """Unit-length last-token embeddings (the default pooling of a causal LM used as an embedder)."""
        return self.encode(prompts, "last", True, lora_request)

    def score(self, text_1, text_2) -> List[float]:
        """Cosine similarity of the embeddings of ``text_1[i]`` and ``text_2[i]`` (one ``text_1`` is
        paired with every ``text_2``)."""
        a = [text_1] if isinstance(text_1, str) else list(text_1)
        b = [text_2] if isinstance(text_2, str) else list(text_2)
        if len(a) == 1:
            a = a * len(b)
        if len(a) != len(b):
            raise ValueError("text_1 must hold one entry, or as many as text_2")
        vecs = self.embed(a + b)
        return [float(sum(x * y for x, y in zip(vecs[i], vecs[len(a) + i]))) for i in range(len(a))]

    def shutdown(self) -> None:
        self.engine.shutdown()


def merge_samples(outs: List[RequestOutput], sp: SamplingParams) -> RequestOutput:
    """The finished requests of one prompt's samples as one ``RequestOutput``: the ``n`` best by
    cumulative log-probability when ``best_of`` asked for more samples than it returns."""
    if len(outs) == 1:
        return outs[0]
    comps = [o.outputs[0] for o in outs]
    if sp.best_of is not None and sp.best_of > sp.n:
        comps.sort(key=lambda c: -(c.cumulative_logprob if c.cumulative_logprob is not None else float("-inf")))
        comps = comps[:sp.n]
    for i, c in enumerate(comps):
        c.index = i
    first = outs[0]
    first.outputs = comps
    return first


def l2_normalize(v: List[float]) -> List[float]:
    n = sum(x * x for x in v) ** 0.5
    return [x / n for x in v] if n > 0 else list(v)
