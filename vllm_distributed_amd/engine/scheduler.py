"""Continuous-batching scheduler with chunked prefill and preemption.

Every step hands out a token budget (``max_num_batched_tokens``): running
requests first (decodes need one token; a partially prefilled prompt takes as
much of its remainder as fits), then waiting requests in FCFS or priority
order.  There is no separate prefill/decode phase - a step is just "request ->
number of tokens to run".  When KV blocks run out the most recently admitted
running request is preempted (blocks freed, recomputed later).

A request's ``num_computed_tokens`` is advanced when it is *scheduled*, so with
several batches in flight (pipeline parallelism / async scheduling,
``max_concurrent_batches`` in reference ``src/launch.py:298-302``) a request
whose next token is still being computed is simply not schedulable until
``update_from_output`` appends that token.
"""
from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass
from typing import Deque, Dict, Iterable, List, Optional, Set

import numpy as np

from ..config import CacheConfig, SchedulerConfig
from ..logger import init_logger
from .kv_cache_manager import KVCacheManager
from .request import Request, RequestStatus
from .types import (CachedRequestData, ModelRunnerOutput, NewRequestData, SchedulerOutput)

logger = init_logger(__name__)


_NO_BLOCKS: List[int] = []


@dataclass
class EngineCoreOutput:
    request_id: str
    new_token_ids: List[int]
    finish_reason: Optional[str] = None
    stop_reason: Optional[object] = None
    logprobs: Optional[List[Dict[int, float]]] = None
    num_cached_tokens: int = 0
    # with the first output token of a request that asked: one entry per prompt token (None for the first)
    prompt_logprobs: Optional[List[Optional[Dict[int, float]]]] = None
    pooled: Optional[List[float]] = None


@dataclass
class SchedulerStats:
    num_running: int = 0
    num_waiting: int = 0
    kv_cache_usage: float = 0.0
    prefix_cache_queries: int = 0
    prefix_cache_hits: int = 0
    num_preemptions: int = 0


class Scheduler:
    SPEC_WINDOW = 8192            # most recent tokens searched for a draft

    def __init__(self, scheduler_config: SchedulerConfig, cache_config: CacheConfig,
                 num_blocks: int, include_new_token_ids: bool = True,
                 kv_connector=None, max_loras: int = 0, speculative_config=None,
                 sliding_window: Optional[int] = None, pipeline_depth: int = 1):
        self.config = scheduler_config
        # pipeline parallelism keeps `pipeline_depth` batches in flight, and a stage only has work
        # while a batch is at it: the running requests are dealt out over that many micro-batches
        # instead of all going into the first one (which would leave every other stage idle)
        self.pipeline_depth = max(1, int(pipeline_depth))
        self.cache_config = cache_config
        self.max_num_seqs = scheduler_config.max_num_seqs
        self.max_num_batched_tokens = scheduler_config.max_num_batched_tokens
        self.max_model_len = scheduler_config.max_model_len
        # async scheduling: step n+1 is scheduled before step n's tokens are known;
        # a request that will sample carries a placeholder for the unknown token
        self.async_scheduling = bool(scheduler_config.async_scheduling)
        self.kv = KVCacheManager(num_blocks, cache_config.block_size, self.max_model_len,
                                 cache_config.enable_prefix_caching, sliding_window=sliding_window)
        self.requests: Dict[str, Request] = {}
        self.waiting: Deque[Request] = deque()
        self.running: List[Request] = []
        # ids the workers still hold state for and must drop on the next step
        self._finished_since_last: Set[str] = set()
        self._preempted_since_last: Set[str] = set()
        # requests the workers already have cached state for
        self._known_to_workers: Set[str] = set()
        self._pending_new_tokens: Dict[str, List[int]] = {}
        self.include_new_token_ids = include_new_token_ids
        self.kv_connector = kv_connector
        # requests whose prompt KV the workers are copying in from another engine (not schedulable until every
        # rank has reported the copy), those whose copy has landed and that wait at the head of the queue, and
        # those that were aborted mid-copy (their blocks are only freed once nobody writes to them any more)
        self.kv_loading: Dict[str, Request] = {}
        self._kv_loaded: Set[str] = set()
        self._kv_load_orphans: Dict[str, Request] = {}
        self.spec = speculative_config
        self.spec_stats = {"drafted": 0, "accepted": 0}
        self._ngram = None
        # draft-model speculation: the workers' proposal loop writes draft KV this many positions past
        # the tokens a step schedules, so every allocation reaches that far
        self.lookahead = (self.spec.num_speculative_tokens
                          if self.spec is not None and self.spec.worker_side else 0)
        if self.spec is not None and self.spec.method == "ngram":
            import ctypes
            from ..ops.grammar import load as load_core
            lib = load_core()
            lib.vdamd_ngram_propose.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_void_p]
            lib.vdamd_ngram_propose.restype = ctypes.c_int
            self._ngram = lib.vdamd_ngram_propose
        # distinct adapters one batch may use (= the workers' resident slots); 0 = LoRA off
        self.max_loras = max_loras
        self.step_id = 0
        self.num_preemptions = 0

    # ---------------------------------------------------------------- intake
    def add_request(self, request: Request) -> None:
        if request.request_id in self.requests:
            raise ValueError(f"duplicate request id {request.request_id}")
        if request.num_prompt_tokens == 0:
            raise ValueError("empty prompt")
        if request.num_prompt_tokens >= self.max_model_len:
            raise ValueError(f"prompt ({request.num_prompt_tokens} tokens) does not fit "
                             f"max_model_len ({self.max_model_len})")
        if request.lora_request is not None and self.max_loras <= 0:
            raise ValueError("the engine was started without --enable-lora")
        self.requests[request.request_id] = request
        self._enqueue_waiting(request)

    def _enqueue_waiting(self, request: Request, front: bool = False) -> None:
        if self.config.policy == "priority":
            items = list(self.waiting)
            items.append(request)
            items.sort(key=lambda r: (r.priority, r.arrival_time))
            self.waiting = deque(items)
        elif front:
            self.waiting.appendleft(request)
        else:
            self.waiting.append(request)

    # ------------------------------------------------------- speculative decoding
    def _propose_drafts(self, req: Request, budget: int) -> List[int]:
        """Drafts for a request in plain decode (looked up in its own text, or what the workers' draft
        model proposed after its newest token), or [].  Requests whose next-token distribution
        depends on more than the text (penalties, logit edits, grammars, a seed's draw order) or
        that report logprobs are decoded one token at a time."""
        sc = self.spec
        sp = req.sampling_params
        proposed, req.spec_token_ids = req.spec_token_ids, []
        if (req.num_computed_tokens != req.num_tokens - 1 or req.num_output_placeholders
                or sp.has_penalties or sp.has_logit_adjust or sp.logprobs is not None or sp.seed is not None
                or sp.pooling is not None or req.lora_request is not None):
            return []
        room = min(sc.num_speculative_tokens, budget - 1, self.max_model_len - req.num_tokens - 1)
        if req.max_tokens is not None:
            room = min(room, req.max_tokens - len(req.output_token_ids) - 1)
        if room <= 0:
            return []
        if sc.worker_side:
            return proposed[:room]
        ids = req.all_token_ids
        arr = np.asarray(ids[-self.SPEC_WINDOW:], dtype=np.int32)
        out = np.empty(room, dtype=np.int32)
        n = self._ngram(arr.ctypes.data, len(arr), sc.prompt_lookup_max, sc.prompt_lookup_min, room, out.ctypes.data)
        return out[:n].tolist()

    def abort_requests(self, request_ids: Iterable[str]) -> List[str]:
        aborted = []
        for rid in list(request_ids):
            req = self.requests.get(rid)
            if req is None or req.finished:
                continue
            if self._orphan_if_loading(req, RequestStatus.FINISHED_ABORTED):
                aborted.append(rid)
                continue
            if req.status == RequestStatus.RUNNING:
                self.running.remove(req)
            else:
                try:
                    self.waiting.remove(req)
                except ValueError:
                    pass
            req.status = RequestStatus.FINISHED_ABORTED
            self._free_request(req)
            aborted.append(rid)
        return aborted

    def has_unfinished_requests(self) -> bool:
        if self.waiting or self.running:
            return True
        # KV transfers in flight keep the engine stepping: their completion arrives with a step's output
        return self.kv_connector is not None and bool(self.kv_loading or self._kv_load_orphans
                                                      or self.kv_connector.sending)

    def has_requests(self) -> bool:
        return self.has_unfinished_requests() or bool(self._finished_since_last)

    def get_num_unfinished_requests(self) -> int:
        return len(self.waiting) + len(self.running)

    # -------------------------------------------------------------- schedule
    def schedule(self) -> SchedulerOutput:
        budget = self.max_num_batched_tokens
        num_scheduled: Dict[str, int] = {}
        new_reqs: List[NewRequestData] = []
        cached = CachedRequestData()
        preempted: List[Request] = []
        spec_tokens: Dict[str, List[int]] = {}
        now = time.time()
        req_blocks, block_size = self.kv.req_blocks, self.kv.block_size

        # 1. running requests
        seq_cap = -(-len(self.running) // self.pipeline_depth) if self.pipeline_depth > 1 else len(self.running)
        # the plain decode step of a request (one token, room left in its last block, nothing that
        # needs bookkeeping) is most of a large batch: it is handled in a dozen statements up front
        quick = (self.spec is None and self.kv_connector is None and self.lookahead == 0
                 and self.config.long_prefill_token_threshold <= 0)
        pending_new = self._pending_new_tokens
        running_state = RequestStatus.RUNNING
        max_model_len, async_mode = self.max_model_len, self.async_scheduling
        for req in list(self.running):
            if budget <= 0 or len(num_scheduled) >= seq_cap:
                break
            if req.status != running_state:
                continue             # preempted earlier in this loop
            if quick:
                ph, comp, have = req.num_output_placeholders, req.num_computed_tokens, len(req.all_token_ids)
                if have + ph - comp == 1 and comp + 1 < max_model_len:
                    mt = req.max_tokens
                    if ph and ((mt is not None and len(req.output_token_ids) + ph >= mt) or have + ph >= max_model_len):
                        continue     # would overshoot max_tokens / max_model_len: wait for the token
                    rid = req.request_id
                    held = req_blocks.get(rid)
                    if held is not None and comp + 1 <= len(held) * block_size:
                        if pending_new and rid in pending_new:
                            self._add_cached(cached, req, _NO_BLOCKS)
                        num_scheduled[rid] = 1
                        req.num_computed_tokens = comp + 1
                        if async_mode and comp + 1 >= have + ph:
                            req.num_output_placeholders = ph + 1
                        budget -= 1
                        continue
            n = req.num_tokens + req.num_output_placeholders - req.num_computed_tokens
            if n <= 0:               # its next token is still in flight
                continue
            if req.num_output_placeholders and not self._may_run_ahead(req):
                continue             # would overshoot max_tokens / max_model_len: wait for the token
            n = min(n, budget)
            if self.config.long_prefill_token_threshold > 0:
                n = min(n, self.config.long_prefill_token_threshold)
            n = min(n, self.max_model_len - req.num_computed_tokens)
            if n <= 0:
                continue
            drafts = self._propose_drafts(req, budget) if self.spec is not None and n == 1 else None
            if drafts:
                n = 1 + len(drafts)
            held = req_blocks.get(req.request_id)
            reach = min(n + self.lookahead, self.max_model_len - req.num_computed_tokens)
            if held is not None and req.num_computed_tokens + reach <= len(held) * block_size:
                new_blocks = _NO_BLOCKS          # the common decode step: room left in the last block
            else:
                new_blocks = self.kv.allocate_slots(req, reach)
            while new_blocks is None:
                victim = self._pick_victim(exclude_scheduled=num_scheduled)
                if victim is None:
                    break
                self._preempt(victim)
                preempted.append(victim)
                if victim is req:
                    break
                new_blocks = self.kv.allocate_slots(req, reach)
            if new_blocks is None or req.status != RequestStatus.RUNNING:
                continue
            self._add_cached(cached, req, new_blocks)
            num_scheduled[req.request_id] = n
            if drafts:
                spec_tokens[req.request_id] = drafts
                self.spec_stats["drafted"] += len(drafts)
            before = req.num_computed_tokens
            req.num_computed_tokens += n
            if self.kv_connector is not None and before < req.num_prompt_tokens <= req.num_computed_tokens:
                self.kv_connector.note_prefill_scheduled(req, self.kv.get_block_ids(req.request_id))
            self._note_sampling(req)
            budget -= n

        # 2. waiting requests (not while we are under memory pressure)
        if not preempted:
            while self.waiting and budget > 0 and len(self.running) + len(self.kv_loading) < self.max_num_seqs:
                req = self.waiting[0]
                if req.lora_request is not None:
                    active = {r.lora_request.lora_int_id for r in self.running if r.lora_request is not None}
                    if req.lora_request.lora_int_id not in active and len(active) >= self.max_loras:
                        break            # every adapter slot is busy with running requests
                computed_blocks, num_cached = ([], 0)
                loaded = req.request_id in self._kv_loaded          # back from a load: its blocks are in place
                if loaded:
                    num_cached = req.num_computed_tokens
                elif req.num_computed_tokens == 0:
                    computed_blocks, num_cached = self.kv.get_computed_blocks(req)
                # prompt KV another engine left in the shared storage: the workers copy it into fresh
                # blocks on a side thread while steps go on; the request is parked (``kv_loading``) and
                # comes back to the head of the queue once every rank has reported the copy done, with
                # the loaded tokens counted as computed like a cache hit
                if self.kv_connector is not None and not loaded and req.num_computed_tokens == 0:
                    ext = self.kv_connector.get_num_new_matched_tokens(req, num_cached)
                    if ext:
                        if self.kv.allocate_slots(req, ext, computed_blocks) is None:
                            break
                        self.kv_connector.note_load(req, num_cached, ext, self.kv.get_block_ids(req.request_id))
                        self.waiting.popleft()
                        req.num_computed_tokens = num_cached + ext
                        if req.num_cached_tokens < 0:
                            req.num_cached_tokens = num_cached + ext
                            req.metrics.num_cached_tokens = num_cached + ext
                        self.kv_loading[req.request_id] = req
                        continue
                ext = 0
                n = req.num_tokens - num_cached - ext
                if not self.config.enable_chunked_prefill and n > budget:
                    break
                n = min(n, budget)
                if self.config.long_prefill_token_threshold > 0:
                    n = min(n, self.config.long_prefill_token_threshold)
                if n <= 0:
                    break
                blocks = self.kv.allocate_slots(
                    req, min(n + ext + self.lookahead, self.max_model_len - num_cached), computed_blocks)
                if blocks is None:
                    break
                if loaded:
                    self._kv_loaded.discard(req.request_id)
                    blocks = self.kv.get_block_ids(req.request_id)      # the workers meet this request for the first time
                self.waiting.popleft()
                resumed = req.status == RequestStatus.PREEMPTED
                req.status = RequestStatus.RUNNING
                self.running.append(req)
                if req.metrics.first_scheduled_time is None:
                    req.metrics.first_scheduled_time = now
                if req.num_cached_tokens < 0:
                    req.num_cached_tokens = num_cached
                    req.metrics.num_cached_tokens = num_cached
                req.num_computed_tokens = num_cached
                # workers dropped all state for finished/preempted requests, so
                # (re)introduce it as new, carrying any tokens already generated
                new_reqs.append(NewRequestData(
                    req_id=req.request_id, prompt_token_ids=req.prompt_token_ids,
                    sampling_params=req.sampling_params, block_ids=blocks,
                    num_computed_tokens=num_cached,
                    output_token_ids=list(req.output_token_ids) if resumed else [],
                    lora_request=req.lora_request, mm_inputs=req.mm_inputs))
                self._known_to_workers.add(req.request_id)
                self._pending_new_tokens.pop(req.request_id, None)
                num_scheduled[req.request_id] = n
                req.num_computed_tokens += n
                if self.kv_connector is not None and req.num_computed_tokens >= req.num_prompt_tokens:
                    self.kv_connector.note_prefill_scheduled(req, self.kv.get_block_ids(req.request_id))
                self._note_sampling(req)
                budget -= n

        total = sum(num_scheduled.values())
        out = SchedulerOutput(
            step_id=self.step_id, scheduled_new_reqs=new_reqs, scheduled_cached_reqs=cached,
            num_scheduled_tokens=num_scheduled, total_num_scheduled_tokens=total,
            finished_req_ids=self._finished_since_last,
            preempted_req_ids=self._preempted_since_last,
            scheduled_spec_decode_tokens=spec_tokens or None)
        self._finished_since_last = set()
        self._preempted_since_last = set()
        if self.kv_connector is not None:
            out.kv_connector_metadata = self.kv_connector.build_connector_meta(
                out, pending=bool(self.kv_loading or self._kv_load_orphans or self.kv_connector.sending))
        self.step_id += 1
        return out

    def _note_sampling(self, req: Request) -> None:
        """Async mode: if this step runs the request's last known position it will
        sample a token; account for it so the next step can be scheduled now."""
        if self.async_scheduling and \
                req.num_computed_tokens >= req.num_tokens + req.num_output_placeholders:
            req.num_output_placeholders += 1

    def _may_run_ahead(self, req: Request) -> bool:
        n_out = req.num_output_tokens + req.num_output_placeholders
        if req.max_tokens is not None and n_out >= req.max_tokens:
            return False
        return req.num_tokens + req.num_output_placeholders < self.max_model_len

    def _add_cached(self, cached: CachedRequestData, req: Request, new_blocks: List[int]) -> None:
        """Record a delta for an already-known request.  The list is sparse:
        workers advance computed-token counts themselves, and TP ranks append
        the tokens they sampled, so a request with no new block and no token
        to forward needs no entry."""
        if not new_blocks and req.request_id not in self._pending_new_tokens:
            return
        cached.req_ids.append(req.request_id)
        cached.new_block_ids.append(new_blocks)
        cached.num_computed_tokens.append(req.num_computed_tokens)
        toks = self._pending_new_tokens.pop(req.request_id, None) if self.include_new_token_ids else None
        cached.new_token_ids.append(toks or [])

    def _pick_victim(self, exclude_scheduled: Dict[str, int]) -> Optional[Request]:
        """Lowest-priority running request that is not already part of this
        step's batch (its blocks are about to be written)."""
        candidates = [r for r in self.running if r.request_id not in exclude_scheduled]
        if not candidates:
            return None
        if self.config.policy == "priority":
            return max(candidates, key=lambda r: (r.priority, r.arrival_time))
        return candidates[-1]

    def _preempt(self, req: Request) -> None:
        self.running.remove(req)
        self.kv.free(req)
        req.status = RequestStatus.PREEMPTED
        req.num_computed_tokens = 0
        req.num_output_placeholders = 0
        req.num_preemptions += 1
        self.num_preemptions += 1
        if req.request_id in self._known_to_workers:
            self._known_to_workers.discard(req.request_id)
            self._preempted_since_last.add(req.request_id)
        self._pending_new_tokens.pop(req.request_id, None)
        self._enqueue_waiting(req, front=True)

    # ------------------------------------------------------------ post-step
    def update_from_output(self, scheduler_output: SchedulerOutput,
                           runner_output: ModelRunnerOutput) -> List[EngineCoreOutput]:
        """Fold one step's sampled tokens into the requests.  This loop runs once
        per running request per step, so the common case (one token, no stop)
        is kept free of helper calls."""
        outputs: List[EngineCoreOutput] = []
        if runner_output is None:
            return outputs
        if self.kv_connector is not None:
            for held in self.kv_connector.update_from_output(runner_output):
                self.kv.free(held)                 # its save has been confirmed by every rank
            for rid in (runner_output.finished_recving or ()):
                req = self.kv_loading.pop(rid, None)
                if req is not None:                # every rank holds the loaded blocks: schedulable, ahead of newcomers
                    self._kv_loaded.add(rid)
                    self._enqueue_waiting(req, front=True)
                    continue
                req = self._kv_load_orphans.pop(rid, None)
                if req is not None:
                    self.kv.free(req)
        lps = dict(zip(runner_output.req_ids, runner_output.logprobs)) if runner_output.logprobs else None
        now = time.time()
        requests = self.requests
        scheduled = scheduler_output.num_scheduled_tokens
        spec = scheduler_output.scheduled_spec_decode_tokens
        if runner_output.prompt_logprobs:
            for rid, (first, entries) in runner_output.prompt_logprobs.items():
                req = requests.get(rid)
                if req is not None and req.prompt_logprobs is not None:
                    for j, d in enumerate(entries):
                        req.prompt_logprobs.setdefault(first + j, d)
        caching = self.kv.enable_caching
        proposed = runner_output.draft_token_ids
        windowed = self.kv.window is not None
        fwd_tokens = self.include_new_token_ids
        running_state = RequestStatus.RUNNING
        max_len = self.max_model_len
        any_finished = False
        # one token for a request that can only end by length, with nothing else to record: most of a
        # large decode batch, and the reason this loop has a short path
        quick = (not spec and not caching and not windowed and lps is None and proposed is None
                 and not runner_output.pooled)
        pending_new = self._pending_new_tokens
        for rid, new_tokens in zip(runner_output.req_ids, runner_output.sampled_token_ids):
            req = requests.get(rid)
            if req is None or req.status != running_state or rid not in scheduled:
                continue             # aborted / preempted / finished while in flight
            if quick and req.plain_stop and len(new_tokens) == 1:
                out_ids, all_ids, mt = req.output_token_ids, req.all_token_ids, req.max_tokens
                n_out = len(out_ids) + 1
                if (n_out > 1 and (mt is None or n_out < mt) and len(all_ids) + 1 < max_len
                        and req.prompt_logprobs is None):
                    tok = new_tokens[0]
                    out_ids.append(tok)
                    all_ids.append(tok)
                    if req.num_output_placeholders:
                        req.num_output_placeholders -= 1
                    req.metrics.num_generation_tokens = n_out
                    if fwd_tokens:
                        got = pending_new.get(rid)
                        if got is None:
                            pending_new[rid] = list(new_tokens)
                        else:
                            got.extend(new_tokens)
                    nc = req.num_cached_tokens
                    outputs.append(EngineCoreOutput(rid, new_tokens, None, req.stop_reason, None, nc if nc > 0 else 0))
                    continue
            if spec and new_tokens and rid in spec:
                # 1 + len(drafts) positions ran; the accepted drafts and one more token came back:
                # the positions of the rejected drafts count as not computed (their KV is overwritten)
                req.num_computed_tokens -= len(spec[rid]) + 1 - len(new_tokens)
                self.spec_stats["accepted"] += len(new_tokens) - 1
            if caching:              # blocks that are now full and fully known can be shared
                self.kv.cache_blocks(req, req.num_computed_tokens)
            if windowed:             # ... and blocks every layer's window has left go back to the pool
                self.kv.release_out_of_window(req)
            if not new_tokens:
                continue             # partial prefill: nothing sampled
            if req.num_output_placeholders:
                req.num_output_placeholders = max(0, req.num_output_placeholders - len(new_tokens))
            sp = req.sampling_params
            out_ids, all_ids = req.output_token_ids, req.all_token_ids
            accepted = new_tokens
            for i, tok in enumerate(new_tokens):
                out_ids.append(tok)
                all_ids.append(tok)
                n_out = len(out_ids)
                stop = False
                if n_out >= sp.min_tokens:
                    if (not sp.ignore_eos and tok == req.eos_token_id and req.eos_token_id is not None):
                        req.status = RequestStatus.FINISHED_STOPPED
                        stop = True
                    elif sp.stop_token_ids and tok in sp.stop_token_ids:
                        req.status = RequestStatus.FINISHED_STOPPED
                        req.stop_reason = tok
                        stop = True
                if not stop and ((req.max_tokens is not None and n_out >= req.max_tokens)
                                 or len(all_ids) >= max_len):
                    req.status = RequestStatus.FINISHED_LENGTH
                    stop = True
                if stop:
                    accepted = new_tokens[:i + 1]
                    break
            m = req.metrics
            if m.first_token_time is None:
                m.first_token_time = now
            m.num_generation_tokens = len(out_ids)
            finished = req.status != running_state
            if fwd_tokens and not finished:
                self._pending_new_tokens.setdefault(rid, []).extend(accepted)
            req_lps = lps.get(rid) if lps else None
            outputs.append(EngineCoreOutput(
                rid, accepted, req.status.finish_reason if finished else None, req.stop_reason,
                req_lps[:len(accepted)] if req_lps else None,
                req.num_cached_tokens if req.num_cached_tokens > 0 else 0))
            if runner_output.pooled and rid in runner_output.pooled:
                outputs[-1].pooled = runner_output.pooled[rid]
            if req.prompt_logprobs is not None:
                got = req.prompt_logprobs
                outputs[-1].prompt_logprobs = [got.get(i) for i in range(req.num_prompt_tokens)]
                req.prompt_logprobs = None            # sent once
            if finished:
                m.finished_time = now
                any_finished = True
                self._free_request(req)
            elif proposed is not None:
                req.spec_token_ids = proposed.get(rid) or []
        if any_finished:             # one pass instead of a list.remove per request
            self.running = [r for r in self.running if r.status == running_state]
        return outputs

    def _check_stop(self, req: Request, tok: int) -> bool:
        sp = req.sampling_params
        n_out = req.num_output_tokens
        if n_out >= sp.min_tokens:
            if not sp.ignore_eos and req.eos_token_id is not None and tok == req.eos_token_id:
                req.status = RequestStatus.FINISHED_STOPPED
                return True
            if tok in sp.stop_token_ids:
                req.status = RequestStatus.FINISHED_STOPPED
                req.stop_reason = tok
                return True
        if (req.max_tokens is not None and n_out >= req.max_tokens) or \
                req.num_tokens >= self.max_model_len:
            req.status = RequestStatus.FINISHED_LENGTH
            return True
        return False

    def finish_requests(self, request_ids: Iterable[str], status: RequestStatus) -> None:
        """Front-end initiated finish (stop string matched, client went away)."""
        for rid in request_ids:
            req = self.requests.get(rid)
            if req is None or req.finished:
                continue
            if self._orphan_if_loading(req, status):
                continue
            if req.status == RequestStatus.RUNNING:
                self.running.remove(req)
            else:
                try:
                    self.waiting.remove(req)
                except ValueError:
                    pass
            req.status = status
            self._free_request(req)

    def _orphan_if_loading(self, req: Request, status: RequestStatus) -> bool:
        """A request that ends while the workers are still copying its KV in: it leaves the books now,
        its blocks stay allocated until the copy has been reported finished."""
        if self.kv_loading.pop(req.request_id, None) is None:
            self._kv_loaded.discard(req.request_id)
            return False
        req.status = status
        self._kv_load_orphans[req.request_id] = req
        self._pending_new_tokens.pop(req.request_id, None)
        self.requests.pop(req.request_id, None)
        return True

    def _free_request(self, req: Request) -> None:
        if self.kv_connector is None or not self.kv_connector.request_finished(req):
            self.kv.free(req)                      # (a producer keeps the blocks while a save reads them)
        if req.request_id in self._known_to_workers:
            self._known_to_workers.discard(req.request_id)
            self._finished_since_last.add(req.request_id)
        self._pending_new_tokens.pop(req.request_id, None)
        self.requests.pop(req.request_id, None)

    # ----------------------------------------------------------------- stats
    def make_stats(self) -> SchedulerStats:
        return SchedulerStats(num_running=len(self.running), num_waiting=len(self.waiting),
                              kv_cache_usage=self.kv.usage,
                              prefix_cache_queries=self.kv.prefix_queries,
                              prefix_cache_hits=self.kv.prefix_hits,
                              num_preemptions=self.num_preemptions)

    def reset_prefix_cache(self) -> bool:
        return self.kv.reset_prefix_cache()
