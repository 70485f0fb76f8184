"""Persistent per-worker batch state.

Workers are told about a request once (prompt, sampling params, first block
ids); afterwards only deltas arrive.  State lives in row-indexed numpy arrays
so that preparing a step's device inputs is a handful of vectorised gathers,
independent of how many requests are running.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Optional

import numpy as np

from ..sampling_params import SamplingParams


class InputBatch:
    def __init__(self, max_num_reqs: int, max_model_len: int, block_size: int, vocab_size: int,
                 base_seed: int = 0):
        self.max_num_reqs = max_num_reqs
        self.max_model_len = max_model_len
        self.block_size = block_size
        self.vocab_size = vocab_size
        self.base_seed = base_seed
        self.max_blocks = (max_model_len + block_size - 1) // block_size
        n = max_num_reqs
        # calloc-backed: rows that are never used never become resident
        self.token_ids = np.zeros((n, max_model_len), dtype=np.int32)
        self.num_tokens = np.zeros(n, dtype=np.int32)         # positions assigned (prompt + outputs)
        # tokens whose *values* are on the host; lags num_tokens by the samples still in flight
        self.num_known = np.zeros(n, dtype=np.int32)
        self.row_gen = np.zeros(n, dtype=np.int64)             # bumped whenever a row changes owner
        self.num_prompt_tokens = np.zeros(n, dtype=np.int32)
        self.num_computed = np.zeros(n, dtype=np.int32)
        self.block_table = np.zeros((n, self.max_blocks), dtype=np.int32)
        self.num_blocks = np.zeros(n, dtype=np.int32)
        self.temperature = np.zeros(n, dtype=np.float32)
        self.top_p = np.ones(n, dtype=np.float32)
        self.top_k = np.full(n, -1, dtype=np.int32)
        self.min_p = np.zeros(n, dtype=np.float32)
        self.presence = np.zeros(n, dtype=np.float32)
        self.frequency = np.zeros(n, dtype=np.float32)
        self.repetition = np.ones(n, dtype=np.float32)
        self.num_logprobs = np.full(n, -1, dtype=np.int32)
        self.has_penalty = np.zeros(n, dtype=bool)
        self.num_prompt_logprobs = np.full(n, -1, dtype=np.int32)
        self.pooling = np.zeros(n, dtype=np.int8)           # 0 none, 1 last token, 2 mean over the prompt
        self.has_adjust = np.zeros(n, dtype=bool)             # logit bias / allow / ban lists / min_tokens
        self.lora_id = np.zeros(n, dtype=np.int32)              # 0 = base model
        self.req_ids: List[Optional[str]] = [None] * n
        self.row_of: Dict[str, int] = {}
        self.params: List[Optional[SamplingParams]] = [None] * n
        self.generators: List[Optional[np.random.Generator]] = [None] * n
        # constrained decoding: the request's grammar position and how many output tokens it has consumed
        self.grammar: List[Optional[object]] = [None] * n
        self.grammar_pos = np.zeros(n, dtype=np.int32)
        self._free_rows = list(range(n - 1, -1, -1))

    @property
    def num_reqs(self) -> int:
        return len(self.row_of)

    def add_request(self, req_id: str, prompt_token_ids: List[int], output_token_ids: List[int],
                    sp: SamplingParams, block_ids: List[int], num_computed: int, lora_id: int = 0) -> int:
        if req_id in self.row_of:
            self.remove_request(req_id)
        if not self._free_rows:
            raise RuntimeError("persistent batch is full (max_num_seqs exceeded)")
        row = self._free_rows.pop()
        n_p, n_o = len(prompt_token_ids), len(output_token_ids)
        if n_p + n_o > self.max_model_len:
            raise ValueError("request longer than max_model_len")
        self.token_ids[row, :n_p] = prompt_token_ids
        if n_o:
            self.token_ids[row, n_p:n_p + n_o] = output_token_ids
        self.num_tokens[row] = n_p + n_o
        self.num_known[row] = n_p + n_o
        self.row_gen[row] += 1
        self.num_prompt_tokens[row] = n_p
        self.num_computed[row] = num_computed
        nb = len(block_ids)
        self.block_table[row, :nb] = block_ids
        self.num_blocks[row] = nb
        self.temperature[row] = sp.temperature
        self.top_p[row] = sp.top_p
        self.top_k[row] = sp.top_k
        self.min_p[row] = sp.min_p
        self.presence[row] = sp.presence_penalty
        self.frequency[row] = sp.frequency_penalty
        self.repetition[row] = sp.repetition_penalty
        self.has_penalty[row] = sp.has_penalties
        self.has_adjust[row] = sp.has_logit_adjust
        self.num_prompt_logprobs[row] = -1 if sp.prompt_logprobs is None else sp.prompt_logprobs
        self.pooling[row] = {None: 0, "last": 1, "mean": 2}[sp.pooling]
        self.grammar[row] = None
        self.grammar_pos[row] = 0
        if sp.json_object:
            from ..ops.grammar import JsonGrammar
            self.grammar[row] = JsonGrammar()
        elif sp.guided_choice:
            from ..ops.grammar import ChoiceGrammar
            self.grammar[row] = ChoiceGrammar(sp.guided_choice)
        elif sp.guided_regex:
            from ..ops.grammar import RegexGrammar
            self.grammar[row] = RegexGrammar(sp.guided_regex)
        elif sp.guided_grammar:
            from ..ops.grammar import CfgGrammar
            self.grammar[row] = CfgGrammar(sp.guided_grammar)
        self.num_logprobs[row] = -1 if sp.logprobs is None else sp.logprobs
        self.lora_id[row] = lora_id
        self.req_ids[row] = req_id
        self.row_of[req_id] = row
        self.params[row] = sp
        if sp.greedy:
            self.generators[row] = None
        else:
            # independent of batch composition: seeded per request
            seed = sp.seed if sp.seed is not None else (
                (self.base_seed * 1000003 + zlib.crc32(req_id.encode())) & 0x7FFFFFFF)
            # masked, so that no value a client manages to send can raise here, inside a worker step
            self.generators[row] = np.random.default_rng(int(seed) & 0x7FFFFFFFFFFFFFFF)
        return row

    def remove_request(self, req_id: str) -> None:
        row = self.row_of.pop(req_id, None)
        if row is None:
            return
        self.req_ids[row] = None
        self.params[row] = None
        self.generators[row] = None
        self.num_tokens[row] = 0
        self.num_known[row] = 0
        self.row_gen[row] += 1
        self.num_computed[row] = 0
        self.num_blocks[row] = 0
        self.lora_id[row] = 0
        self.has_adjust[row] = False
        self.num_prompt_logprobs[row] = -1
        self.pooling[row] = 0
        self.grammar[row] = None
        self.has_penalty[row] = False
        self._free_rows.append(row)

    def append_blocks(self, row: int, block_ids: List[int]) -> None:
        if not block_ids:
            return
        nb = int(self.num_blocks[row])
        self.block_table[row, nb:nb + len(block_ids)] = block_ids
        self.num_blocks[row] = nb + len(block_ids)

    def append_tokens(self, row: int, token_ids: List[int]) -> None:
        n = int(self.num_tokens[row])
        self.token_ids[row, n:n + len(token_ids)] = token_ids
        self.num_tokens[row] = n + len(token_ids)
        self.num_known[row] = n + len(token_ids)

    def draw_uniform(self, rows: np.ndarray) -> np.ndarray:
        """One U(0,1) per row from the row's own generator (0.5 for greedy rows)."""
        out = np.full(len(rows), 0.5, dtype=np.float32)
        for i, r in enumerate(rows.tolist()):
            g = self.generators[r]
            if g is not None:
                u = g.random()
                out[i] = min(max(u, 1e-7), 1.0 - 1e-7)
        return out
