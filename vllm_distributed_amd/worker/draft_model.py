"""Draft-model speculative decoding, worker side.

A second, smaller model with the target's vocabulary lives on every worker next to the target.
It keeps its own KV cache over the SAME blocks (one block table and one slot mapping serve both
models), so the scheduler's paging decisions cover it for free.  Per step:

1. *catch up*: the draft model runs the step's scheduled tokens under the target's attention
   metadata, which writes its KV for exactly the positions the target wrote;
2. *propose*: starting from the token the target just sampled for each request, it decodes
   ``num_speculative_tokens`` tokens greedily, one batched decode step per token.  Those steps write
   draft KV past the scheduled positions; the scheduler allocates that far ahead
   (``Scheduler.lookahead``), and the next catch-up pass rewrites the same slots with the same
   tokens (or, after a rejection, with the corrected ones).

The proposals travel back in ``ModelRunnerOutput.draft_token_ids``; the scheduler hands them to the
next step as ``scheduled_spec_decode_tokens`` and the target verifies them like prompt-lookup drafts
(vLLM: ``--speculative-config '{"model": ..., "num_speculative_tokens": k}'``).
"""
from __future__ import annotations

import copy
import dataclasses
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import ops
from ..config import ModelConfig, VllmConfig
from ..logger import init_logger
from ..models.layers.attention import AttentionMetadata, set_forward_context
from ..models.loader import load_model

logger = init_logger(__name__)


class DraftProposer:
    needs_hidden = False          # True: drafts from the target's hidden states, after the step has sampled

    def _draft_model_config(self, vllm_config: VllmConfig) -> ModelConfig:
        sc, mc = vllm_config.speculative_config, vllm_config.model_config
        fmt = sc.load_format or ("auto" if mc.load_format == "sharded_state" else mc.load_format)
        return ModelConfig(model=sc.model, dtype=mc.dtype, seed=mc.seed, max_model_len=mc.max_model_len,
                           load_format=fmt, quantization=sc.quantization)

    def __init__(self, vllm_config: VllmConfig, device: torch.device):
        sc, mc = vllm_config.speculative_config, vllm_config.model_config
        dmc = self._draft_model_config(vllm_config)
        if int(dmc.hf_config["vocab_size"]) != int(mc.hf_config["vocab_size"]):
            raise ValueError(f"draft model {sc.model!r} has vocabulary {dmc.hf_config['vocab_size']}, "
                             f"the target {mc.hf_config['vocab_size']}: they must share a tokenizer")
        cfg = copy.copy(vllm_config)            # no __post_init__: the target's checks do not apply twice
        cfg.model_config, cfg.speculative_config, cfg.lora_config = dmc, None, None
        self.model_config = dmc
        self.model = load_model(cfg, device)
        self.k = int(sc.num_speculative_tokens)
        self.device = device
        self.kv_caches: List[Tuple[torch.Tensor, torch.Tensor]] = []
        logger.info("draft model %s loaded: %d tokens proposed per step", dmc.model, self.k)

    # ------------------------------------------------------------------ KV cache
    def kv_bytes_per_block(self, block_size: int, dtype_bytes: int) -> int:
        return sum(block_size * a.kv_values_per_token() * dtype_bytes for a in self.model.attention_layers())

    def allocate_kv(self, num_blocks: int, block_size: int, dtype: torch.dtype, fp8: bool) -> None:
        self.kv_caches = []
        for attn in self.model.attention_layers():
            k_shape, v_shape = attn.kv_cache_shapes(num_blocks, block_size)
            k = torch.zeros(k_shape, dtype=dtype, device=self.device)
            v = torch.zeros(v_shape, dtype=dtype, device=self.device)
            scales = attn.kv_scales if fp8 else None
            if scales is not None:
                scales = tuple(t.to(self.device) for t in scales)
            attn.bind_kv_cache(k, v, scales)
            self.kv_caches.append((k, v))

    def release_kv(self) -> None:
        for attn in self.model.attention_layers():
            attn.k_cache = attn.v_cache = None
        self.kv_caches = []

    # ------------------------------------------------------------------ per step
    def catch_up(self, md: AttentionMetadata, input_ids: torch.Tensor, positions: torch.Tensor) -> None:
        """Draft KV for the tokens this step ran (prompt chunks, decode tokens, drafts under
        verification), at the slots the target used."""
        md = dataclasses.replace(md, decode_workspace=None)     # sized for the target's heads
        set_forward_context(md, positions)
        try:
            self.model(input_ids, positions, None)
        finally:
            set_forward_context(None)

    def propose(self, last_tokens: torch.Tensor, start_pos: np.ndarray, block_table: np.ndarray,
                num_blocks: np.ndarray, block_size: int, max_model_len: int, to_dev) -> List[List[int]]:
        """``k`` greedy tokens after ``last_tokens[r]`` (which sits at position ``start_pos[r]``) for
        every row ``r``.  A row stops early where its blocks or the model length end."""
        r = int(start_pos.shape[0])
        # positions a row may still write: inside its allocated blocks and below max_model_len - 1
        # (the token proposed at the last step needs a position of its own)
        limit = np.minimum(num_blocks.astype(np.int64) * block_size, max_model_len - 1)
        room = np.clip(limit - start_pos, 0, self.k).astype(np.int64)
        steps = int(room.max()) if r else 0
        if steps <= 0:
            return [[] for _ in range(r)]
        width = int(min(block_table.shape[1], (int((start_pos + steps).max()) + block_size - 1) // block_size + 1))
        bt = to_dev(block_table[:, :width])
        qsl = to_dev(np.arange(r + 1, dtype=np.int32))
        cur = last_tokens.to(torch.int32)
        outs = []
        for i in range(steps):
            pos = start_pos.astype(np.int64) + i
            live = i < room
            safe = np.where(live, pos, 0)
            slot = np.where(live, block_table[np.arange(r), safe // block_size].astype(np.int64) * block_size
                            + safe % block_size, -1)
            seq = (safe + 1).astype(np.int32)
            md = AttentionMetadata(
                num_decodes=r, num_decode_tokens=r, num_prefills=0, num_prefill_tokens=0,
                slot_mapping=to_dev(slot.astype(np.int32)), block_table=bt, seq_lens=to_dev(seq),
                query_start_loc=qsl, max_query_len=1, max_seq_len=int(seq.max()),
                max_decode_seq_len=int(seq.max()), decode_workspace=None)
            positions = to_dev(safe.astype(np.int32))
            set_forward_context(md, positions)
            try:
                hidden = self.model(cur, positions, None)
            finally:
                set_forward_context(None)
            logits = self.model.compute_logits(hidden)
            cur = logits.float().argmax(dim=-1).to(torch.int32)
            outs.append(cur)
        toks = torch.stack(outs, dim=1).cpu().numpy()            # [r, steps]: the one host sync of the loop
        return [toks[j, :int(room[j])].tolist() for j in range(r)]


class HiddenStateProposer(DraftProposer):
    """EAGLE heads and DeepSeek's multi-token-prediction module (``models/spec_heads.py``): the head reads
    (token ``i + 1``, target hidden state ``i``) at position ``i`` and its output there predicts token ``i + 2``.

    Per step, AFTER the target has sampled: one pass over the step's tokens - each paired with its successor,
    which for a sampled position is the token the target just produced there - writes the head's KV at the
    target's slots and leaves, at every request's last kept position, the feature that predicts the first draft.
    ``k - 1`` single-token passes follow, each fed the previous draft and the head's own previous feature.
    Embedding and LM head are the target's unless the head's checkpoint brings its own."""
    needs_hidden = True
    # True: the head reads the target's residual stream at several depths (``model.last_aux``) instead of its output
    wants_aux = False

    def _draft_model_config(self, vllm_config: VllmConfig) -> ModelConfig:
        sc, mc = vllm_config.speculative_config, vllm_config.model_config
        self.wants_aux = sc.method == "eagle3"
        if sc.method == "mtp":
            if int(mc.hf_config.get("num_nextn_predict_layers") or 0) < 1:
                raise ValueError(f"{mc.model} has no multi-token-prediction module (num_nextn_predict_layers)")
            dmc = ModelConfig(model=sc.model or mc.model, dtype=mc.dtype, seed=mc.seed, max_model_len=mc.max_model_len,
                              load_format=sc.load_format or mc.load_format, quantization=sc.quantization)
            dmc.hf_config = dict(dmc.hf_config, architectures=["DeepSeekMTP"])
            return dmc
        dmc = super()._draft_model_config(vllm_config)
        if dmc.hf_config.get("model_type") not in ("llama", "mistral", "qwen2", "qwen3"):
            raise ValueError(f"EAGLE head {sc.model!r}: model_type {dmc.hf_config.get('model_type')!r} is not one "
                             f"of the Llama-shaped families this head is written for")
        arch = "Eagle3LlamaForCausalLM" if sc.method == "eagle3" else "EagleLlamaForCausalLM"
        dmc.hf_config = dict(dmc.hf_config, architectures=[arch])
        return dmc

    def bind_target(self, target: torch.nn.Module) -> None:
        """Share what the head's checkpoint did not bring."""
        head = self.model
        if self.wants_aux:
            if int(head.target_hidden) != int(target.cfg["hidden_size"]):
                raise ValueError(f"EAGLE-3 head written for a target of width {head.target_hidden}, this one is "
                                 f"{target.cfg['hidden_size']} wide")
            if not (getattr(target, "is_first", True) and getattr(target, "is_last", True)) \
                    or not hasattr(target, "aux_hidden_layers"):
                raise ValueError("an EAGLE-3 head needs a Llama-shaped target on one pipeline stage")
            n = int(target.cfg["num_hidden_layers"])
            taps = tuple(head.aux_layers(n))
            if len(taps) != 3 or not all(0 <= i < n for i in taps):
                raise ValueError(f"EAGLE-3 reads three of the target's {n} layers, got {taps}")
            target.aux_hidden_layers = taps
        if int(head.cfg["hidden_size"]) != int(target.cfg["hidden_size"]) and not (
                self.wants_aux and "embed_tokens.weight" not in set(getattr(head, "missing_optional", ()))):
            raise ValueError(f"draft head width {head.cfg['hidden_size']} != target width {target.cfg['hidden_size']}")
        missing = set(getattr(head, "missing_optional", ()))
        if "embed_tokens.weight" in missing:
            head.embed_tokens = target.embed_tokens
        if "lm_head.weight" in missing or getattr(head, "lm_head", None) is None:
            head.lm_head, head.tie = target.lm_head, target.tie
            if target.lm_head is None and "embed_tokens.weight" not in missing:
                logger.info("draft head: logits through its own embedding (the target ties embedding and head)")
        logger.info("draft head shares with the target: %s", ", ".join(sorted(missing)) or "nothing")

    def catch_up(self, md: AttentionMetadata, input_ids: torch.Tensor, positions: torch.Tensor) -> None:
        raise RuntimeError("a hidden-state head catches up in propose_after")

    def run_step_tokens(self, md: AttentionMetadata, shifted_ids: torch.Tensor, positions: torch.Tensor,
                        hidden: torch.Tensor) -> torch.Tensor:
        """The head over a step's tokens (``shifted_ids[i]`` = the token after position ``i``): KV written at the
        target's slots, features ``[T, H]`` returned."""
        md = dataclasses.replace(md, decode_workspace=None)
        set_forward_context(md, positions)
        try:
            return self.model(shifted_ids, positions, hidden)
        finally:
            set_forward_context(None)

    def propose(self, feats: torch.Tensor, start_pos: np.ndarray, block_table: np.ndarray,
                num_blocks: np.ndarray, block_size: int, max_model_len: int, to_dev) -> List[List[int]]:
        """``feats[r]``: the head's feature at row ``r``'s last kept position (``start_pos[r] - 1``).  It predicts
        the token after the one at ``start_pos[r]``; every further draft costs one pass at the next position."""
        r = int(start_pos.shape[0])
        if r == 0:
            return []
        limit = np.minimum(num_blocks.astype(np.int64) * block_size, max_model_len - 1)
        room = np.clip(limit - start_pos, 0, self.k - 1).astype(np.int64)       # passes a row may still run
        steps = int(room.max())
        to_target = getattr(self.model, "map_draft_ids", None) or (lambda ids: ids)     # (a reduced draft vocabulary)
        cur = to_target(self.model.compute_logits(feats).float().argmax(dim=-1).to(torch.int32))
        outs = [cur]
        if steps > 0:
            width = int(min(block_table.shape[1], (int((start_pos + steps).max()) + block_size - 1) // block_size + 1))
            bt = to_dev(block_table[:, :width])
            qsl = to_dev(np.arange(r + 1, dtype=np.int32))
        for i in range(steps):
            pos = start_pos.astype(np.int64) + i
            live = i < room
            safe = np.where(live, pos, 0)
            slot = np.where(live, block_table[np.arange(r), safe // block_size].astype(np.int64) * block_size
                            + safe % block_size, -1)
            seq = (safe + 1).astype(np.int32)
            md = AttentionMetadata(
                num_decodes=r, num_decode_tokens=r, num_prefills=0, num_prefill_tokens=0,
                slot_mapping=to_dev(slot.astype(np.int32)), block_table=bt, seq_lens=to_dev(seq),
                query_start_loc=qsl, max_query_len=1, max_seq_len=int(seq.max()),
                max_decode_seq_len=int(seq.max()), decode_workspace=None)
            positions = to_dev(safe.astype(np.int32))
            set_forward_context(md, positions)
            try:
                feats = self.model(cur, positions, feats)
            finally:
                set_forward_context(None)
            cur = to_target(self.model.compute_logits(feats).float().argmax(dim=-1).to(torch.int32))
            outs.append(cur)
        toks = torch.stack(outs, dim=1).cpu().numpy()            # [r, 1 + steps]: the one host sync of the loop
        return [toks[j, :1 + int(room[j])].tolist() for j in range(r)]


class MedusaProposer(HiddenStateProposer):
    """Medusa heads (``models/spec_heads.py:MedusaModel``): every draft of a request comes from the target's
    hidden state at its last kept position - no pass over the step's tokens, no cache of its own."""

    def _draft_model_config(self, vllm_config: VllmConfig) -> ModelConfig:
        dmc = DraftProposer._draft_model_config(self, vllm_config)
        dmc.hf_config = dict(dmc.hf_config, architectures=["MedusaModel"])
        return dmc

    def bind_target(self, target: torch.nn.Module) -> None:
        if int(self.model.cfg["hidden_size"]) != int(target.cfg["hidden_size"]):
            raise ValueError(f"Medusa heads of width {self.model.cfg['hidden_size']} on a target of width "
                             f"{target.cfg['hidden_size']}")
        self.k = min(self.k, self.model.num_heads)

    def kv_bytes_per_block(self, block_size: int, dtype_bytes: int) -> int:
        return 0

    def allocate_kv(self, num_blocks: int, block_size: int, dtype: torch.dtype, fp8: bool) -> None:
        self.kv_caches = []

    def run_step_tokens(self, md, shifted_ids, positions, hidden: torch.Tensor) -> torch.Tensor:
        return hidden

    def propose(self, feats: torch.Tensor, start_pos: np.ndarray, block_table: np.ndarray, num_blocks: np.ndarray,
                block_size: int, max_model_len: int, to_dev) -> List[List[int]]:
        r = int(start_pos.shape[0])
        if r == 0:
            return []
        room = np.clip(max_model_len - 2 - start_pos, 0, self.k).astype(np.int64)      # drafts sit at start + 1 ...
        toks = self.model(feats, self.k).cpu().numpy()
        return [toks[j, :int(room[j])].tolist() for j in range(r)]


def accepted_tails(tokens: np.ndarray, counts: Optional[np.ndarray],
                   drafts: Optional[List[Optional[List[int]]]]) -> Tuple[np.ndarray, np.ndarray]:
    """Per sampled request: the index (into ``tokens``) of the last token the step keeps, and how many
    drafts it accepted before it.  Without verification every request keeps its single token."""
    if counts is None:
        n = tokens.shape[0]
        return np.arange(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    last, acc = [], []
    off = 0
    toks = tokens.tolist()
    for j, c in enumerate(counts.tolist()):
        d = drafts[j] if drafts is not None else None
        n_acc = 0
        if d:
            while n_acc < len(d) and toks[off + n_acc] == d[n_acc]:
                n_acc += 1
        last.append(off + n_acc)
        acc.append(n_acc)
        off += c
    return np.asarray(last, dtype=np.int64), np.asarray(acc, dtype=np.int64)
