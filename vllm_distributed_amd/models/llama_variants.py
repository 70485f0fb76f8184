"""Families that are the Llama decoder with a few constants changed.

* Gemma: RMSNorm multiplies by ``1 + w`` (folded into the weight at load), the embedding output is
  scaled by ``sqrt(hidden)``, the MLP gate uses tanh-GELU, the LM head is the embedding.
* Granite: four multipliers - on the embedding output, on each residual branch (folded into the
  ``o_proj`` / ``down_proj`` weights at load, so the fused add + norm kernels stay as they are), on the
  attention scores (``attention_multiplier`` replaces ``1 / sqrt(head_dim)``) and a divisor on the logits.
"""
from __future__ import annotations

from typing import Optional

import torch

from .llama import LlamaForCausalLM
from .utils import materialize


class GemmaForCausalLM(LlamaForCausalLM):
    def __init__(self, vllm_config):
        vllm_config.model_config.hf_config.setdefault("tie_word_embeddings", True)
        super().__init__(vllm_config)

    def embedding_scale(self, cfg: dict) -> Optional[float]:
        return float(cfg["hidden_size"]) ** 0.5

    def transform_weight(self, name: str, w):
        if name.endswith("norm.weight"):
            return materialize(w).float().add(1.0)
        return w


class GraniteForCausalLM(LlamaForCausalLM):
    def embedding_scale(self, cfg: dict) -> Optional[float]:
        m = cfg.get("embedding_multiplier")
        return float(m) if m not in (None, 1.0) else None

    def transform_weight(self, name: str, w):
        m = self.cfg.get("residual_multiplier")
        if m not in (None, 1.0) and (".o_proj." in name or ".down_proj." in name) and name.endswith(("weight", "bias")):
            # the whole tensor is read here; a row-parallel loader then narrows it to its shard
            return materialize(w).float().mul(float(m))
        return w

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        y = super().compute_logits(hidden)
        s = self.cfg.get("logits_scaling")
        return y if s in (None, 1.0) else y / float(s)
