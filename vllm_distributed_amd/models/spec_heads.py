"""Draft heads for speculative decoding that read the TARGET model's hidden states.

Both heads here predict token ``i + 2`` from the pair (embedding of token ``i + 1``, the target's final hidden
state at position ``i``): one cheap pass over a step's tokens yields a first draft per request, and feeding a
head its own output as the next "hidden state" yields further ones.

* ``EagleLlamaForCausalLM`` - EAGLE (Li et al., 2024): ``fc([embed ; hidden])`` followed by Llama decoder layers,
  the first of them without an input norm, and no final norm; logits through the target's LM head.  Checkpoints
  (``yuhuili/EAGLE-*``) carry ``fc.*`` and ``layers.0.*`` and usually their own copy of ``embed_tokens``; names
  are accepted with and without a ``model.`` prefix.  What a checkpoint leaves out (embedding, LM head) is shared
  with the target.
* ``DeepSeekMTP`` - the multi-token-prediction module DeepSeek-V3 checkpoints carry as layer
  ``num_hidden_layers``: ``eh_proj([enorm(embed) ; hnorm(hidden)])`` -> one full decoder layer ->
  ``shared_head.norm`` -> LM head.  Embedding and head are the target's.

vLLM serves both through ``--speculative-config '{"method": "eagle" | "mtp", ...}'``; the reference framework
reaches them through the vLLM app it launches (/root/reference/src/launch.py:429).  The reference ships no
fixtures for either: the tests pin these heads against plain PyTorch re-statements of the same formulas.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from ..config import VllmConfig
from .deepseek import DeepseekV3ForCausalLM
from .layers.attention import RMSNorm
from .layers.linear import ReplicatedLinear
from .llama import LlamaForCausalLM


class EagleLlamaForCausalLM(LlamaForCausalLM):
    # parameters a head checkpoint may leave out: the proposer binds the target's modules instead
    optional_parameters = ("embed_tokens.weight", "lm_head.weight")

    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
        cfg, dtype = self.cfg, vllm_config.model_config.torch_dtype
        h = cfg["hidden_size"]
        self.fc = ReplicatedLinear(2 * h, h, bool(cfg.get("bias", True)), dtype, quant=False)
        first = next(iter(self.layers.values()))
        first.input_layernorm = None            # EAGLE's first layer attends over the fc output as it is
        self.norm = None
        self.missing_optional: List[str] = []

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, hidden_in: torch.Tensor) -> torch.Tensor:
        """The head's feature per token, ``[T, H]``: both the input of ``compute_logits`` and the
        ``hidden_in`` of the next draft step."""
        x = self.fc(torch.cat([self.embed_tokens(input_ids), hidden_in.to(self.fc.weight.dtype)], dim=-1))
        hidden, residual = x, None
        for layer in self.layers.values():
            if layer.input_layernorm is None:
                residual = hidden
                attn = layer.self_attn(positions, hidden)
                hidden, residual = layer.post_attention_layernorm(attn, residual)
                hidden = layer.mlp(hidden)
            else:
                hidden, residual = layer(positions, hidden, residual)
                if hidden is None:
                    hidden = torch.zeros_like(residual)
                elif not isinstance(hidden, torch.Tensor):
                    hidden = hidden.materialize()
        return hidden + residual

    def fused_tp_mode(self, num_tokens: int) -> Optional[bool]:
        return None

    def map_weight_name(self, name: str) -> Optional[str]:
        name = super().map_weight_name(name)
        if name is None or name.endswith("layers.0.input_layernorm.weight") or name == "norm.weight":
            return None
        return name


class DeepSeekMTP(DeepseekV3ForCausalLM):
    """One multi-token-prediction module of a DeepSeek-V3 checkpoint: the tensors under
    ``model.layers.<num_hidden_layers>.``.  Built as a one-layer instance of the target family, so the layer's
    tensors go through the family's own loader (expert stacking, rope row order)."""
    optional_parameters = ("embed_tokens.weight", "lm_head.weight")

    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
        cfg, dtype = self.cfg, vllm_config.model_config.torch_dtype
        h, eps = cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-6)
        self.enorm = RMSNorm(h, eps, dtype)
        self.hnorm = RMSNorm(h, eps, dtype)
        self.eh_proj = ReplicatedLinear(2 * h, h, False, dtype, quant=False)
        self.missing_optional: List[str] = []

    def layer_range(self, cfg: dict, vllm_config: VllmConfig) -> Tuple[int, int]:
        n = int(cfg["num_hidden_layers"])
        return n, n + 1

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, hidden_in: torch.Tensor) -> torch.Tensor:
        e = self.enorm(self.embed_tokens(input_ids))
        x = self.eh_proj(torch.cat([e, self.hnorm(hidden_in.to(e.dtype))], dim=-1))
        (layer,) = self.layers.values()
        hidden, residual = layer(positions, x, None)
        if hidden is None:
            return residual
        if not isinstance(hidden, torch.Tensor):
            hidden = hidden.materialize()
        return hidden + residual

    def fused_tp_mode(self, num_tokens: int) -> Optional[bool]:
        return None

    def compute_logits(self, feature: torch.Tensor) -> torch.Tensor:
        return super().compute_logits(self.norm(feature))          # norm = the module's shared_head.norm

    def map_weight_name(self, name: str) -> Optional[str]:
        prefix = f"model.layers.{self.start_layer}."
        if not name.startswith(prefix) or "rotary_emb.inv_freq" in name:
            return None
        rest = name[len(prefix):]
        if rest.startswith(("embed_tokens.", "shared_head.head.")):
            return None                         # copies of the target's embedding and LM head
        if rest.startswith("shared_head.norm."):
            return "norm." + rest[len("shared_head.norm."):]
        if rest.startswith(("enorm.", "hnorm.", "eh_proj.")):
            return rest
        return f"layers.{self.start_layer}.{rest}"
