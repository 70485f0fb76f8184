"""Bloom: the sequential Falcon block (``models/falcon.py``) with biases on every linear layer and norm,
ALiBi position biases (unscaled slopes, no rotary embedding), tanh-approximated GELU, a LayerNorm on the
embeddings, and optionally residual branches that start from the normalised input
(``apply_residual_connection_post_layernorm``).  ``query_key_value`` rows are stored ``[head][q, k, v]``.
Parity is pinned against ``transformers``' ``BloomForCausalLM``."""
from __future__ import annotations

from typing import Optional

from ..config import VllmConfig
from .falcon import FalconForCausalLM
from .layers.attention import LayerNorm
from .utils import IntermediateTensors


class BloomForCausalLM(FalconForCausalLM):
    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        cfg.update(alibi=True, alibi_kind="bloom", alibi_scaled_with_scores=False, bias=True, multi_query=False,
                   new_decoder_architecture=False, parallel_attn=False, activation="gelu_pytorch_tanh")
        cfg.setdefault("tie_word_embeddings", True)
        super().__init__(vllm_config)
        if self.is_first:
            self.word_embeddings_layernorm = LayerNorm(cfg["hidden_size"], cfg.get("layer_norm_epsilon", 1e-5),
                                                      vllm_config.model_config.torch_dtype)

    def forward(self, input_ids, positions, intermediate: Optional[IntermediateTensors] = None):
        hidden = self.word_embeddings_layernorm(self.embed_in(input_ids)) if self.is_first else intermediate.hidden
        for layer in self.layers.values():
            hidden = layer(positions, hidden)
        if not self.is_last:
            return IntermediateTensors(hidden, None)
        return self.final_layer_norm(hidden)
