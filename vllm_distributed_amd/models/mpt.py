"""MPT: the sequential Falcon block (``models/falcon.py``) without any bias (linear layers and LayerNorms),
ALiBi with MPT's own slope sequence (``alibi_bias_max``), exact GELU, an optional clamp on q / k / v and an
optional fixed softmax scale; ``Wqkv`` is stored ``[q | k | v]``.  Parity is pinned against ``transformers``'
``MptForCausalLM`` (its implementation has no biases and no q/k LayerNorm either: ``no_bias = false`` and
``qk_ln`` checkpoints are refused)."""
from __future__ import annotations

from typing import Optional

from ..config import VllmConfig
from .falcon import FalconForCausalLM
from .layers.attention import LayerNorm


class MptForCausalLM(FalconForCausalLM):
    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        ac = dict(cfg.get("attn_config") or {})
        if ac.get("qk_ln"):
            raise NotImplementedError("MPT checkpoints with LayerNorm on q and k (qk_ln)")
        if cfg.get("no_bias") is False:
            raise NotImplementedError("MPT checkpoints with biases (no_bias = false)")
        cfg.update(alibi=bool(ac.get("alibi", True)), alibi_kind="mpt", alibi_bias_max=ac.get("alibi_bias_max", 8),
                   alibi_scaled_with_scores=False, softmax_scale=ac.get("softmax_scale"), clip_qkv=ac.get("clip_qkv"),
                   bias=False, norm_bias=False, multi_query=False, new_decoder_architecture=False,
                   parallel_attn=False, activation="gelu", qkv_layout="concat",
                   ffn_hidden_size=int(cfg.get("expansion_ratio", 4)) * cfg["hidden_size"])
        if not cfg["alibi"]:
            raise NotImplementedError("MPT checkpoints with learned position embeddings (alibi = false)")
        cfg.setdefault("tie_word_embeddings", True)
        super().__init__(vllm_config)
        if self.is_last:
            self.final_layer_norm = LayerNorm(cfg["hidden_size"], cfg.get("layer_norm_epsilon", 1e-5),
                                              vllm_config.model_config.torch_dtype, bias=False)

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("transformer."):
            name = name[len("transformer."):]
        if name.startswith("blocks."):
            name = "layers." + name[len("blocks."):]
        return (name.replace("wte.", "embed_in.").replace("norm_f.", "final_layer_norm.")
                .replace("lm_head.", "embed_out.").replace(".norm_1.", ".input_layernorm.")
                .replace(".norm_2.", ".post_attention_layernorm.").replace(".attn.Wqkv.", ".query_key_value.")
                .replace(".attn.out_proj.", ".dense.").replace(".ffn.up_proj.", ".dense_h_to_4h.")
                .replace(".ffn.down_proj.", ".dense_4h_to_h."))
