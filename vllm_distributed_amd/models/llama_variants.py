"""Families that are the Llama decoder with a few constants changed.

* Gemma: RMSNorm multiplies by ``1 + w`` (folded into the weight at load), the embedding output is
  scaled by ``sqrt(hidden)``, the MLP gate uses tanh-GELU, the LM head is the embedding.
* Granite: four multipliers - on the embedding output, on each residual branch (folded into the
  ``o_proj`` / ``down_proj`` weights at load, so the fused add + norm kernels stay as they are), on the
  attention scores (``attention_multiplier`` replaces ``1 / sqrt(head_dim)``) and a divisor on the logits.
* Gemma-2: Gemma-3's layer shape without the q/k norms and with tanh caps on attention scores and logits.
* Gemma-3 (text): Gemma's constants plus per-head q/k RMSNorm, a norm on the OUTPUT of the attention and
  of the MLP branch as well as on their inputs, sliding-window layers with their own rotary base next to
  global ones, ``query_pre_attn_scalar ** -0.5`` as the score scale and an optional tanh cap on the logits.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..config import layer_sliding_window
from .layers.attention import RMSNorm
from .llama import LlamaAttention, LlamaForCausalLM, LlamaMLP
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


class Gemma3DecoderLayer(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        eps, h = cfg.get("rms_norm_eps", 1e-6), cfg["hidden_size"]
        acfg = dict(cfg, attention_multiplier=float(cfg.get("query_pre_attn_scalar") or
                                                    (cfg.get("head_dim") or h // cfg["num_attention_heads"])) ** -0.5)
        self.self_attn = LlamaAttention(acfg, layer_idx, dtype, cos_sin)
        self.mlp = LlamaMLP(h, cfg["intermediate_size"], dtype, False,
                            cfg.get("hidden_activation") or cfg.get("hidden_act") or "gelu_pytorch_tanh")
        self.input_layernorm = RMSNorm(h, eps, dtype)
        self.post_attention_layernorm = RMSNorm(h, eps, dtype)
        self.pre_feedforward_layernorm = RMSNorm(h, eps, dtype)
        self.post_feedforward_layernorm = RMSNorm(h, eps, dtype)
        self.local = layer_sliding_window(cfg, layer_idx) is not None

    def forward(self, positions: torch.Tensor, hidden: Optional[torch.Tensor],
                residual: Optional[torch.Tensor]) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        """The whole stream travels in ``residual`` (``hidden`` None), the form the model's layer loop
        uses for "already added"."""
        stream = hidden if residual is None else (residual if hidden is None else hidden + residual)
        a = self.self_attn(positions, self.input_layernorm(stream))
        stream = stream + self.post_attention_layernorm(a)
        m = self.mlp(self.pre_feedforward_layernorm(stream))
        return None, stream + self.post_feedforward_layernorm(m)


class Gemma2ForCausalLM(GemmaForCausalLM):
    """Gemma-2: norms on both sides of each branch like Gemma-3, no q/k norms, one rotary base for the
    alternating sliding-window and global layers, tanh caps on the attention scores
    (``attn_logit_softcapping``, applied inside the attention kernels) and on the logits."""
    layer_cls = Gemma3DecoderLayer
    tp_fusions = False

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        y = super().compute_logits(hidden)
        cap = self.cfg.get("final_logit_softcapping")
        return y if not cap else torch.tanh(y.float() / float(cap)).mul_(float(cap)).to(y.dtype)


class Gemma3ForCausalLM(GemmaForCausalLM):
    layer_cls = Gemma3DecoderLayer
    tp_fusions = False

    def __init__(self, vllm_config):
        mc = vllm_config.model_config
        cfg = mc.hf_config
        # per-layer-type rotary settings: newer configs nest them, older ones carry
        # rope_theta (+ rope_scaling) for global and rope_local_base_freq for sliding layers
        rp = cfg.get("rope_parameters") or {}
        full, local = rp.get("full_attention") or {}, rp.get("sliding_attention") or {}
        cfg["rope_theta"] = full.get("rope_theta") or cfg.get("rope_theta") or 1000000.0
        if cfg.get("rope_scaling") is None and full.get("rope_type", "default") not in ("default", None):
            cfg["rope_scaling"] = {k: v for k, v in full.items() if k != "rope_theta"}
        local_theta = local.get("rope_theta") or cfg.get("rope_local_base_freq") or 10000.0
        super().__init__(vllm_config)
        head_dim = cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"]
        self.register_buffer("cos_sin_local", ops.build_cos_sin_cache(head_dim, mc.max_model_len, float(local_theta)),
                             persistent=False)
        self.rebind_buffers()

    def rebind_buffers(self) -> None:
        for layer in self.layers.values():
            layer.self_attn.cos_sin = self.cos_sin_local if layer.local else self.cos_sin_cache

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        y = super().compute_logits(hidden)
        cap = self.cfg.get("final_logit_softcapping")
        return y if not cap else torch.tanh(y.float() / float(cap)).mul_(float(cap)).to(y.dtype)
