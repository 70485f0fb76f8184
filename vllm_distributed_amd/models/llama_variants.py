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
* SmolLM3: layers listed in ``no_rope_layers`` carry no positional embedding.
* Seed-OSS, Ministral: Llama's layer with other bias switches / per-layer windows (no code here).
* Arcee: the MLP is ``up -> relu^2 -> down`` without a gate.
* GLM-4: GLM with a norm on the output of each branch as well.
* OLMo-2 / OLMo-3: norms on the branch OUTPUTS only, RMSNorm over the whole q and k projections;
  OLMo-3's sliding-window layers use the unscaled rotary base.
* OLMo: LayerNorms without parameters, optional clamp on q / k / v.
* EXAONE-4: OLMo-2's layer shape with per-head q/k norms; with a sliding window configured, the global
  layers carry no positional embedding.
* StableLM: LayerNorm (with bias) where Llama has RMSNorm, rotary embedding over a quarter of each head.
* Nemotron: LayerNorm whose stored weight is ``w - 1``, ``up -> relu^2 -> down`` MLP, partial rotary.
* Cohere (Command-R): one bias-free LayerNorm per layer feeding attention and MLP side by side, rotary
  embedding over adjacent pairs, logits multiplied by ``logit_scale``, tied embeddings.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..config import layer_sliding_window
from ..parallel.state import get_tp_group
from .layers.attention import LayerNorm, RMSNorm
from .layers.linear import ColumnParallelLinear, RowParallelLinear
from .llama import LlamaAttention, LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, identity_cos_sin
from .utils import materialize

QKV_ONLY = [(".q_proj", ".qkv_proj", "q"), (".k_proj", ".qkv_proj", "k"), (".v_proj", ".qkv_proj", "v")]


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
        """``residual_multiplier`` folded into the two projections that feed the residual stream.  For a
        dense tensor that is the weight (and bias) itself; a packed 4-bit tensor must not be touched -
        there the factor goes into what the kernels multiply the integers with: the per-group ``scales``
        of AWQ (zero points are subtracted before the scale, so they stay), the per-matrix
        ``weight_alpha`` of NVFP4 (names as ``nvfp4_rename`` leaves them)."""
        m = self.cfg.get("residual_multiplier")
        if m in (None, 1.0) or not (".o_proj." in name or ".down_proj." in name):
            return w
        if name.endswith((".weight", ".bias", ".scales", ".weight_alpha")):
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


# ------------------------------------------------------------------ layers without a positional embedding
class _PerLayerRope(LlamaForCausalLM):
    """Models whose layers do not all share one cos / sin table: ``layer_table(i)`` names the buffer
    layer ``i`` rotates with (``cos_sin_cache``, or one a subclass registered; ``cos_sin_identity``
    rotates nothing)."""

    def __init__(self, vllm_config):
        super().__init__(vllm_config)
        self.register_buffer("cos_sin_identity", identity_cos_sin(self.cos_sin_cache), persistent=False)
        self.rebind_buffers()

    def layer_table(self, layer_idx: int) -> str:
        return "cos_sin_cache"

    def rebind_buffers(self) -> None:
        for i, layer in self.layers.items():
            layer.self_attn.cos_sin = getattr(self, self.layer_table(int(i)), self.cos_sin_cache)


class SmolLM3ForCausalLM(_PerLayerRope):
    def layer_table(self, layer_idx: int) -> str:
        use = self.cfg.get("no_rope_layers")            # 1 = rotary embedding, 0 = none
        if use is None:
            every = int(self.cfg.get("no_rope_layer_interval") or 4)
            return "cos_sin_identity" if (layer_idx + 1) % every == 0 else "cos_sin_cache"
        return "cos_sin_cache" if use[layer_idx] else "cos_sin_identity"


# ------------------------------------------------------------------------------------------ MLP without a gate
class PlainMLP(nn.Module):
    def __init__(self, hidden: int, inter: int, dtype: torch.dtype, bias: bool, act: str):
        super().__init__()
        self.up_proj = ColumnParallelLinear(hidden, [inter], bias, dtype)
        self.down_proj = RowParallelLinear(inter, hidden, bias, dtype)
        self.act = act

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False) -> torch.Tensor:
        return self.down_proj(ops.activation(self.up_proj(x), self.act), reduce, staged)


class ArceeDecoderLayer(LlamaDecoderLayer):
    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return PlainMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype, bool(cfg.get("mlp_bias", False)),
                        cfg.get("hidden_act") or "relu2")


class ArceeForCausalLM(LlamaForCausalLM):
    layer_cls = ArceeDecoderLayer
    stacked_params = QKV_ONLY


# ------------------------------------------------------------------ norms on other sides of the branches
class BranchNormLayer(nn.Module):
    """Decoder layer with a norm on any of: attention input, attention output, MLP input, MLP output
    (``norm_names``, None = no norm there).  The stream travels in ``residual`` (``hidden`` None)."""
    norm_names: Tuple[Optional[str], ...] = ("input_layernorm", None, "post_attention_layernorm", None)

    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        self.self_attn = LlamaAttention(self.attention_config(cfg), layer_idx, dtype, cos_sin)
        self.mlp = self.make_mlp(cfg, dtype)
        for name in self.norm_names:
            if name is not None:
                setattr(self, name, self.make_norm(cfg, dtype))
        self.local = layer_sliding_window(cfg, layer_idx) is not None

    def attention_config(self, cfg: dict) -> dict:
        return cfg

    def make_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return RMSNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-5), dtype)

    def make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype, bool(cfg.get("mlp_bias", False)),
                        cfg.get("hidden_act") or "silu")

    def forward(self, positions: torch.Tensor, hidden: Optional[torch.Tensor],
                residual: Optional[torch.Tensor]) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        stream = hidden if residual is None else (residual if hidden is None else hidden + residual)
        pre_a, post_a, pre_m, post_m = (getattr(self, n) if n else None for n in self.norm_names)
        a = self.self_attn(positions, stream if pre_a is None else pre_a(stream))
        stream = stream + (a if post_a is None else post_a(a))
        m = self.mlp(stream if pre_m is None else pre_m(stream))
        return None, stream + (m if post_m is None else post_m(m))


class Glm4DecoderLayer(BranchNormLayer):
    norm_names = ("input_layernorm", "post_self_attn_layernorm", "post_attention_layernorm", "post_mlp_layernorm")


class Glm4ForCausalLM(LlamaForCausalLM):
    layer_cls = Glm4DecoderLayer
    tp_fusions = False


class Olmo2DecoderLayer(BranchNormLayer):
    norm_names = (None, "post_attention_layernorm", None, "post_feedforward_layernorm")


class Olmo2ForCausalLM(LlamaForCausalLM):
    layer_cls = Olmo2DecoderLayer
    tp_fusions = False


class Olmo3ForCausalLM(_PerLayerRope):
    """Global layers rotate with the (possibly scaled) table, sliding-window layers with the plain one."""
    layer_cls = Olmo2DecoderLayer
    tp_fusions = False

    def __init__(self, vllm_config):
        mc = vllm_config.model_config
        cfg = mc.hf_config
        rp = cfg.get("rope_parameters") or {}
        full = rp.get("full_attention") or rp
        local = rp.get("sliding_attention") or {}
        cfg["rope_theta"] = full.get("rope_theta") or cfg.get("rope_theta") or 10000.0
        if cfg.get("rope_scaling") is None and full.get("rope_type", "default") not in ("default", None):
            cfg["rope_scaling"] = {k: v for k, v in full.items() if k != "rope_theta"}
        self._local_theta = float(local.get("rope_theta") or cfg["rope_theta"])
        super().__init__(vllm_config)
        self.register_buffer("cos_sin_local", ops.build_cos_sin_cache(self.head_dim, mc.max_model_len,
                                                                      self._local_theta), persistent=False)
        self.rebind_buffers()

    def layer_table(self, layer_idx: int) -> str:
        return "cos_sin_local" if self.layers[str(layer_idx)].local else "cos_sin_cache"


class Exaone4ForCausalLM(_PerLayerRope):
    layer_cls = Olmo2DecoderLayer
    tp_fusions = False

    def layer_table(self, layer_idx: int) -> str:
        hybrid = bool(self.cfg.get("sliding_window"))
        return "cos_sin_identity" if hybrid and not self.layers[str(layer_idx)].local else "cos_sin_cache"


# ------------------------------------------------------------------------------------ LayerNorm families
class PlainLayerNorm(nn.Module):
    """LayerNorm without weight or bias (OLMo)."""

    def __init__(self, hidden: int, eps: float = 1e-5):
        super().__init__()
        self.hidden, self.eps = hidden, eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.layer_norm(x.float(), (self.hidden,), None, None, self.eps).to(x.dtype)


class OlmoDecoderLayer(BranchNormLayer):
    def make_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return PlainLayerNorm(cfg["hidden_size"])


class OlmoForCausalLM(LlamaForCausalLM):
    """OLMo (1): Llama with parameter-free LayerNorms and an optional clamp on q / k / v."""
    layer_cls = OlmoDecoderLayer
    tp_fusions = False

    def make_final_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return PlainLayerNorm(cfg["hidden_size"])


class StableLmDecoderLayer(BranchNormLayer):
    def attention_config(self, cfg: dict) -> dict:
        if cfg.get("qk_layernorm") or cfg.get("use_parallel_residual"):
            raise NotImplementedError("StableLM checkpoints with per-head q/k LayerNorm or a parallel residual "
                                      "(stablelm-2-12b) are not supported")
        return dict(cfg, attention_bias=bool(cfg.get("use_qkv_bias", False)), attention_out_bias=False)

    def make_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return LayerNorm(cfg["hidden_size"], cfg.get("layer_norm_eps", 1e-5), dtype)


class StableLmForCausalLM(LlamaForCausalLM):
    layer_cls = StableLmDecoderLayer
    tp_fusions = False

    def make_final_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return LayerNorm(cfg["hidden_size"], cfg.get("layer_norm_eps", 1e-5), dtype)


class NemotronDecoderLayer(BranchNormLayer):
    def make_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return LayerNorm(cfg["hidden_size"], cfg.get("norm_eps", 1e-5), dtype)

    def make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return PlainMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype, bool(cfg.get("mlp_bias", False)),
                        cfg.get("hidden_act") or "relu2")


class NemotronForCausalLM(LlamaForCausalLM):
    layer_cls = NemotronDecoderLayer
    tp_fusions = False
    stacked_params = QKV_ONLY

    def make_final_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return LayerNorm(cfg["hidden_size"], cfg.get("norm_eps", 1e-5), dtype)

    def transform_weight(self, name: str, w):
        if name.endswith("norm.weight"):
            return materialize(w).float().add(1.0)
        return w


class CohereDecoderLayer(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        if cfg.get("use_qk_norm"):
            raise NotImplementedError("Cohere checkpoints with use_qk_norm are not supported")
        self.self_attn = LlamaAttention(cfg, layer_idx, dtype, cos_sin)
        self.mlp = LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype, False, cfg.get("hidden_act") or "silu")
        self.input_layernorm = LayerNorm(cfg["hidden_size"], cfg.get("layer_norm_eps", 1e-5), dtype, bias=False)

    def forward(self, positions: torch.Tensor, hidden: Optional[torch.Tensor],
                residual: Optional[torch.Tensor]) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        stream = hidden if residual is None else (residual if hidden is None else hidden + residual)
        x = self.input_layernorm(stream)
        # the two branches' partial sums share one all-reduce
        both = self.self_attn(positions, x, reduce=False) + self.mlp(x, reduce=False)
        tp = get_tp_group()
        if tp.world_size > 1:
            both = tp.all_reduce(both)
        return None, stream + both


class CohereForCausalLM(LlamaForCausalLM):
    layer_cls = CohereDecoderLayer
    tp_fusions = False

    def __init__(self, vllm_config):
        vllm_config.model_config.hf_config.setdefault("tie_word_embeddings", True)
        super().__init__(vllm_config)

    def make_final_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return LayerNorm(cfg["hidden_size"], cfg.get("layer_norm_eps", 1e-5), dtype, bias=False)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        y = super().compute_logits(hidden)
        s = self.cfg.get("logit_scale")
        return y if s in (None, 1.0) else y * float(s)
class Cohere2ForCausalLM(_PerLayerRope, CohereForCausalLM):
    """Command-R7B / Command-A: Cohere's parallel block with a sliding window on most layers (``layer_types``, or
    every layer but each ``sliding_window_pattern``-th) - and the global layers carry no positional embedding."""

    def __init__(self, vllm_config):
        cfg = vllm_config.model_config.hf_config
        if not cfg.get("layer_types"):
            every = int(cfg.get("sliding_window_pattern") or 4)
            cfg["layer_types"] = ["full_attention" if (i + 1) % every == 0 else "sliding_attention"
                                  for i in range(int(cfg["num_hidden_layers"]))]
        super().__init__(vllm_config)

    def layer_table(self, layer_idx: int) -> str:
        return "cos_sin_cache" if self.cfg["layer_types"][layer_idx] == "sliding_attention" else "cos_sin_identity"


