"""Falcon (7B / 40B / 180B shapes, and the ALiBi variants: Falcon-RW): GPT-NeoX's block with
multi-query or grouped-query attention, no biases, and one of three norm arrangements -

* ``new_decoder_architecture`` (40B, 180B): ``ln_attn`` feeds the attention and ``ln_mlp`` the MLP, both
  applied to the layer input, branches added together (one shared ``input_layernorm`` when
  ``num_ln_in_parallel_attn`` is 1);
* ``parallel_attn`` (7B): one ``input_layernorm`` feeds both branches;
* otherwise the sequential form with ``post_attention_layernorm`` before the MLP.

The fused ``query_key_value`` rows are stored per KV group as ``[q heads of the group..., k, v]`` in the
new architecture, ``[all q heads, k, v]`` with ``multi_query``, and ``[head][q, k, v]`` otherwise; all three
are re-ordered to ``[q | k | v]`` at load.  Parity is pinned against ``transformers``' ``FalconForCausalLM``.

With ``alibi`` there is no rotary embedding: the attention kernels add ``slope * (key - query)`` to the scores
(``Attention.alibi_slopes``).  Falcon multiplies its bias by ``1 / sqrt(head_dim)`` together with the scores, so
the slopes are pre-scaled here; Bloom (``models/bloom.py``) and MPT (``models/mpt.py``) are this layer with other
switches and do not."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..config import VllmConfig
from .gpt_neox import GPTNeoXForCausalLM
from .layers.attention import Attention, LayerNorm
from .layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from .utils import materialize


def falcon_kv_heads(cfg: dict) -> int:
    if cfg.get("new_decoder_architecture"):
        return int(cfg.get("num_kv_heads") or cfg["num_attention_heads"])
    return 1 if cfg.get("multi_query", True) else cfg["num_attention_heads"]


class FalconLayer(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        h, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        eps, bias = cfg.get("layer_norm_epsilon", 1e-5), bool(cfg.get("bias", False))
        norm_bias = bool(cfg.get("norm_bias", True))
        self.head_dim = h // heads
        new = bool(cfg.get("new_decoder_architecture"))
        self.parallel = new or bool(cfg.get("parallel_attn", True))
        n_ln = cfg.get("num_ln_in_parallel_attn")
        self.two_norms = new and (2 if n_ln is None else int(n_ln)) == 2
        if self.two_norms:
            self.ln_attn, self.ln_mlp = LayerNorm(h, eps, dtype, norm_bias), LayerNorm(h, eps, dtype, norm_bias)
        else:
            self.input_layernorm = LayerNorm(h, eps, dtype, norm_bias)
            if not self.parallel:
                self.post_attention_layernorm = LayerNorm(h, eps, dtype, norm_bias)
        # Bloom's option: the residual branch starts from the normalised input instead of the raw one
        self.residual_post_ln = bool(cfg.get("apply_residual_connection_post_layernorm", False))
        self.clip_qkv = float(cfg["clip_qkv"]) if cfg.get("clip_qkv") else None
        self.query_key_value = QKVParallelLinear(h, self.head_dim, heads, falcon_kv_heads(cfg), bias, dtype)
        self.dense = RowParallelLinear(h, h, bias, dtype)
        scale = float(cfg.get("softmax_scale") or self.head_dim ** -0.5)
        self.attn = Attention(self.query_key_value.num_heads, self.head_dim, self.query_key_value.num_kv_heads,
                              scale, layer_idx)
        self.use_alibi = bool(cfg.get("alibi"))
        if self.use_alibi:
            from ..ops.torch_ref import alibi_slopes
            slopes = alibi_slopes(heads, cfg.get("alibi_kind", "bloom"), float(cfg.get("alibi_bias_max", 8)))
            if cfg.get("alibi_scaled_with_scores", True):
                slopes = slopes * scale
            qp = self.query_key_value
            self.register_buffer("alibi_slopes", slopes[qp.tp_rank * qp.num_heads:(qp.tp_rank + 1) * qp.num_heads]
                                 .contiguous(), persistent=False)
        inner = cfg.get("ffn_hidden_size") or 4 * h
        self.dense_h_to_4h = ColumnParallelLinear(h, [inner], bias, dtype)
        self.dense_4h_to_h = RowParallelLinear(inner, h, bias, dtype)
        self.act = cfg.get("activation") or "gelu"
        self.cos_sin = cos_sin

    def _attend(self, positions: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        qkv = self.query_key_value(x)
        if self.clip_qkv is not None:
            qkv = qkv.clamp_(-self.clip_qkv, self.clip_qkv)
        qs, ks = self.query_key_value.q_size, self.query_key_value.kv_size
        q, k, v = qkv[:, :qs].contiguous(), qkv[:, qs:qs + ks].contiguous(), qkv[:, qs + ks:]
        if self.use_alibi:
            self.attn.alibi_slopes = self.alibi_slopes.float()
        else:
            ops.rotary_embedding(positions, q, k, self.cos_sin, self.head_dim)
        return self.dense(self.attn(q, k, v))

    def _mlp(self, x: torch.Tensor) -> torch.Tensor:
        return self.dense_4h_to_h(ops.activation(self.dense_h_to_4h(x), self.act))

    def forward(self, positions: torch.Tensor, hidden: torch.Tensor) -> torch.Tensor:
        if self.two_norms:
            return hidden + self._attend(positions, self.ln_attn(hidden)) + self._mlp(self.ln_mlp(hidden))
        x = self.input_layernorm(hidden)
        if self.parallel:
            return hidden + self._attend(positions, x) + self._mlp(x)
        hidden = (x if self.residual_post_ln else hidden) + self._attend(positions, x)
        y = self.post_attention_layernorm(hidden)
        return (y if self.residual_post_ln else hidden) + self._mlp(y)


class FalconForCausalLM(GPTNeoXForCausalLM):
    layer_cls = FalconLayer

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        cfg["partial_rotary_factor"] = 1.0
        cfg["layer_norm_eps"] = cfg.get("layer_norm_epsilon", 1e-5)
        cfg.setdefault("tie_word_embeddings", True)
        super().__init__(vllm_config)

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("transformer."):
            name = name[len("transformer."):]
        if "rotary_emb.inv_freq" in name:
            return None
        if name.startswith("h."):
            name = "layers." + name[2:]
        return (name.replace("word_embeddings.", "embed_in.").replace("ln_f.", "final_layer_norm.")
                .replace("lm_head.", "embed_out.").replace(".self_attention.", ".").replace(".mlp.", "."))

    def _split_qkv(self, t: torch.Tensor) -> torch.Tensor:
        """Checkpoint row order of ``query_key_value`` -> ``[q | k | v]``, heads in order."""
        cfg = self.cfg
        heads, kvh = cfg["num_attention_heads"], falcon_kv_heads(cfg)
        d = cfg["hidden_size"] // heads
        rest = t.shape[1:]
        if cfg.get("new_decoder_architecture"):
            g = t.view(kvh, heads // kvh + 2, d, *rest)
            q, k, v = g[:, :-2].reshape(heads * d, *rest), g[:, -2].reshape(kvh * d, *rest), g[:, -1].reshape(kvh * d, *rest)
            return torch.cat([q, k, v], dim=0)
        if kvh == 1 or cfg.get("qkv_layout") == "concat":
            return t                                        # already [q heads | k | v]
        g = t.view(heads, 3, d, *rest).transpose(0, 1)
        return g.reshape(-1, *rest).contiguous()

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        loaded = []
        for name, w in weights:
            name = self.map_weight_name(name)
            if name is None or not self._owns_layer(name):
                continue
            p = params.get(name)
            if p is None:
                continue
            if ".query_key_value." in name:
                w = self._split_qkv(materialize(w))
            loader = getattr(p, "weight_loader", None)
            if loader is None:
                p.data.copy_(materialize(w))
            else:
                loader(p, w, None)
            loaded.append(name)
        return loaded
