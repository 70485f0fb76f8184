"""GPT-NeoX / Pythia: LayerNorm, a rotary embedding over the first ``rotary_pct`` of every head, the
attention and MLP branches added to the SAME input when ``use_parallel_residual`` is set, and a fused
``query_key_value`` weight stored head by head as ``[heads, (q, k, v), head_dim]`` rows."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..config import VllmConfig
from ..parallel.state import get_pp_group
from .layers.attention import Attention, LayerNorm
from .layers.linear import (ColumnParallelLinear, ParallelLMHead, QKVParallelLinear, RowParallelLinear,
                            VocabParallelEmbedding)
from .utils import IntermediateTensors, get_pp_layer_range, materialize


class GPTNeoXLayer(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        h, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        eps = cfg.get("layer_norm_eps", 1e-5)
        bias = bool(cfg.get("attention_bias", True))
        self.head_dim = h // heads
        self.parallel = bool(cfg.get("use_parallel_residual", True))
        self.input_layernorm = LayerNorm(h, eps, dtype)
        self.post_attention_layernorm = LayerNorm(h, eps, dtype)
        self.query_key_value = QKVParallelLinear(h, self.head_dim, heads, heads, bias, dtype)
        self.dense = RowParallelLinear(h, h, bias, dtype)
        self.attn = Attention(self.query_key_value.num_heads, self.head_dim, self.query_key_value.num_kv_heads,
                              self.head_dim ** -0.5, layer_idx)
        self.dense_h_to_4h = ColumnParallelLinear(h, [cfg["intermediate_size"]], True, dtype)
        self.dense_4h_to_h = RowParallelLinear(cfg["intermediate_size"], h, True, dtype)
        self.act = cfg.get("hidden_act", "gelu")
        self.cos_sin = cos_sin

    def _attend(self, positions: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        qkv = self.query_key_value(x)
        qs, ks = self.query_key_value.q_size, self.query_key_value.kv_size
        q, k, v = qkv[:, :qs].contiguous(), qkv[:, qs:qs + ks].contiguous(), qkv[:, qs + ks:]
        ops.rotary_embedding(positions, q, k, self.cos_sin, self.head_dim)
        return self.dense(self.attn(q, k, v))

    def _mlp(self, x: torch.Tensor) -> torch.Tensor:
        return self.dense_4h_to_h(ops.activation(self.dense_h_to_4h(x), self.act))

    def forward(self, positions: torch.Tensor, hidden: torch.Tensor) -> torch.Tensor:
        attn = self._attend(positions, self.input_layernorm(hidden))
        if self.parallel:
            return hidden + attn + self._mlp(self.post_attention_layernorm(hidden))
        hidden = hidden + attn
        return hidden + self._mlp(self.post_attention_layernorm(hidden))


class GPTNeoXForCausalLM(nn.Module):
    layer_cls = GPTNeoXLayer

    def __init__(self, vllm_config: VllmConfig):
        super().__init__()
        mc = vllm_config.model_config
        cfg, dtype = mc.hf_config, mc.torch_dtype
        self.cfg = cfg
        pp = get_pp_group()
        self.is_first, self.is_last = pp.is_first_rank, pp.is_last_rank
        self.start_layer, self.end_layer = get_pp_layer_range(
            cfg["num_hidden_layers"], pp.rank_in_group, pp.world_size,
            vllm_config.parallel_config.pp_layer_partition)
        h, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        head_dim = h // heads
        pct = cfg.get("partial_rotary_factor", cfg.get("rotary_pct", 0.25))
        theta = cfg.get("rope_theta") or cfg.get("rotary_emb_base", 10000.0)
        cos_sin = ops.build_cos_sin_cache(head_dim, mc.max_model_len, theta, cfg.get("rope_scaling"),
                                          rotary_dim=int(head_dim * pct))
        self.register_buffer("cos_sin_cache", cos_sin, persistent=False)
        tie = bool(cfg.get("tie_word_embeddings", False))
        if self.is_first or (tie and self.is_last):
            self.embed_in = VocabParallelEmbedding(cfg["vocab_size"], h, dtype)
        self.layers = nn.ModuleDict({str(i): self.layer_cls(cfg, i, dtype, self.cos_sin_cache)
                                     for i in range(self.start_layer, self.end_layer)})
        if self.is_last:
            self.final_layer_norm = LayerNorm(h, cfg.get("layer_norm_eps", 1e-5), dtype)
            self.embed_out = None if tie else ParallelLMHead(cfg["vocab_size"], h, dtype)

    def rebind_buffers(self) -> None:
        for layer in self.layers.values():
            layer.cos_sin = self.cos_sin_cache

    def attention_layers(self) -> List[Attention]:
        return [layer.attn for layer in self.layers.values()]

    def forward(self, input_ids, positions, intermediate: Optional[IntermediateTensors] = None):
        hidden = self.embed_in(input_ids) if self.is_first else intermediate.hidden
        for layer in self.layers.values():
            hidden = layer(positions, hidden)
        if not self.is_last:
            return IntermediateTensors(hidden, None)
        return self.final_layer_norm(hidden)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        if self.embed_out is not None:
            return self.embed_out.logits(hidden)
        y = torch.nn.functional.linear(hidden, self.embed_in.weight)
        if self.embed_in.tp > 1:
            from ..parallel.state import tensor_model_parallel_all_gather
            y = tensor_model_parallel_all_gather(y, dim=-1)
        return y[..., :self.embed_in.vocab_size]

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("gpt_neox."):
            name = name[len("gpt_neox."):]
        if "rotary_emb.inv_freq" in name or name.endswith((".attention.bias", ".attention.masked_bias")):
            return None
        return name.replace(".attention.", ".").replace(".mlp.", ".")

    def _owns_layer(self, name: str) -> bool:
        if not name.startswith("layers."):
            return True
        return self.start_layer <= int(name.split(".")[1]) < self.end_layer

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        heads = self.cfg["num_attention_heads"]
        loaded = []
        for name, w in weights:
            name = self.map_weight_name(name)
            if name is None or not self._owns_layer(name):
                continue
            p = params.get(name)
            if p is None:
                continue
            if ".query_key_value." in name:
                # [heads, 3, head_dim, ...] rows -> [q | k | v], each head-major
                t = materialize(w)
                t = t.view(heads, 3, t.shape[0] // (3 * heads), *t.shape[1:]).transpose(0, 1)
                w = t.reshape(-1, *t.shape[3:]).contiguous()
            loader = getattr(p, "weight_loader", None)
            if loader is None:
                p.data.copy_(materialize(w))
            else:
                loader(p, w, None)
            loaded.append(name)
        return loaded
