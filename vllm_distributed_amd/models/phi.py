"""Phi-1 / Phi-1.5 / Phi-2: one LayerNorm per layer feeding the attention and the MLP branch side by side
(both added to the layer input), biased projections, a rotary embedding over the first
``partial_rotary_factor`` of every head, a tanh-GELU ``fc1 -> fc2`` MLP and an LM head with a bias."""
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


class PhiLayer(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        h, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        kv = cfg.get("num_key_value_heads") or heads
        if cfg.get("qk_layernorm"):
            raise ValueError("Phi checkpoints with qk_layernorm are not supported")
        self.head_dim = h // heads
        self.input_layernorm = LayerNorm(h, cfg.get("layer_norm_eps", 1e-5), dtype)
        attn_bias = cfg.get("model_type") != "gptj"         # GPT-J: biases in the MLP only
        self.qkv_proj = QKVParallelLinear(h, self.head_dim, heads, kv, attn_bias, dtype)
        self.dense = RowParallelLinear(h, h, attn_bias, dtype)
        self.attn = Attention(self.qkv_proj.num_heads, self.head_dim, self.qkv_proj.num_kv_heads,
                              self.head_dim ** -0.5, layer_idx, pad_head_dim=True)
        self.fc1 = ColumnParallelLinear(h, [cfg["intermediate_size"]], True, dtype)
        self.fc2 = RowParallelLinear(cfg["intermediate_size"], h, True, dtype)
        self.act = cfg.get("hidden_act", "gelu_new")
        self.cos_sin = cos_sin

    def forward(self, positions: torch.Tensor, hidden: torch.Tensor) -> torch.Tensor:
        x = self.input_layernorm(hidden)
        qkv = self.qkv_proj(x)
        qs, ks = self.qkv_proj.q_size, self.qkv_proj.kv_size
        q, k, v = qkv[:, :qs].contiguous(), qkv[:, qs:qs + ks].contiguous(), qkv[:, qs + ks:]
        ops.rotary_embedding(positions, q, k, self.cos_sin, self.head_dim)
        attn = self.dense(self.attn(q, k, v))
        return hidden + attn + self.fc2(ops.activation(self.fc1(x), self.act))


class PhiForCausalLM(nn.Module):
    stacked_params = [(".q_proj", ".qkv_proj", "q"), (".k_proj", ".qkv_proj", "k"), (".v_proj", ".qkv_proj", "v")]

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
        h = cfg["hidden_size"]
        head_dim = h // cfg["num_attention_heads"]
        cos_sin = ops.build_cos_sin_cache(head_dim, mc.max_model_len, cfg.get("rope_theta", 10000.0),
                                          cfg.get("rope_scaling"),
                                          rotary_dim=int(head_dim * cfg.get("partial_rotary_factor", 0.5)))
        self.register_buffer("cos_sin_cache", cos_sin, persistent=False)
        if self.is_first:
            self.embed_tokens = VocabParallelEmbedding(cfg["vocab_size"], h, dtype)
        self.layers = nn.ModuleDict({str(i): PhiLayer(cfg, i, dtype, self.cos_sin_cache)
                                     for i in range(self.start_layer, self.end_layer)})
        if self.is_last:
            self.final_layernorm = LayerNorm(h, cfg.get("layer_norm_eps", 1e-5), dtype)
            self.lm_head = ParallelLMHead(cfg["vocab_size"], h, dtype, bias=True)

    def rebind_buffers(self) -> None:
        for layer in self.layers.values():
            layer.cos_sin = self.cos_sin_cache

    def attention_layers(self) -> List[Attention]:
        return [layer.attn for layer in self.layers.values()]

    def forward(self, input_ids, positions, intermediate: Optional[IntermediateTensors] = None):
        hidden = self.embed_tokens(input_ids) if self.is_first else intermediate.hidden
        for layer in self.layers.values():
            hidden = layer(positions, hidden)
        if not self.is_last:
            return IntermediateTensors(hidden, None)
        return self.final_layernorm(hidden)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        return self.lm_head.logits(hidden)

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("model."):
            name = name[len("model."):]
        if "rotary_emb.inv_freq" in name:
            return None
        return name.replace(".self_attn.", ".").replace(".mlp.", ".")

    def transform_weight(self, name: str, w):
        return w

    def _owns_layer(self, name: str) -> bool:
        if not name.startswith("layers."):
            return True
        return self.start_layer <= int(name.split(".")[1]) < self.end_layer

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        loaded = []
        for name, w in weights:
            name = self.map_weight_name(name)
            if name is None or not self._owns_layer(name):
                continue
            w = self.transform_weight(name, w)
            target, shard = name, None
            for frag, fused, sid in self.stacked_params:
                if frag in name:
                    target, shard = name.replace(frag, fused), sid
                    break
            p = params.get(target)
            if p is None:
                continue
            loader = getattr(p, "weight_loader", None)
            if loader is None:
                p.data.copy_(materialize(w))
            else:
                loader(p, w, shard)
            loaded.append(name)
        return loaded


class GPTJForCausalLM(PhiForCausalLM):
    """GPT-J: Phi's layer shape (one LayerNorm feeding attention and MLP side by side, LM head with a
    bias) without attention biases, and a rotary embedding over ADJACENT pairs of the first
    ``rotary_dim`` dimensions: the q / k rows of every head are de-interleaved at load, after which the
    half-apart rotation of the first ``rotary_dim`` dimensions computes the same scores."""

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        h, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        self._head_dim = h // heads
        self._rotary_dim = int(cfg.get("rotary_dim") or self._head_dim)
        cfg["intermediate_size"] = cfg.get("n_inner") or 4 * h
        cfg["partial_rotary_factor"] = self._rotary_dim / self._head_dim
        cfg["layer_norm_eps"] = cfg.get("layer_norm_epsilon", 1e-5)
        cfg["hidden_act"] = cfg.get("activation_function") or "gelu_new"
        cfg["num_key_value_heads"] = heads
        super().__init__(vllm_config)
        rd, d = self._rotary_dim, self._head_dim
        self._perm = torch.cat([torch.arange(0, rd, 2), torch.arange(1, rd, 2), torch.arange(rd, d)])

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("transformer."):
            name = name[len("transformer."):]
        if name.endswith((".attn.bias", ".attn.masked_bias")):
            return None                                     # causal-mask buffers of old checkpoints
        if name.startswith("h."):
            name = "layers." + name[2:]
        return (name.replace("wte.", "embed_tokens.").replace("ln_f.", "final_layernorm.")
                .replace(".ln_1.", ".input_layernorm.").replace(".attn.out_proj.", ".dense.")
                .replace(".attn.", ".").replace(".mlp.fc_in.", ".fc1.").replace(".mlp.fc_out.", ".fc2."))

    def transform_weight(self, name: str, w):
        if ".q_proj." in name or ".k_proj." in name:
            t = materialize(w)
            return t.reshape(-1, self._head_dim, *t.shape[1:]).index_select(1, self._perm.to(t.device)).reshape(t.shape)
        return w
