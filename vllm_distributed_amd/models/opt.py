"""OPT decoder: learned positions (offset 2), LayerNorm, ReLU, biases
everywhere.  Used by the CPU/gloo plumbing configuration (OPT-125m)."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..parallel.state import get_pp_group
from .layers.attention import Attention, LayerNorm
from .layers.linear import (ColumnParallelLinear, QKVParallelLinear, ReplicatedLinear,
                            RowParallelLinear, VocabParallelEmbedding)
from .utils import IntermediateTensors, get_pp_layer_range, materialize


class OPTDecoderLayer(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype):
        super().__init__()
        h, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        bias = bool(cfg.get("enable_bias", True))
        self.head_dim = h // heads
        self.pre_ln = bool(cfg.get("do_layer_norm_before", True))
        self.qkv_proj = QKVParallelLinear(h, self.head_dim, heads, heads, bias, dtype)
        self.out_proj = RowParallelLinear(h, h, bias, dtype)
        self.attn = Attention(self.qkv_proj.num_heads, self.head_dim, self.qkv_proj.num_kv_heads,
                              self.head_dim ** -0.5, layer_idx)
        affine = bool(cfg.get("layer_norm_elementwise_affine", True))
        self.self_attn_layer_norm = LayerNorm(h, 1e-5, dtype, bias=affine)
        self.final_layer_norm = LayerNorm(h, 1e-5, dtype, bias=affine)
        self.fc1 = ColumnParallelLinear(h, [cfg["ffn_dim"]], bias, dtype)
        self.fc2 = RowParallelLinear(cfg["ffn_dim"], h, bias, dtype)
        self.act = {"relu": F.relu, "gelu": F.gelu}[cfg.get("activation_function", "relu")]

    def forward(self, hidden: torch.Tensor) -> torch.Tensor:
        residual = hidden
        if self.pre_ln:
            hidden = self.self_attn_layer_norm(hidden)
        qkv = self.qkv_proj(hidden)
        qs, ks = self.qkv_proj.q_size, self.qkv_proj.kv_size
        q, k, v = qkv[:, :qs], qkv[:, qs:qs + ks], qkv[:, qs + ks:]
        hidden = residual + self.out_proj(self.attn(q, k, v))
        if not self.pre_ln:
            hidden = self.self_attn_layer_norm(hidden)
        residual = hidden
        if self.pre_ln:
            hidden = self.final_layer_norm(hidden)
        hidden = residual + self.fc2(self.act(self.fc1(hidden)))
        if not self.pre_ln:
            hidden = self.final_layer_norm(hidden)
        return hidden


class OPTForCausalLM(nn.Module):
    stacked_params = [(".q_proj", ".qkv_proj", "q"), (".k_proj", ".qkv_proj", "k"),
                      (".v_proj", ".qkv_proj", "v")]
    POSITION_OFFSET = 2

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
        proj = cfg.get("word_embed_proj_dim", h)
        if self.is_first or self.is_last:     # tied LM head lives on the last stage too
            self.embed_tokens = VocabParallelEmbedding(cfg["vocab_size"], proj, dtype)
        if self.is_first:
            self.embed_positions = nn.Embedding(
                cfg["max_position_embeddings"] + self.POSITION_OFFSET, h, dtype=dtype)
            self.embed_positions.weight.requires_grad_(False)
            self.project_in = ReplicatedLinear(proj, h, False, dtype) if proj != h else None
        self.layers = nn.ModuleDict({str(i): OPTDecoderLayer(cfg, i, dtype)
                                     for i in range(self.start_layer, self.end_layer)})
        if self.is_last:
            ln = cfg.get("do_layer_norm_before", True) and not cfg.get("_remove_final_layer_norm", False)
            self.final_layer_norm = LayerNorm(h, 1e-5, dtype) if ln else None
            self.project_out = ReplicatedLinear(h, proj, False, dtype) if proj != h else None

    def rebind_buffers(self) -> None:
        return None

    def attention_layers(self) -> List[Attention]:
        return [layer.attn for layer in self.layers.values()]

    def forward(self, input_ids, positions, intermediate: Optional[IntermediateTensors] = None):
        if self.is_first:
            hidden = self.embed_tokens(input_ids)
            if self.project_in is not None:
                hidden = self.project_in(hidden)
            hidden = hidden + self.embed_positions(positions.long() + self.POSITION_OFFSET)
        else:
            hidden = intermediate.hidden
        for layer in self.layers.values():
            hidden = layer(hidden)
        if not self.is_last:
            return IntermediateTensors(hidden, None)
        if self.final_layer_norm is not None:
            hidden = self.final_layer_norm(hidden)
        if self.project_out is not None:
            hidden = self.project_out(hidden)
        return hidden

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        w = self.embed_tokens
        y = F.linear(hidden, w.weight)
        if w.tp > 1:
            from ..parallel.state import tensor_model_parallel_all_gather
            y = tensor_model_parallel_all_gather(y, dim=-1)
        return y[..., :w.vocab_size]

    def map_weight_name(self, name: str) -> Optional[str]:
        for prefix in ("model.decoder.", "decoder.", "model."):
            if name.startswith(prefix):
                name = name[len(prefix):]
                break
        if name.startswith("lm_head."):
            return None          # tied to embed_tokens
        return name.replace(".self_attn.", ".")

    def _owns_layer(self, name: str) -> bool:
        if not name.startswith("layers."):
            return True
        idx = int(name.split(".")[1])
        return self.start_layer <= idx < self.end_layer

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        loaded = []
        for name, w in weights:
            name = self.map_weight_name(name)
            if name is None or not self._owns_layer(name):
                continue
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
