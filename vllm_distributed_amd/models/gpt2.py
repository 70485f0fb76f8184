"""GPT-2: learned absolute positions, pre-LayerNorm blocks, fused ``c_attn`` and ``Conv1D`` weights
(stored ``[in, out]``: transposed on load), tanh-GELU MLP, LM head tied to the token embedding."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import VllmConfig
from ..parallel.state import get_pp_group
from .layers.attention import Attention, LayerNorm
from .layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear, VocabParallelEmbedding
from .utils import IntermediateTensors, get_pp_layer_range, materialize


class GPT2Block(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype):
        super().__init__()
        h, heads = cfg["n_embd"], cfg["n_head"]
        inner = cfg.get("n_inner") or 4 * h
        eps = cfg.get("layer_norm_epsilon", 1e-5)
        self.head_dim = h // heads
        self.ln_1 = LayerNorm(h, eps, dtype)
        # GPT-BigCode (StarCoder): one K/V head shared by every query head
        kv_heads = 1 if cfg.get("multi_query") else heads
        self.c_attn = QKVParallelLinear(h, self.head_dim, heads, kv_heads, True, dtype)
        self.c_proj = RowParallelLinear(h, h, True, dtype)
        scale = self.head_dim ** -0.5 if cfg.get("scale_attn_weights", True) else 1.0
        self.attn = Attention(self.c_attn.num_heads, self.head_dim, self.c_attn.num_kv_heads, scale, layer_idx)
        self.ln_2 = LayerNorm(h, eps, dtype)
        self.c_fc = ColumnParallelLinear(h, [inner], True, dtype)
        self.mlp_c_proj = RowParallelLinear(inner, h, True, dtype)
        self.act = cfg.get("activation_function") or "gelu_new"

    def forward(self, hidden: torch.Tensor) -> torch.Tensor:
        qkv = self.c_attn(self.ln_1(hidden))
        qs, ks = self.c_attn.q_size, self.c_attn.kv_size
        hidden = hidden + self.c_proj(self.attn(qkv[:, :qs], qkv[:, qs:qs + ks], qkv[:, qs + ks:]))
        return hidden + self.mlp_c_proj(ops.activation(self.c_fc(self.ln_2(hidden)), self.act))


class GPT2LMHeadModel(nn.Module):
    _TRANSPOSED = (".c_attn.weight", ".c_proj.weight", ".c_fc.weight", ".mlp_c_proj.weight")

    def __init__(self, vllm_config: VllmConfig):
        super().__init__()
        mc = vllm_config.model_config
        cfg, dtype = mc.hf_config, mc.torch_dtype
        cfg.setdefault("n_embd", cfg.get("hidden_size"))
        cfg.setdefault("n_head", cfg.get("num_attention_heads"))
        cfg.setdefault("n_layer", cfg.get("num_hidden_layers"))
        self.cfg = cfg
        pp = get_pp_group()
        self.is_first, self.is_last = pp.is_first_rank, pp.is_last_rank
        self.start_layer, self.end_layer = get_pp_layer_range(
            cfg["n_layer"], pp.rank_in_group, pp.world_size, vllm_config.parallel_config.pp_layer_partition)
        h = cfg["n_embd"]
        if self.is_first or self.is_last:          # tied LM head lives on the last stage too
            self.wte = VocabParallelEmbedding(cfg["vocab_size"], h, dtype)
        if self.is_first:
            self.wpe = nn.Embedding(cfg.get("n_positions", 1024), h, dtype=dtype)
            self.wpe.weight.requires_grad_(False)
        self.h = nn.ModuleDict({str(i): GPT2Block(cfg, i, dtype) for i in range(self.start_layer, self.end_layer)})
        if self.is_last:
            self.ln_f = LayerNorm(h, cfg.get("layer_norm_epsilon", 1e-5), dtype)

    def rebind_buffers(self) -> None:
        return None

    def attention_layers(self) -> List[Attention]:
        return [blk.attn for blk in self.h.values()]

    def forward(self, input_ids, positions, intermediate: Optional[IntermediateTensors] = None):
        if self.is_first:
            hidden = self.wte(input_ids) + self.wpe(positions.long())
        else:
            hidden = intermediate.hidden
        for blk in self.h.values():
            hidden = blk(hidden)
        if not self.is_last:
            return IntermediateTensors(hidden, None)
        return self.ln_f(hidden)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        y = F.linear(hidden, self.wte.weight)
        if self.wte.tp > 1:
            from ..parallel.state import tensor_model_parallel_all_gather
            y = tensor_model_parallel_all_gather(y, dim=-1)
        return y[..., :self.wte.vocab_size]

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("transformer."):
            name = name[len("transformer."):]
        if name.startswith("lm_head.") or name.endswith((".attn.bias", ".attn.masked_bias")):
            return None                    # tied head; causal-mask buffers of old checkpoints
        return name.replace(".attn.c_attn.", ".c_attn.").replace(".attn.c_proj.", ".c_proj.") \
                   .replace(".mlp.c_fc.", ".c_fc.").replace(".mlp.c_proj.", ".mlp_c_proj.")

    def _owns_layer(self, name: str) -> bool:
        if not name.startswith("h."):
            return True
        return self.start_layer <= int(name.split(".")[1]) < self.end_layer

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
            if name.endswith(self._TRANSPOSED):
                w = materialize(w).t()
            loader = getattr(p, "weight_loader", None)
            if loader is None:
                p.data.copy_(materialize(w))
            else:
                loader(p, w, None)
            loaded.append(name)
        return loaded


class GPTBigCodeForCausalLM(GPT2LMHeadModel):
    """GPT-2's block with multi-query attention and ``nn.Linear`` weights (stored ``[out, in]``)."""
    _TRANSPOSED = ()

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        if not cfg.get("multi_query", True):
            raise NotImplementedError("GPT-BigCode checkpoints without multi-query attention store q/k/v "
                                      "interleaved per head; only multi_query=True is supported")
        cfg["multi_query"] = True
        cfg.setdefault("activation_function", "gelu_pytorch_tanh")
        super().__init__(vllm_config)
