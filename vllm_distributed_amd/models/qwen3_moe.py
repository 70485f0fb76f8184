"""Qwen3-MoE (the architecture of the reference's documented deployment,
Qwen3-Coder-480B-A35B, ``.env.server:11``): Llama-style attention with per-head
q/k RMSNorm, and a sparse MoE MLP on every ``decoder_sparse_step``-th layer."""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Tuple

import torch
import torch.nn as nn

from ..config import VllmConfig
from .layers.linear import nvfp4_rename
from .layers.moe import FusedMoE
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP

_EP = False


class Qwen3MoeDecoderLayer(LlamaDecoderLayer):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        self._layer_idx = layer_idx
        super().__init__(cfg, layer_idx, dtype, cos_sin)

    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        step = cfg.get("decoder_sparse_step", 1)
        sparse = (cfg.get("num_experts", 0) > 0 and self._layer_idx not in cfg.get("mlp_only_layers", [])
                  and (self._layer_idx + 1) % step == 0)
        if sparse:
            return FusedMoE(cfg["hidden_size"], cfg["moe_intermediate_size"], cfg["num_experts"],
                            cfg["num_experts_per_tok"], bool(cfg.get("norm_topk_prob", True)),
                            dtype, expert_parallel=_EP)
        return LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype)


class Qwen3MoeForCausalLM(LlamaForCausalLM):
    layer_cls = Qwen3MoeDecoderLayer
    _expert_re = re.compile(r"(.*\.mlp)\.experts\.(\d+)\.(gate_proj|up_proj|down_proj)\.(weight|qweight|qzeros|scales|weight_packed|weight_scale|weight_global_scale|weight_scale_2|input_global_scale|input_scale)$")

    def __init__(self, vllm_config: VllmConfig):
        global _EP
        _EP = vllm_config.parallel_config.enable_expert_parallel
        super().__init__(vllm_config)

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        loaded: List[str] = []

        def others():
            """Expert tensors are placed as they stream by; everything else is handed on, still
            unread, to the dense loader (tensors are only valid while their file is open)."""
            for name, w in weights:
                mapped = self.map_weight_name(name)
                m = self._expert_re.match(mapped) if mapped else None
                if m is None:
                    yield name, w
                    continue
                if not self._owns_layer(mapped):
                    continue
                prefix, eid, proj, kind = m.group(1), int(m.group(2)), m.group(3), m.group(4)
                if self.quant_method == "nvfp4":
                    renamed = nvfp4_rename("." + kind, w)
                    if renamed is None:
                        continue
                    kind, w = renamed[0][1:], renamed[1]
                suffix = "" if kind == "weight" else "_" + kind
                if proj == "down_proj":
                    p = params[prefix + ".w2" + suffix]
                    p.weight_loader(p, w, (eid, 0))
                else:
                    p = params[prefix + ".w13" + suffix]
                    p.weight_loader(p, w, (eid, 0 if proj == "gate_proj" else 1))
                loaded.append(mapped)
        dense = super().load_weights(others())
        return loaded + dense


class OlmoeForCausalLM(Qwen3MoeForCausalLM):
    """OLMoE: every layer sparse, experts of ``intermediate_size``, top-k weights not renormalised by
    default, RMSNorm over the whole q and k projections (``llama.FullWidthRMSNorm``)."""

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        cfg.setdefault("moe_intermediate_size", cfg["intermediate_size"])
        cfg.setdefault("norm_topk_prob", False)
        super().__init__(vllm_config)
