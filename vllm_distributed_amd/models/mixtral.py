"""Mixtral: Llama/Mistral attention with a sparse MoE MLP on every layer
(``block_sparse_moe``: router ``gate``, experts ``w1`` = gate, ``w3`` = up,
``w2`` = down; softmax over all experts, top-k, renormalised)."""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from ..config import VllmConfig
from .layers.linear import nvfp4_rename
from .layers.moe import FusedMoE
from .llama import LlamaDecoderLayer, LlamaForCausalLM

_EP = False


class MixtralDecoderLayer(LlamaDecoderLayer):
    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return FusedMoE(cfg["hidden_size"], cfg["intermediate_size"], cfg["num_local_experts"],
                        cfg["num_experts_per_tok"], True, dtype, expert_parallel=_EP)


class MixtralForCausalLM(LlamaForCausalLM):
    layer_cls = MixtralDecoderLayer
    _expert_re = re.compile(r"(.*\.mlp)\.experts\.(\d+)\.(w1|w2|w3)\.(weight|qweight|qzeros|scales|weight_packed|weight_scale|weight_global_scale|weight_scale_2|input_global_scale|input_scale)$")

    def __init__(self, vllm_config: VllmConfig):
        global _EP
        _EP = vllm_config.parallel_config.enable_expert_parallel
        super().__init__(vllm_config)

    def map_weight_name(self, name: str) -> Optional[str]:
        name = super().map_weight_name(name)
        return None if name is None else name.replace(".block_sparse_moe.", ".mlp.")

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
                if proj == "w2":
                    p = params[prefix + ".w2" + suffix]
                    p.weight_loader(p, w, (eid, 0))
                else:
                    p = params[prefix + ".w13" + suffix]
                    p.weight_loader(p, w, (eid, 0 if proj == "w1" else 1))
                loaded.append(mapped)
        dense = super().load_weights(others())
        return loaded + dense


class GraniteMoeForCausalLM(MixtralForCausalLM):
    """GraniteMoE: Granite's four multipliers (embedding, residual branches, attention scores, logits) around
    Mixtral-style experts whose weights are stored stacked: ``input_linear.weight [E, 2I, H]`` = [gate ; up] per
    expert, ``output_linear.weight [E, H, I]``, router ``router.layer.weight`` (newer ``transformers`` write the
    same tensors as ``experts.gate_up_proj`` / ``experts.down_proj`` / ``router.weight``).  The residual
    multiplier is folded into ``o_proj`` and the experts' output weights at load."""

    _STACKED = {"input_linear.weight": "w13", "experts.gate_up_proj": "w13",
                "output_linear.weight": "w2", "experts.down_proj": "w2"}

    def __init__(self, vllm_config: VllmConfig):
        if vllm_config.model_config.quantization in ("awq", "nvfp4"):
            raise NotImplementedError("packed 4-bit GraniteMoE checkpoints (stacked expert tensors)")
        super().__init__(vllm_config)

    def embedding_scale(self, cfg: dict) -> Optional[float]:
        m = cfg.get("embedding_multiplier")
        return float(m) if m not in (None, 1.0) else None

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        y = super().compute_logits(hidden)
        s = self.cfg.get("logits_scaling")
        return y if s in (None, 1.0) else y / float(s)

    def map_weight_name(self, name: str) -> Optional[str]:
        name = super().map_weight_name(name)
        if name is None:
            return None
        return name.replace(".mlp.router.layer.", ".mlp.gate.").replace(".mlp.router.", ".mlp.gate.")

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        from .utils import materialize
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        mult = float(self.cfg.get("residual_multiplier") or 1.0)
        loaded: List[str] = []

        def others():
            for name, w in weights:
                mapped = self.map_weight_name(name)
                kind = next((v for k, v in self._STACKED.items() if mapped and mapped.endswith(".mlp." + k)), None)
                if kind is None:
                    if mapped and mapped.endswith((".o_proj.weight", ".o_proj.bias")) and mult != 1.0:
                        w = materialize(w).float().mul(mult)
                    yield name, w
                    continue
                if not self._owns_layer(mapped):
                    continue
                p = params[mapped[:mapped.index(".mlp.")] + ".mlp." + kind]
                t = materialize(w)
                half = t.shape[1] // 2
                for e in range(t.shape[0]):
                    if kind == "w13":
                        p.weight_loader(p, t[e, :half], (e, 0))
                        p.weight_loader(p, t[e, half:], (e, 1))
                    else:
                        p.weight_loader(p, t[e].float().mul(mult).to(t.dtype) if mult != 1.0 else t[e], (e, 0))
                loaded.append(mapped)
        dense = LlamaForCausalLM.load_weights(self, others())
        return loaded + dense
