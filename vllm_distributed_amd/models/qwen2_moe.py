"""Qwen2-MoE (Qwen1.5-MoE-A2.7B, Qwen2-57B-A14B): Qwen2 attention (biased q/k/v projections), and on the
sparse layers a routed expert mixture plus one always-on shared expert whose output is scaled per token by
``sigmoid(shared_expert_gate(x))``.  Parity is pinned against ``transformers``' ``Qwen2MoeForCausalLM``
(``tests/test_hf_parity.py``)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..parallel.state import get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce
from . import qwen3_moe
from .layers.linear import ReplicatedLinear
from .layers.moe import FusedMoE
from .llama import LlamaMLP
from .qwen3_moe import Qwen3MoeDecoderLayer, Qwen3MoeForCausalLM


class Qwen2MoeSparseBlock(FusedMoE):
    """Routed experts + gated shared expert.  Both halves produce partial sums under tensor
    parallelism, so they share one all-reduce."""

    def __init__(self, hidden: int, inter: int, shared_inter: int, num_experts: int, top_k: int,
                 renormalize: bool, dtype: torch.dtype, expert_parallel: bool = False):
        super().__init__(hidden, inter, num_experts, top_k, renormalize, dtype, expert_parallel=expert_parallel)
        self.shared_expert = LlamaMLP(hidden, shared_inter, dtype) if shared_inter > 0 else None
        self.shared_expert_gate = ReplicatedLinear(hidden, 1, False, dtype, quant=False)

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False) -> torch.Tensor:
        out = super().forward(x, reduce=False)
        if self.shared_expert is not None:
            shared = self.shared_expert(x, reduce=False)
            gate = torch.sigmoid(self.shared_expert_gate(x).float()).to(x.dtype)
            out = out + gate * shared
        if reduce and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class Qwen2MoeDecoderLayer(Qwen3MoeDecoderLayer):
    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        step = cfg.get("decoder_sparse_step", 1)
        sparse = (cfg.get("num_experts", 0) > 0 and self._layer_idx not in cfg.get("mlp_only_layers", [])
                  and (self._layer_idx + 1) % step == 0)
        if sparse:
            return Qwen2MoeSparseBlock(cfg["hidden_size"], cfg["moe_intermediate_size"],
                                       int(cfg.get("shared_expert_intermediate_size") or 0), cfg["num_experts"],
                                       cfg["num_experts_per_tok"], bool(cfg.get("norm_topk_prob", False)), dtype,
                                       expert_parallel=qwen3_moe._EP)
        return LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype)


class Qwen2MoeForCausalLM(Qwen3MoeForCausalLM):
    layer_cls = Qwen2MoeDecoderLayer

    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
