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


class _MoEWithSharedMLP(FusedMoE):
    """Routed experts plus one dense SwiGLU MLP every token goes through (GraniteMoeShared)."""

    def __init__(self, hidden: int, inter: int, experts: int, top_k: int, shared_inter: int, dtype: torch.dtype):
        super().__init__(hidden, inter, experts, top_k, True, dtype, expert_parallel=_EP)
        from .llama import LlamaMLP
        self.shared = LlamaMLP(hidden, shared_inter, dtype)

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False, deferred: bool = False):
        return super().forward(x, reduce, staged) + self.shared(x, reduce, staged)


class GraniteMoeSharedDecoderLayer(MixtralDecoderLayer):
    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        shared = int(cfg.get("shared_intermediate_size") or 0)
        if shared <= 0:
            return super()._make_mlp(cfg, dtype)
        return _MoEWithSharedMLP(cfg["hidden_size"], cfg["intermediate_size"], cfg["num_local_experts"],
                                 cfg["num_experts_per_tok"], shared, dtype)


class GraniteMoeSharedForCausalLM(GraniteMoeForCausalLM):
    """GraniteMoE with a shared dense MLP next to the experts (``shared_mlp.input_linear`` = [gate ; up],
    ``shared_mlp.output_linear``); the residual multiplier is folded into its output weights as well."""
    layer_cls = GraniteMoeSharedDecoderLayer

    def map_weight_name(self, name: str) -> Optional[str]:
        name = name.replace(".shared_mlp.input_linear.", ".mlp.shared.gate_up_proj.") \
                   .replace(".shared_mlp.output_linear.", ".mlp.shared.down_proj.")
        return super().map_weight_name(name)

    def transform_weight(self, name: str, w):
        mult = float(self.cfg.get("residual_multiplier") or 1.0)
        if name.endswith(".mlp.shared.down_proj.weight") and mult != 1.0:
            from .utils import materialize
            return materialize(w).float().mul(mult)
        return super().transform_weight(name, w)


# ------------------------------------------------------------------------------------------------- Phi-3.5-MoE
class SparseMixerMoE(FusedMoE):
    """Phi-3.5-MoE's router (``sparsemixer`` at inference).  The first expert is the arg-max; its weight is its share
    of a softmax over the experts whose logit lies within ``2 * jitter_eps`` (relative) of the maximum.  The second
    expert is chosen the same way among the rest, with a softmax of its own: the two weights are not renormalised."""
    custom_routing = True
    jitter_eps = 0.01

    def router_logits(self, x: torch.Tensor) -> torch.Tensor:
        # fp32: the thresholds below are hard, and a 16-bit logit puts an expert on the wrong side of one
        return torch.nn.functional.linear(x.float(), self.gate.weight.float())

    def route(self, logits: torch.Tensor):
        def pick(scores: torch.Tensor):
            top, idx = scores.max(dim=-1, keepdim=True)
            factor = logits.abs().clamp(min=top)
            far = ((top - logits) / factor) > (2 * self.jitter_eps)
            gates = torch.softmax(scores.masked_fill(far, float("-inf")), dim=-1)
            return gates.gather(-1, idx), idx
        w1, i1 = pick(logits)
        w2, i2 = pick(logits.scatter(-1, i1, float("-inf")))
        return torch.cat([w1, w2], dim=-1), torch.cat([i1, i2], dim=-1)


class PhimoeForCausalLM(MixtralForCausalLM):
    """Phi-3.5-MoE: Mixtral's layout (experts ``block_sparse_moe.experts.N.w1 / w3 / w2``, router
    ``block_sparse_moe.gate``) with LayerNorms (weight and bias), biases on the attention projections and the LM
    head, LongRoPE, and the ``sparsemixer`` router."""
    tp_fusions = False

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        if int(cfg.get("num_experts_per_tok") or 2) != 2:
            raise NotImplementedError("Phi-MoE routing is written for two experts per token")
        super().__init__(vllm_config)
        if self.is_last and cfg.get("lm_head_bias") and self.lm_head is not None:
            from .layers.linear import ParallelLMHead
            self.lm_head = ParallelLMHead(cfg["vocab_size"], cfg["hidden_size"], vllm_config.model_config.torch_dtype,
                                          bias=True)

    def make_final_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        from .layers.attention import LayerNorm
        return LayerNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-5), dtype)


def _phimoe_layer():
    from .layers.attention import LayerNorm
    from .llama_variants import BranchNormLayer

    class PhimoeDecoderLayer(BranchNormLayer):
        def make_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
            return LayerNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-5), dtype)

        def make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
            moe = SparseMixerMoE(cfg["hidden_size"], cfg["intermediate_size"], cfg["num_local_experts"], 2, False, dtype,
                                 expert_parallel=_EP)
            moe.jitter_eps = float(cfg.get("router_jitter_noise") or 0.01)
            return moe

    return PhimoeDecoderLayer


PhimoeForCausalLM.layer_cls = _phimoe_layer()


# -------------------------------------------------------------------------------------------------------- DBRX
def _dbrx_layer():
    from .layers.attention import LayerNorm
    from .llama_variants import BranchNormLayer

    class DbrxDecoderLayer(BranchNormLayer):
        def make_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
            return LayerNorm(cfg["hidden_size"], 1e-5, dtype, bias=False)

        def make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
            return FusedMoE(cfg["hidden_size"], cfg["intermediate_size"], cfg["num_local_experts"],
                            cfg["num_experts_per_tok"], True, dtype, expert_parallel=_EP)

    return DbrxDecoderLayer


class DbrxForCausalLM(MixtralForCausalLM):
    """DBRX: bias-free LayerNorms, a fused and clamped ``Wqkv``, 16 experts of which 4 run, their weights stacked
    over experts in three tensors (``w1`` gate and ``v1`` up as ``[E * I, H]``, ``w2`` down as ``[E * I, H]`` too,
    i.e. transposed), router weights renormalised to sum one."""
    tp_fusions = False
    layer_cls = _dbrx_layer()

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        norm = float((cfg.get("ffn_config") or {}).get("moe_normalize_expert_weights") or 0.0)
        if norm != 1.0:
            raise NotImplementedError(f"DBRX with moe_normalize_expert_weights={norm} (only the 1-norm)")
        if vllm_config.model_config.quantization in ("awq", "nvfp4"):
            raise NotImplementedError("packed 4-bit DBRX checkpoints (stacked expert tensors)")
        super().__init__(vllm_config)

    def make_final_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        from .layers.attention import LayerNorm
        return LayerNorm(cfg["hidden_size"], 1e-5, dtype, bias=False)

    _RENAMES = ((".norm_attn_norm.attn.Wqkv.", ".self_attn.qkv_proj."), (".norm_attn_norm.attn.out_proj.", ".self_attn.o_proj."),
                (".norm_attn_norm.norm_1.", ".input_layernorm."), (".norm_attn_norm.norm_2.", ".post_attention_layernorm."),
                (".ffn.router.layer.", ".mlp.gate."), (".ffn.experts.mlp.", ".mlp.experts."))

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("transformer."):
            name = "model." + name[len("transformer."):]
        name = name.replace("model.blocks.", "model.layers.").replace("model.wte.", "model.embed_tokens.") \
                   .replace("model.norm_f.", "model.norm.")
        for old, new in self._RENAMES:
            name = name.replace(old, new)
        return super().map_weight_name(name)

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        from .utils import materialize
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        ne, loaded = int(self.cfg["num_local_experts"]), []

        def others():
            for name, w in weights:
                mapped = self.map_weight_name(name)
                which = mapped.rsplit(".", 1)[-1] if mapped and ".mlp.experts." in mapped else None
                if which not in ("w1", "v1", "w2"):
                    yield name, w
                    continue
                if not self._owns_layer(mapped):
                    continue
                prefix = mapped[:mapped.index(".mlp.")] + ".mlp"
                t = materialize(w)
                t = t.view(ne, t.shape[0] // ne, t.shape[1])
                for e in range(ne):
                    if which == "w2":
                        p = params[prefix + ".w2"]
                        p.weight_loader(p, t[e].t().contiguous(), (e, 0))
                    else:
                        p = params[prefix + ".w13"]
                        p.weight_loader(p, t[e], (e, 0 if which == "w1" else 1))
                loaded.append(mapped)
        return loaded + LlamaForCausalLM.load_weights(self, others())
