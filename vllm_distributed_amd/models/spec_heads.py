"""Draft heads for speculative decoding that read the TARGET model's hidden states.

The first two heads here predict token ``i + 2`` from the pair (embedding of token ``i + 1``, the target's final hidden
state at position ``i``): one cheap pass over a step's tokens yields a first draft per request, and feeding a
head its own output as the next "hidden state" yields further ones.

* ``EagleLlamaForCausalLM`` - EAGLE (Li et al., 2024): ``fc([embed ; hidden])`` followed by Llama decoder layers,
  the first of them without an input norm, and no final norm; logits through the target's LM head.  Checkpoints
  (``yuhuili/EAGLE-*``) carry ``fc.*`` and ``layers.0.*`` and usually their own copy of ``embed_tokens``; names
  are accepted with and without a ``model.`` prefix.  What a checkpoint leaves out (embedding, LM head) is shared
  with the target.
* ``DeepSeekMTP`` - the multi-token-prediction module DeepSeek-V3 checkpoints carry as layer
  ``num_hidden_layers``: ``eh_proj([enorm(embed) ; hnorm(hidden)])`` -> one full decoder layer ->
  ``shared_head.norm`` -> LM head.  Embedding and head are the target's.

* ``MedusaModel`` - Medusa (Cai et al., 2024): ``num_heads`` independent heads over the target's LAST hidden state,
  head ``i`` = a few residual blocks ``x + SiLU(W x + b)`` and an LM head of its own, predicting token ``i + 2``.
  No attention, no KV cache, no pass over the step's tokens: all drafts of a request come from one hidden state.
  Names as vLLM writes them (``blocks.<i>.layers.<j>.{weight,bias}``, ``lm_heads.<i>.weight``) and as the Medusa
  repository does (``medusa_head.<i>.<j>.linear.*``, ``medusa_head.<i>.<n>.weight`` for the LM head).

vLLM serves them through ``--speculative-config '{"method": "eagle" | "mtp" | "medusa", ...}'``; the reference
framework reaches them through the vLLM app it launches (/root/reference/src/launch.py:429).  The reference ships no
fixtures for any: the tests pin these heads against plain PyTorch re-statements of the same formulas.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import re

import torch
import torch.nn as nn

from ..config import VllmConfig
from .deepseek import DeepseekV3ForCausalLM
from .layers.attention import RMSNorm
from .layers.linear import ParallelLMHead, ReplicatedLinear
from .llama import LlamaForCausalLM
from .utils import materialize


class EagleLlamaForCausalLM(LlamaForCausalLM):
    # parameters a head checkpoint may leave out: the proposer binds the target's modules instead
    optional_parameters = ("embed_tokens.weight", "lm_head.weight")

    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
        cfg, dtype = self.cfg, vllm_config.model_config.torch_dtype
        h = cfg["hidden_size"]
        self.fc = ReplicatedLinear(2 * h, h, bool(cfg.get("bias", True)), dtype, quant=False)
        first = next(iter(self.layers.values()))
        first.input_layernorm = None            # EAGLE's first layer attends over the fc output as it is
        self.norm = None
        self.missing_optional: List[str] = []

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, hidden_in: torch.Tensor) -> torch.Tensor:
        """The head's feature per token, ``[T, H]``: both the input of ``compute_logits`` and the
        ``hidden_in`` of the next draft step."""
        x = self.fc(torch.cat([self.embed_tokens(input_ids), hidden_in.to(self.fc.weight.dtype)], dim=-1))
        hidden, residual = x, None
        for layer in self.layers.values():
            if layer.input_layernorm is None:
                residual = hidden
                attn = layer.self_attn(positions, hidden)
                hidden, residual = layer.post_attention_layernorm(attn, residual)
                hidden = layer.mlp(hidden)
            else:
                hidden, residual = layer(positions, hidden, residual)
                if hidden is None:
                    hidden = torch.zeros_like(residual)
                elif not isinstance(hidden, torch.Tensor):
                    hidden = hidden.materialize()
        return hidden + residual

    def fused_tp_mode(self, num_tokens: int) -> Optional[bool]:
        return None

    def map_weight_name(self, name: str) -> Optional[str]:
        name = super().map_weight_name(name)
        if name is None or name.endswith("layers.0.input_layernorm.weight") or name == "norm.weight":
            return None
        return name


class DeepSeekMTP(DeepseekV3ForCausalLM):
    """One multi-token-prediction module of a DeepSeek-V3 checkpoint: the tensors under
    ``model.layers.<num_hidden_layers>.``.  Built as a one-layer instance of the target family, so the layer's
    tensors go through the family's own loader (expert stacking, rope row order)."""
    optional_parameters = ("embed_tokens.weight", "lm_head.weight")

    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
        cfg, dtype = self.cfg, vllm_config.model_config.torch_dtype
        h, eps = cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-6)
        self.enorm = RMSNorm(h, eps, dtype)
        self.hnorm = RMSNorm(h, eps, dtype)
        self.eh_proj = ReplicatedLinear(2 * h, h, False, dtype, quant=False)
        self.missing_optional: List[str] = []

    def layer_range(self, cfg: dict, vllm_config: VllmConfig) -> Tuple[int, int]:
        n = int(cfg["num_hidden_layers"])
        return n, n + 1

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, hidden_in: torch.Tensor) -> torch.Tensor:
        e = self.enorm(self.embed_tokens(input_ids))
        x = self.eh_proj(torch.cat([e, self.hnorm(hidden_in.to(e.dtype))], dim=-1))
        (layer,) = self.layers.values()
        hidden, residual = layer(positions, x, None)
        if hidden is None:
            return residual
        if not isinstance(hidden, torch.Tensor):
            hidden = hidden.materialize()
        return hidden + residual

    def fused_tp_mode(self, num_tokens: int) -> Optional[bool]:
        return None

    def compute_logits(self, feature: torch.Tensor) -> torch.Tensor:
        return super().compute_logits(self.norm(feature))          # norm = the module's shared_head.norm

    def map_weight_name(self, name: str) -> Optional[str]:
        prefix = f"model.layers.{self.start_layer}."
        if not name.startswith(prefix) or "rotary_emb.inv_freq" in name:
            return None
        rest = name[len(prefix):]
        if rest.startswith(("embed_tokens.", "shared_head.head.")):
            return None                         # copies of the target's embedding and LM head
        if rest.startswith("shared_head.norm."):
            return "norm." + rest[len("shared_head.norm."):]
        if rest.startswith(("enorm.", "hnorm.", "eh_proj.")):
            return rest
        return f"layers.{self.start_layer}.{rest}"


class MedusaModel(nn.Module):
    optional_parameters = ()
    packed_quantization = False

    def __init__(self, vllm_config: VllmConfig):
        super().__init__()
        mc = vllm_config.model_config
        cfg, dtype = mc.hf_config, mc.torch_dtype
        self.cfg = cfg
        h = int(cfg["hidden_size"])
        self.num_heads = int(cfg.get("num_heads") or cfg.get("medusa_num_heads") or 5)
        self.num_blocks = int(cfg.get("num_hidden_layers") or cfg.get("medusa_num_layers") or 1)
        self.blocks = nn.ModuleList(nn.ModuleList(ReplicatedLinear(h, h, True, dtype, quant=False)
                                                  for _ in range(self.num_blocks)) for _ in range(self.num_heads))
        self.lm_heads = nn.ModuleList(ParallelLMHead(int(cfg["vocab_size"]), h, dtype) for _ in range(self.num_heads))
        self.start_layer = self.end_layer = 0
        self.missing_optional: List[str] = []

    def rebind_buffers(self) -> None:
        pass

    def attention_layers(self) -> list:
        return []

    @torch.inference_mode()
    def forward(self, hidden: torch.Tensor, heads: int) -> torch.Tensor:
        """``hidden[R, H]`` (the target's last hidden state per request) -> ``[R, heads]`` greedy draft tokens."""
        out = []
        for i in range(min(heads, self.num_heads)):
            x = hidden
            for blk in self.blocks[i]:
                x = x + torch.nn.functional.silu(blk(x))
            out.append(self.lm_heads[i].greedy_tokens(x))
        return torch.stack(out, dim=1)

    # ------------------------------------------------------------- weights
    _ORIGINAL = re.compile(r"^medusa_head\.(\d+)\.(\d+)\.(?:linear\.)?(weight|bias)$")

    def _owns_layer(self, name: str) -> bool:
        return True

    def map_weight_name(self, name: str) -> Optional[str]:
        m = self._ORIGINAL.match(name)
        if m is not None:
            i, j, what = int(m.group(1)), int(m.group(2)), m.group(3)
            return f"lm_heads.{i}.{what}" if j >= self.num_blocks else f"blocks.{i}.{j}.{what}"
        name = name.replace(".layers.", ".") if name.startswith("blocks.") else name
        if name.startswith(("blocks.", "lm_heads.")):
            return name
        return None

    def load_weights(self, weights):
        params = dict(self.named_parameters())
        loaded = []
        for name, w in weights:
            mapped = self.map_weight_name(name)
            if mapped is None or mapped not in params:
                continue
            p = params[mapped]
            loader = getattr(p, "weight_loader", None)
            if loader is None:
                p.data.copy_(materialize(w))
            else:
                loader(p, w, None)
            loaded.append(mapped)
        return loaded
