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


class Eagle3LlamaForCausalLM(LlamaForCausalLM):
    """EAGLE-3 (Li et al., 2025): the head reads the target's residual stream at THREE depths (by default before
    layers 2, L / 2 and L - 3), folds them with ``fc`` (3 H_target -> H), and runs one decoder layer whose attention
    takes ``[input_layernorm(embed) ; hidden_norm(hidden)]`` (2 H wide); ``norm`` + an LM head over a reduced draft
    vocabulary give the logits, ``d2t[i]`` is the offset from draft id ``i`` to its target id.  Further drafts feed the
    head its own pre-norm output.  Checkpoint names (``yuhuili/EAGLE3-*``): ``fc.weight``, ``midlayer.*`` (the
    layer, with ``hidden_norm`` next to the usual two norms), ``norm.weight``, ``lm_head.weight``, ``d2t``, ``t2d``
    (unused here), optionally ``embed_tokens.weight`` (else the target's).  vLLM serves it through
    ``--speculative-config '{"method": "eagle3", ...}'``, the reference through the vLLM app it launches
    (/root/reference/src/launch.py:429); no fixture ships with either: pinned against a PyTorch re-statement."""
    optional_parameters = ("embed_tokens.weight",)

    def __init__(self, vllm_config: VllmConfig):
        mc = vllm_config.model_config
        cfg = mc.hf_config
        # (the base constructor builds no LM head for a tied model: this head's own one is over the draft vocabulary)
        mc.hf_config = dict(cfg, tie_word_embeddings=True)
        try:
            super().__init__(vllm_config)
        finally:
            mc.hf_config = cfg
        self.cfg = cfg
        dtype = mc.torch_dtype
        h = int(cfg["hidden_size"])
        self.target_hidden = int(cfg.get("target_hidden_size") or h)
        self.draft_vocab = int(cfg.get("draft_vocab_size") or cfg["vocab_size"])
        eps = cfg.get("rms_norm_eps", 1e-6)
        self.fc = ReplicatedLinear(3 * self.target_hidden, h, False, dtype, quant=False)
        (layer,) = self.layers.values()
        a = layer.self_attn
        from .layers.linear import QKVParallelLinear
        heads, kvh = int(cfg["num_attention_heads"]), int(cfg.get("num_key_value_heads") or cfg["num_attention_heads"])
        a.qkv_proj = QKVParallelLinear(2 * h, a.head_dim, heads, kvh, bool(cfg.get("attention_bias", False)), dtype)
        layer.hidden_norm = RMSNorm(h, eps, dtype)
        self.lm_head = ParallelLMHead(self.draft_vocab, h, dtype)
        self.tie = False
        # draft id -> target id offset; identity until a checkpoint brings the table
        self.register_buffer("d2t", torch.zeros(self.draft_vocab, dtype=torch.int64), persistent=False)
        self.missing_optional: List[str] = []

    # the target's layers whose INPUT (residual stream) the head reads, for a target of ``n`` layers
    def aux_layers(self, n: int) -> Tuple[int, int, int]:
        ids = (self.cfg.get("eagle_config") or {}).get("eagle_aux_hidden_state_layer_ids")
        return tuple(int(i) for i in ids) if ids else (2, n // 2, n - 3)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, hidden_in: torch.Tensor) -> torch.Tensor:
        """``hidden_in``: the target's three residual streams side by side ``[T, 3 H_target]`` (a step's tokens), or
        this head's previous output ``[T, H]``.  Returns the pre-norm feature: the input of ``compute_logits`` and
        the ``hidden_in`` of the next draft step."""
        hidden_in = hidden_in.to(self.fc.weight.dtype)
        if hidden_in.shape[-1] == 3 * self.target_hidden:
            hidden_in = self.fc(hidden_in)
        (layer,) = self.layers.values()
        embeds = layer.input_layernorm(self.embed_tokens(input_ids))
        x = torch.cat([embeds, layer.hidden_norm(hidden_in)], dim=-1)
        attn = layer.self_attn(positions, x)
        hidden, residual = layer.post_attention_layernorm(attn, hidden_in)
        hidden = layer.mlp(hidden)
        return hidden + residual

    def fused_tp_mode(self, num_tokens: int) -> Optional[bool]:
        return None

    def sequence_parallel_ok(self, num_tokens: int) -> bool:
        return False

    def compute_logits(self, feature: torch.Tensor) -> torch.Tensor:
        return self.lm_head.logits(self.norm(feature))

    def map_draft_ids(self, ids: torch.Tensor) -> torch.Tensor:
        """Greedy picks over the draft vocabulary -> ids of the target's vocabulary."""
        return (ids.long() + self.d2t[ids.long()]).to(ids.dtype)

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("model."):
            name = name[len("model."):]
        if name == "t2d":
            return None
        if name == "d2t":
            return name                          # (a buffer, not a parameter: taken by load_weights below)
        if name.startswith("midlayer."):
            name = f"layers.{self.start_layer}." + name[len("midlayer."):]
        return super().map_weight_name(name)

    def load_weights(self, weights):
        def rest():
            for name, w in weights:
                if name in ("d2t", "model.d2t"):
                    self.d2t.copy_(materialize(w).to(torch.int64))
                    continue
                yield name, w
        return super().load_weights(rest())


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
