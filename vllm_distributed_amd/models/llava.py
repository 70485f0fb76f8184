"""LLaVA-1.5-style vision-language models: a CLIP vision tower, a two-layer projector and a Llama-shaped decoder.

The decoder is this package's Llama implementation unchanged (every TP / PP / quantisation / graph path applies);
the only coupling is the embedding of a prefill step: rows whose token is an image placeholder are overwritten
with the projected patch features of that image (``set_mm_step``).  The tower is small next to the decoder
(0.3 B parameters for CLIP ViT-L/14-336) and runs once per image: it is replicated on every TP rank and uses
PyTorch's library GEMMs and fused attention; the decoder's hand-written kernels are where a request's time goes.

Checkpoint names, both generations of ``transformers``: ``language_model.model.*`` / ``language_model.lm_head.*`` /
``vision_tower.vision_model.*`` / ``multi_modal_projector.*``, and the same under ``model.`` with ``lm_head`` at
top level.  vLLM: ``vllm/model_executor/models/llava.py``, ``clip.py``.  Parity is pinned against
``transformers.LlavaForConditionalGeneration`` on a small random model (``tests/test_hf_parity.py``).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llama import LlamaForCausalLM
from .utils import materialize


def _act(name: str):
    if name == "quick_gelu":
        return lambda x: x * torch.sigmoid(1.702 * x)
    if name in ("gelu", "gelu_new", "gelu_pytorch_tanh"):
        return (lambda x: F.gelu(x, approximate="tanh")) if name != "gelu" else F.gelu
    raise NotImplementedError(f"vision activation {name!r}")


class ClipEncoderLayer(nn.Module):
    def __init__(self, h: int, inter: int, heads: int, eps: float, act: str, dtype: torch.dtype):
        super().__init__()
        self.heads = heads
        self.layer_norm1 = nn.LayerNorm(h, eps=eps, dtype=dtype)
        self.layer_norm2 = nn.LayerNorm(h, eps=eps, dtype=dtype)
        self.qkv = nn.Linear(h, 3 * h, dtype=dtype)                # q_proj | k_proj | v_proj, stacked at load
        self.out_proj = nn.Linear(h, h, dtype=dtype)
        self.fc1 = nn.Linear(h, inter, dtype=dtype)
        self.fc2 = nn.Linear(inter, h, dtype=dtype)
        self.act = _act(act)

    def forward(self, x: torch.Tensor) -> torch.Tensor:           # [B, S, H]
        b, s, h = x.shape
        q, k, v = self.qkv(self.layer_norm1(x)).view(b, s, 3, self.heads, h // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, s, h)
        x = x + self.out_proj(a)
        return x + self.fc2(self.act(self.fc1(self.layer_norm2(x))))


class ClipVisionTower(nn.Module):
    """CLIP's vision transformer up to the layer whose output the projector reads."""

    def __init__(self, v: dict, feature_layer: int, dtype: torch.dtype):
        super().__init__()
        h, self.patch, self.side = int(v["hidden_size"]), int(v["patch_size"]), int(v["image_size"])
        n_layers = int(v["num_hidden_layers"])
        # hidden_states[i] is the input of layer i: feature_layer -2 needs layers 0 .. n-2
        self.depth = feature_layer if feature_layer >= 0 else n_layers + 1 + feature_layer
        if not 0 <= self.depth <= n_layers:
            raise ValueError(f"vision_feature_layer {feature_layer} outside a tower of {n_layers} layers")
        eps = float(v.get("layer_norm_eps", 1e-5))
        self.class_embedding = nn.Parameter(torch.empty(h, dtype=dtype), requires_grad=False)
        self.patch_embedding = nn.Linear(3 * self.patch * self.patch, h, bias=False, dtype=dtype)   # the conv, unfolded
        self.position_embedding = nn.Parameter(torch.empty((self.side // self.patch) ** 2 + 1, h, dtype=dtype),
                                               requires_grad=False)
        self.pre_layrnorm = nn.LayerNorm(h, eps=eps, dtype=dtype)
        self.layers = nn.ModuleList(ClipEncoderLayer(h, int(v["intermediate_size"]), int(v["num_attention_heads"]), eps,
                                                     v.get("hidden_act", "quick_gelu"), dtype) for _ in range(self.depth))

    def forward(self, pixels: torch.Tensor) -> torch.Tensor:      # [B, 3, S, S] -> [B, 1 + patches, H]
        b, p, g = pixels.shape[0], self.patch, self.side // self.patch
        x = pixels.view(b, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b, g * g, 3 * p * p)
        x = self.patch_embedding(x)
        x = torch.cat([self.class_embedding.expand(b, 1, -1), x], dim=1) + self.position_embedding
        x = self.pre_layrnorm(x)
        for layer in self.layers:
            x = layer(x)
        return x


class LlavaForConditionalGeneration(LlamaForCausalLM):
    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
        cfg, dtype = self.cfg, vllm_config.model_config.torch_dtype
        v = cfg["vision_config"]
        self.image_token = int(cfg.get("image_token_index", cfg.get("image_token_id", 32000)))
        self.select_full = cfg.get("vision_feature_select_strategy", "default") == "full"
        layer = cfg.get("vision_feature_layer", -2)
        if not isinstance(layer, int):
            raise NotImplementedError("several tapped vision layers")
        self._mm: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        if self.is_first:                       # the stage that embeds owns the tower
            self.vision_tower = ClipVisionTower(v, layer, dtype)
            bias = bool(cfg.get("multimodal_projector_bias", True))
            self.multi_modal_projector = nn.ModuleDict({
                "linear_1": nn.Linear(int(v["hidden_size"]), cfg["hidden_size"], bias=bias, dtype=dtype),
                "linear_2": nn.Linear(cfg["hidden_size"], cfg["hidden_size"], bias=bias, dtype=dtype)})
            self.projector_act = _act(cfg.get("projector_hidden_act", "gelu"))

    # ------------------------------------------------------------- images
    @torch.inference_mode()
    def encode_images(self, pixels: torch.Tensor) -> torch.Tensor:
        """``[B, 3, S, S]`` normalised pixels -> ``[B, tokens_per_image, hidden]`` decoder-space embeddings."""
        w = self.multi_modal_projector["linear_1"].weight
        feats = self.vision_tower(pixels.to(device=w.device, dtype=w.dtype))
        if not self.select_full:
            feats = feats[:, 1:]
        p = self.multi_modal_projector
        return p["linear_2"](self.projector_act(p["linear_1"](feats)))

    def set_mm_step(self, rows: Optional[torch.Tensor], embeds: Optional[torch.Tensor] = None) -> None:
        """Token rows of the coming forward whose embedding is ``embeds`` (None: a text-only step)."""
        self._mm = None if rows is None else (rows, embeds)

    def embed_input(self, input_ids: torch.Tensor) -> torch.Tensor:
        x = self.embed_tokens(input_ids)
        if self._mm is not None:
            rows, embeds = self._mm
            x.index_copy_(0, rows, embeds.to(x.dtype))
        return x

    def fused_tp_mode(self, num_tokens: int) -> Optional[bool]:
        # the fused forms embed as partial sums left in the peer-mapped area: a step with image rows
        # takes the plain layer loop, whose embedding is a finished tensor
        return None if self._mm is not None else super().fused_tp_mode(num_tokens)

    def sequence_parallel_ok(self, num_tokens: int) -> bool:
        return self._mm is None and super().sequence_parallel_ok(num_tokens)

    # ------------------------------------------------------------- weights
    _QKV = {"q_proj": 0, "k_proj": 1, "v_proj": 2}

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("model.") and not name.startswith("model.layers.") and (
                ".language_model." in "." + name or ".vision_tower." in "." + name or ".multi_modal_projector." in "." + name):
            name = name[len("model."):]                       # newer layout: everything under "model."
        if name.startswith("language_model."):
            rest = name[len("language_model."):]
            return super().map_weight_name(rest if rest.startswith(("model.", "lm_head.")) else "model." + rest)
        if name.startswith("vision_tower."):
            rest = name[len("vision_tower."):]
            rest = rest[len("vision_model."):] if rest.startswith("vision_model.") else rest
            if rest.startswith("post_layernorm.") or rest == "embeddings.position_ids":
                return None
            return "vision_tower." + rest
        if name.startswith("multi_modal_projector."):
            return name
        return super().map_weight_name(name)

    def _owns_layer(self, name: str) -> bool:
        if name.startswith(("vision_tower.", "multi_modal_projector.")):
            return self.is_first
        return super()._owns_layer(name)

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        own: Dict[str, nn.Parameter] = dict(self.named_parameters())
        loaded: List[str] = []

        def text_only():
            for name, w in weights:
                mapped = self.map_weight_name(name)
                if mapped is None or not mapped.startswith(("vision_tower.", "multi_modal_projector.")):
                    yield name, w
                    continue
                if not self.is_first:
                    continue
                if self._load_vision(own, mapped, materialize(w)):
                    loaded.append(mapped)

        loaded += super().load_weights(text_only())
        return loaded

    def _load_vision(self, own: Dict[str, nn.Parameter], name: str, w: torch.Tensor) -> bool:
        if name.startswith("multi_modal_projector."):
            own[name].data.copy_(w)
            return True
        rest = name[len("vision_tower."):]
        if rest == "embeddings.class_embedding":
            own["vision_tower.class_embedding"].data.copy_(w)
        elif rest == "embeddings.patch_embedding.weight":      # conv [H, 3, p, p] -> linear over (c, i, j)
            own["vision_tower.patch_embedding.weight"].data.copy_(w.reshape(w.shape[0], -1))
        elif rest == "embeddings.position_embedding.weight":
            own["vision_tower.position_embedding"].data.copy_(w)
        elif rest.startswith("pre_layrnorm."):
            own["vision_tower." + rest].data.copy_(w)
        elif rest.startswith("encoder.layers."):
            parts = rest.split(".")
            idx, tail = int(parts[2]), ".".join(parts[3:])
            if idx >= self.vision_tower.depth:
                return False                                   # layers behind the tapped one are never run
            base = f"vision_tower.layers.{idx}."
            if tail.startswith("self_attn.") and parts[4] in self._QKV:
                p = own[base + "qkv." + parts[5]]
                h = p.shape[0] // 3
                p.data[self._QKV[parts[4]] * h:(self._QKV[parts[4]] + 1) * h].copy_(w)
            elif tail.startswith("self_attn.out_proj."):
                own[base + "out_proj." + parts[5]].data.copy_(w)
            elif tail.startswith("mlp."):
                own[base + tail[len("mlp."):]].data.copy_(w)
            else:
                own[base + tail].data.copy_(w)
        else:
            return False
        return True


class LlavaNextForConditionalGeneration(LlavaForConditionalGeneration):
    """LLaVA-NeXT (LLaVA-1.6): any-resolution input.  An image arrives as a base view plus the tiles of the best of
    ``image_grid_pinpoints`` (``multimodal.AnyResImageProcessor``); every view goes through the tower and the
    projector, the tiles' patch grids are stitched into one, the padding the image got on its way to that resolution
    is cut off again, the learned ``image_newline`` vector closes every row, and the base view's patches go in front.
    The number of embeddings therefore depends on the image's aspect (``multimodal.anyres_tokens``).  The reference
    serves it through the vLLM app it launches (/root/reference/src/launch.py:429); pinned against transformers in
    ``tests/test_hf_parity.py``."""

    def __init__(self, vllm_config: VllmConfig):
        super().__init__(vllm_config)
        if self.select_full:
            raise NotImplementedError("LLaVA-NeXT with vision_feature_select_strategy=full")
        if self.is_first:
            dtype = vllm_config.model_config.torch_dtype
            self.image_newline = nn.Parameter(torch.empty(self.cfg["hidden_size"], dtype=dtype), requires_grad=False)

    @torch.inference_mode()
    def encode_anyres_image(self, pixels: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
        """``[1 + tiles, 3, S, S]`` views of one image of ``size`` (height, width) -> ``[tokens, hidden]``."""
        from ..multimodal import anyres_unpadded_grid, select_best_resolution
        feats = self.encode_images(pixels)                      # [1 + tiles, per * per, H]
        v = self.cfg["vision_config"]
        side = int(v["image_size"])
        per = side // int(v["patch_size"])
        bh, bw = select_best_resolution(size, self.cfg["image_grid_pinpoints"])
        gh, gw = bh // side, bw // side
        if feats.shape[0] != 1 + gh * gw or feats.shape[1] != per * per:
            raise RuntimeError(f"{feats.shape[0]} views of {feats.shape[1]} patches for a {gh} x {gw} tile grid")
        h = feats.shape[-1]
        grid = feats[1:].view(gh, gw, per, per, h).permute(0, 2, 1, 3, 4).reshape(gh * per, gw * per, h)
        rows, cols = anyres_unpadded_grid(size, (gh * per, gw * per))
        top, left = (gh * per - rows) // 2, (gw * per - cols) // 2
        grid = grid[top:top + rows, left:left + cols]
        newline = self.image_newline.to(grid.dtype).view(1, 1, h).expand(rows, 1, h)
        return torch.cat([feats[0], torch.cat([grid, newline], dim=1).reshape(rows * (cols + 1), h)], dim=0)

    def map_weight_name(self, name: str) -> Optional[str]:
        if name in ("image_newline", "model.image_newline"):
            return "image_newline"
        return super().map_weight_name(name)
