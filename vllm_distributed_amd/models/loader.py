"""Model construction and weight loading.

``load_model`` builds the architecture for this rank's (TP, PP) coordinates
directly on the target device and fills it from one of:

* safetensors shards in a local directory / the HF hub cache (each rank copies
  only the slices it owns; reached through ``collective_rpc("load_model")`` in
  the reference, ``src/launch.py:292``);
* deterministic random weights (``load_format=dummy``): every *checkpoint*
  tensor is generated whole from a seed derived from its name and then sharded
  by the same ``weight_loader`` path, so TP/PP shards of a dummy model are
  exactly the shards of one fixed full model;
* AWQ 4-bit checkpoints (the reference's documented model is AWQ,
  ``.env.server:11``): ``qweight/qzeros/scales`` triples stay packed for the
  W4A16 kernels (``quantization=awq``) or are dequantised to the compute dtype
  at load time (``awq-dequant``);
* NVFP4 checkpoints (``Dockerfile:5-6``): ``weight_packed / weight_scale /
  weight_global_scale`` stay packed for the fp4 weight-only kernels.
"""
from __future__ import annotations

import glob
import hashlib
import json
import os
import re
from typing import Callable, Dict, Iterable, Iterator, Optional, Tuple

import torch

from ..config import ModelConfig, VllmConfig, resolve_model_path
from ..logger import init_logger
from .registry import get_model_class
from .utils import LazyTensor, materialize

logger = init_logger(__name__)


# ------------------------------------------------------------ dummy weights
def _seed_for(name: str, base_seed: int) -> int:
    h = hashlib.blake2b(f"{base_seed}:{name}".encode(), digest_size=8).digest()
    return int.from_bytes(h, "little") & ((1 << 62) - 1)


def hf_weight_shapes(mc: ModelConfig) -> Dict[str, Tuple[int, ...]]:
    """Checkpoint tensor names and shapes for the supported architectures."""
    return _weight_shapes(mc.hf_config, mc.architecture)


def _weight_shapes(c: dict, arch: str) -> Dict[str, Tuple[int, ...]]:
    if arch == "MedusaModel":
        h = c["hidden_size"]
        out = {}
        for i in range(int(c.get("num_heads") or 5)):
            for j in range(int(c.get("num_hidden_layers") or 1)):
                out[f"blocks.{i}.layers.{j}.weight"], out[f"blocks.{i}.layers.{j}.bias"] = (h, h), (h,)
            out[f"lm_heads.{i}.weight"] = (c["vocab_size"], h)
        return out
    if arch in ("Eagle3LlamaForCausalLM", "LlamaForCausalLMEagle3"):
        # the names of the published EAGLE-3 heads: one decoder layer under "midlayer.", no embedding of its own
        h, th = int(c["hidden_size"]), int(c.get("target_hidden_size") or c["hidden_size"])
        heads, kvh = int(c["num_attention_heads"]), int(c.get("num_key_value_heads") or c["num_attention_heads"])
        hd, inter = int(c.get("head_dim") or h // heads), int(c["intermediate_size"])
        dv = int(c.get("draft_vocab_size") or c["vocab_size"])
        out = {"fc.weight": (h, 3 * th), "norm.weight": (h,), "lm_head.weight": (dv, h), "d2t": (dv,)}
        m = "midlayer."
        out[m + "self_attn.q_proj.weight"] = (heads * hd, 2 * h)
        out[m + "self_attn.k_proj.weight"] = out[m + "self_attn.v_proj.weight"] = (kvh * hd, 2 * h)
        out[m + "self_attn.o_proj.weight"] = (h, heads * hd)
        out[m + "mlp.gate_proj.weight"] = out[m + "mlp.up_proj.weight"] = (inter, h)
        out[m + "mlp.down_proj.weight"] = (h, inter)
        for n in ("input_layernorm", "hidden_norm", "post_attention_layernorm"):
            out[m + n + ".weight"] = (h,)
        return out
    if arch == "LlavaNextForConditionalGeneration":
        out = _weight_shapes(c, "LlavaForConditionalGeneration")
        out["image_newline"] = (c["hidden_size"],)
        return out
    if arch == "LlavaForConditionalGeneration":
        # the decoder's tensors under "language_model.", a CLIP vision tower and the two-layer projector
        text_arch = {"mistral": "MistralForCausalLM", "qwen2": "Qwen2ForCausalLM"}.get(c.get("model_type"), "LlamaForCausalLM")
        out = {"language_model." + n: s for n, s in _weight_shapes(c, text_arch).items()}
        v = c["vision_config"]
        vh, vi, ps = int(v["hidden_size"]), int(v["intermediate_size"]), int(v["patch_size"])
        p = "vision_tower.vision_model."
        out[p + "embeddings.class_embedding"] = (vh,)
        out[p + "embeddings.patch_embedding.weight"] = (vh, 3, ps, ps)
        out[p + "embeddings.position_embedding.weight"] = ((int(v["image_size"]) // ps) ** 2 + 1, vh)
        for n in ("pre_layrnorm", "post_layernorm"):
            out[p + n + ".weight"] = out[p + n + ".bias"] = (vh,)
        for i in range(int(v["num_hidden_layers"])):
            q = f"{p}encoder.layers.{i}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                out[q + f"self_attn.{n}.weight"], out[q + f"self_attn.{n}.bias"] = (vh, vh), (vh,)
            for n in ("layer_norm1", "layer_norm2"):
                out[q + n + ".weight"] = out[q + n + ".bias"] = (vh,)
            out[q + "mlp.fc1.weight"], out[q + "mlp.fc1.bias"] = (vi, vh), (vi,)
            out[q + "mlp.fc2.weight"], out[q + "mlp.fc2.bias"] = (vh, vi), (vh,)
        th = c["hidden_size"]
        out["multi_modal_projector.linear_1.weight"], out["multi_modal_projector.linear_1.bias"] = (th, vh), (th,)
        out["multi_modal_projector.linear_2.weight"], out["multi_modal_projector.linear_2.bias"] = (th, th), (th,)
        return out
    h = c["hidden_size"]
    L = c.get("num_hidden_layers", 0)
    shapes: Dict[str, Tuple[int, ...]] = {}
    if arch == "OPTForCausalLM":
        proj = c.get("word_embed_proj_dim", h)
        ffn = c["ffn_dim"]
        p = "model.decoder."
        shapes[p + "embed_tokens.weight"] = (c["vocab_size"], proj)
        shapes[p + "embed_positions.weight"] = (c["max_position_embeddings"] + 2, h)
        if proj != h:
            shapes[p + "project_in.weight"] = (h, proj)
            shapes[p + "project_out.weight"] = (proj, h)
        if c.get("do_layer_norm_before", True):
            shapes[p + "final_layer_norm.weight"] = (h,)
            shapes[p + "final_layer_norm.bias"] = (h,)
        for i in range(L):
            q = f"{p}layers.{i}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                shapes[q + f"self_attn.{n}.weight"] = (h, h)
                shapes[q + f"self_attn.{n}.bias"] = (h,)
            shapes[q + "fc1.weight"], shapes[q + "fc1.bias"] = (ffn, h), (ffn,)
            shapes[q + "fc2.weight"], shapes[q + "fc2.bias"] = (h, ffn), (h,)
            for n in ("self_attn_layer_norm", "final_layer_norm"):
                shapes[q + n + ".weight"] = (h,)
                shapes[q + n + ".bias"] = (h,)
        return shapes
    if arch == "GPT2LMHeadModel":
        inner = c.get("n_inner") or 4 * h
        shapes["transformer.wte.weight"] = (c["vocab_size"], h)
        shapes["transformer.wpe.weight"] = (c.get("n_positions", c.get("max_position_embeddings", 1024)), h)
        shapes["transformer.ln_f.weight"] = shapes["transformer.ln_f.bias"] = (h,)
        for i in range(L):
            q = f"transformer.h.{i}."
            for n in ("ln_1", "ln_2"):
                shapes[q + n + ".weight"] = shapes[q + n + ".bias"] = (h,)
            # Conv1D weights are stored [in, out]
            shapes[q + "attn.c_attn.weight"], shapes[q + "attn.c_attn.bias"] = (h, 3 * h), (3 * h,)
            shapes[q + "attn.c_proj.weight"], shapes[q + "attn.c_proj.bias"] = (h, h), (h,)
            shapes[q + "mlp.c_fc.weight"], shapes[q + "mlp.c_fc.bias"] = (h, inner), (inner,)
            shapes[q + "mlp.c_proj.weight"], shapes[q + "mlp.c_proj.bias"] = (inner, h), (h,)
        return shapes
    if arch == "GPTBigCodeForCausalLM":
        inner = c.get("n_inner") or 4 * h
        hd = h // c["num_attention_heads"]
        shapes["transformer.wte.weight"] = (c["vocab_size"], h)
        shapes["transformer.wpe.weight"] = (c.get("n_positions", c.get("max_position_embeddings", 1024)), h)
        shapes["transformer.ln_f.weight"] = shapes["transformer.ln_f.bias"] = (h,)
        for i in range(L):
            q = f"transformer.h.{i}."
            for n in ("ln_1", "ln_2"):
                shapes[q + n + ".weight"] = shapes[q + n + ".bias"] = (h,)
            shapes[q + "attn.c_attn.weight"], shapes[q + "attn.c_attn.bias"] = (h + 2 * hd, h), (h + 2 * hd,)
            shapes[q + "attn.c_proj.weight"], shapes[q + "attn.c_proj.bias"] = (h, h), (h,)
            shapes[q + "mlp.c_fc.weight"], shapes[q + "mlp.c_fc.bias"] = (inner, h), (inner,)
            shapes[q + "mlp.c_proj.weight"], shapes[q + "mlp.c_proj.bias"] = (h, inner), (h,)
        return shapes
    if arch == "GPTJForCausalLM":
        inner = c.get("n_inner") or 4 * h
        shapes["transformer.wte.weight"] = (c["vocab_size"], h)
        shapes["transformer.ln_f.weight"] = shapes["transformer.ln_f.bias"] = (h,)
        shapes["lm_head.weight"], shapes["lm_head.bias"] = (c["vocab_size"], h), (c["vocab_size"],)
        for i in range(L):
            q = f"transformer.h.{i}."
            shapes[q + "ln_1.weight"] = shapes[q + "ln_1.bias"] = (h,)
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                shapes[q + f"attn.{n}.weight"] = (h, h)
            shapes[q + "mlp.fc_in.weight"], shapes[q + "mlp.fc_in.bias"] = (inner, h), (inner,)
            shapes[q + "mlp.fc_out.weight"], shapes[q + "mlp.fc_out.bias"] = (h, inner), (h,)
        return shapes
    if arch == "BloomForCausalLM":
        heads = c["num_attention_heads"]
        shapes["transformer.word_embeddings.weight"] = (c["vocab_size"], h)
        for n in ("word_embeddings_layernorm", "ln_f"):
            shapes[f"transformer.{n}.weight"] = shapes[f"transformer.{n}.bias"] = (h,)
        for i in range(L):
            q = f"transformer.h.{i}."
            for n in ("input_layernorm", "post_attention_layernorm"):
                shapes[q + n + ".weight"] = shapes[q + n + ".bias"] = (h,)
            for n, (o, k) in (("self_attention.query_key_value", (3 * h, h)), ("self_attention.dense", (h, h)),
                              ("mlp.dense_h_to_4h", (4 * h, h)), ("mlp.dense_4h_to_h", (h, 4 * h))):
                shapes[q + n + ".weight"], shapes[q + n + ".bias"] = (o, k), (o,)
        return shapes
    if arch in ("MptForCausalLM", "MPTForCausalLM"):
        inner = int(c.get("expansion_ratio", 4)) * h
        shapes["transformer.wte.weight"] = (c["vocab_size"], h)
        shapes["transformer.norm_f.weight"] = (h,)
        for i in range(L):
            q = f"transformer.blocks.{i}."
            shapes[q + "norm_1.weight"] = shapes[q + "norm_2.weight"] = (h,)
            shapes[q + "attn.Wqkv.weight"], shapes[q + "attn.out_proj.weight"] = (3 * h, h), (h, h)
            shapes[q + "ffn.up_proj.weight"], shapes[q + "ffn.down_proj.weight"] = (inner, h), (h, inner)
        return shapes
    if arch in ("FalconForCausalLM", "RWForCausalLM"):
        from .falcon import falcon_kv_heads
        heads, kvh = c["num_attention_heads"], falcon_kv_heads(c)
        hd, inner, bias = h // heads, c.get("ffn_hidden_size") or 4 * h, bool(c.get("bias", False))
        new = bool(c.get("new_decoder_architecture"))
        n_ln = c.get("num_ln_in_parallel_attn")
        if new and (2 if n_ln is None else int(n_ln)) == 2:
            lns = ("ln_attn", "ln_mlp")
        else:
            lns = ("input_layernorm",) if (new or c.get("parallel_attn", True)) else \
                ("input_layernorm", "post_attention_layernorm")
        shapes["transformer.word_embeddings.weight"] = (c["vocab_size"], h)
        shapes["transformer.ln_f.weight"] = shapes["transformer.ln_f.bias"] = (h,)
        if not c.get("tie_word_embeddings", True):
            shapes["lm_head.weight"] = (c["vocab_size"], h)
        for i in range(L):
            q = f"transformer.h.{i}."
            for n in lns:
                shapes[q + n + ".weight"] = shapes[q + n + ".bias"] = (h,)
            for n, (o, k) in (("self_attention.query_key_value", ((heads + 2 * kvh) * hd, h)),
                              ("self_attention.dense", (h, h)), ("mlp.dense_h_to_4h", (inner, h)),
                              ("mlp.dense_4h_to_h", (h, inner))):
                shapes[q + n + ".weight"] = (o, k)
                if bias:
                    shapes[q + n + ".bias"] = (o,)
        return shapes
    if arch == "GPTNeoXForCausalLM":
        inter = c["intermediate_size"]
        shapes["gpt_neox.embed_in.weight"] = (c["vocab_size"], h)
        shapes["gpt_neox.final_layer_norm.weight"] = shapes["gpt_neox.final_layer_norm.bias"] = (h,)
        if not c.get("tie_word_embeddings", False):
            shapes["embed_out.weight"] = (c["vocab_size"], h)
        for i in range(L):
            q = f"gpt_neox.layers.{i}."
            for n in ("input_layernorm", "post_attention_layernorm"):
                shapes[q + n + ".weight"] = shapes[q + n + ".bias"] = (h,)
            shapes[q + "attention.query_key_value.weight"] = (3 * h, h)
            shapes[q + "attention.query_key_value.bias"] = (3 * h,)
            shapes[q + "attention.dense.weight"], shapes[q + "attention.dense.bias"] = (h, h), (h,)
            shapes[q + "mlp.dense_h_to_4h.weight"], shapes[q + "mlp.dense_h_to_4h.bias"] = (inter, h), (inter,)
            shapes[q + "mlp.dense_4h_to_h.weight"], shapes[q + "mlp.dense_4h_to_h.bias"] = (h, inter), (h,)
        return shapes
    heads = c["num_attention_heads"]
    kvh = c.get("num_key_value_heads") or heads
    hd = c.get("head_dim") or h // heads
    inter = c.get("intermediate_size", 0)
    if arch == "PhiForCausalLM":
        shapes["model.embed_tokens.weight"] = (c["vocab_size"], h)
        shapes["model.final_layernorm.weight"] = shapes["model.final_layernorm.bias"] = (h,)
        shapes["lm_head.weight"], shapes["lm_head.bias"] = (c["vocab_size"], h), (c["vocab_size"],)
        for i in range(L):
            q = f"model.layers.{i}."
            shapes[q + "input_layernorm.weight"] = shapes[q + "input_layernorm.bias"] = (h,)
            for n, rows in (("q_proj", heads * hd), ("k_proj", kvh * hd), ("v_proj", kvh * hd), ("dense", h)):
                shapes[q + f"self_attn.{n}.weight"], shapes[q + f"self_attn.{n}.bias"] = (rows, h), (rows,)
            shapes[q + "mlp.fc1.weight"], shapes[q + "mlp.fc1.bias"] = (inter, h), (inter,)
            shapes[q + "mlp.fc2.weight"], shapes[q + "mlp.fc2.bias"] = (h, inter), (h,)
        return shapes
    if arch == "Starcoder2ForCausalLM":
        shapes["model.embed_tokens.weight"] = (c["vocab_size"], h)
        shapes["model.norm.weight"] = shapes["model.norm.bias"] = (h,)
        if not c.get("tie_word_embeddings", True):
            shapes["lm_head.weight"] = (c["vocab_size"], h)
        for i in range(L):
            q = f"model.layers.{i}."
            for n in ("input_layernorm", "post_attention_layernorm"):
                shapes[q + n + ".weight"] = shapes[q + n + ".bias"] = (h,)
            for n, rows in (("q_proj", heads * hd), ("k_proj", kvh * hd), ("v_proj", kvh * hd), ("o_proj", h)):
                shapes[q + f"self_attn.{n}.weight"] = (rows, h if n != "o_proj" else heads * hd)
                shapes[q + f"self_attn.{n}.bias"] = (rows,)
            shapes[q + "mlp.c_fc.weight"], shapes[q + "mlp.c_fc.bias"] = (inter, h), (inter,)
            shapes[q + "mlp.c_proj.weight"], shapes[q + "mlp.c_proj.bias"] = (h, inter), (h,)
        return shapes
    if arch in ("DeepseekV2ForCausalLM", "DeepseekV3ForCausalLM", "DeepSeekMTP"):
        nope, rope, vd, rank = c["qk_nope_head_dim"], c["qk_rope_head_dim"], c["v_head_dim"], c["kv_lora_rank"]
        ab = bool(c.get("attention_bias", False))
        ne = c.get("n_routed_experts") or c.get("num_experts") or c.get("num_local_experts") or 0
        mi = c.get("moe_intermediate_size", 0)
        shapes["model.embed_tokens.weight"] = (c["vocab_size"], h)
        shapes["model.norm.weight"] = (h,)
        if not c.get("tie_word_embeddings", False):
            shapes["lm_head.weight"] = (c["vocab_size"], h)

        def lin(name: str, out_f: int, in_f: int, bias: bool = False) -> None:
            shapes[name + ".weight"] = (out_f, in_f)
            if bias:
                shapes[name + ".bias"] = (out_f,)

        nextn = int(c.get("num_nextn_predict_layers") or 0) if arch in ("DeepseekV3ForCausalLM", "DeepSeekMTP") else 0
        for i in range(L + nextn):
            q = f"model.layers.{i}."
            a = q + "self_attn."
            if i >= L:            # a multi-token-prediction module: a decoder layer plus its own glue
                shapes[q + "enorm.weight"] = shapes[q + "hnorm.weight"] = shapes[q + "shared_head.norm.weight"] = (h,)
                shapes[q + "eh_proj.weight"] = (h, 2 * h)
            if c.get("q_lora_rank"):
                lin(a + "q_a_proj", c["q_lora_rank"], h, ab)
                shapes[a + "q_a_layernorm.weight"] = (c["q_lora_rank"],)
                lin(a + "q_b_proj", heads * (nope + rope), c["q_lora_rank"])
            else:
                lin(a + "q_proj", heads * (nope + rope), h)
            lin(a + "kv_a_proj_with_mqa", rank + rope, h, ab)
            shapes[a + "kv_a_layernorm.weight"] = (rank,)
            lin(a + "kv_b_proj", heads * (nope + vd), rank)
            lin(a + "o_proj", h, heads * vd, ab)
            shapes[q + "input_layernorm.weight"] = shapes[q + "post_attention_layernorm.weight"] = (h,)
            dense = not ne or i < int(c.get("first_k_dense_replace") or 0) or i % int(c.get("moe_layer_freq") or 1)
            if dense:
                for n, (o, k) in (("gate_proj", (inter, h)), ("up_proj", (inter, h)), ("down_proj", (h, inter))):
                    lin(q + "mlp." + n, o, k)
                continue
            shapes[q + "mlp.gate.weight"] = (ne, h)
            if c.get("model_type") == "deepseek_v3":
                shapes[q + "mlp.gate.e_score_correction_bias"] = (ne,)
            for e in range(ne):
                lin(q + f"mlp.experts.{e}.gate_proj", mi, h)
                lin(q + f"mlp.experts.{e}.up_proj", mi, h)
                lin(q + f"mlp.experts.{e}.down_proj", h, mi)
            ns = int(c.get("n_shared_experts") or 0)
            if ns:
                lin(q + "mlp.shared_experts.gate_proj", mi * ns, h)
                lin(q + "mlp.shared_experts.up_proj", mi * ns, h)
                lin(q + "mlp.shared_experts.down_proj", h, mi * ns)
        return shapes
    moe = arch in ("Qwen3MoeForCausalLM", "Qwen2MoeForCausalLM", "OlmoeForCausalLM")
    mixtral = arch in ("MixtralForCausalLM", "PhimoeForCausalLM")
    phi3 = arch == "Phi3ForCausalLM"
    mt = c.get("model_type")
    qkv_bias = (bool(c.get("attention_bias", False)) or mt in ("qwen2", "qwen2_moe")
                or (mt == "stablelm" and bool(c.get("use_qkv_bias", False))))
    o_bias = c.get("attention_out_bias")
    if o_bias is None:
        o_bias = bool(c.get("attention_bias", False)) and mt not in ("glm", "glm4")
    mlp_bias = bool(c.get("mlp_bias", False))
    norm_bias = mt in ("stablelm", "nemotron", "phimoe")    # LayerNorm families
    # which norms a layer has, by family
    norms = {"gemma2": 4, "gemma3_text": 4}.get(mt) and (
        "input_layernorm", "post_attention_layernorm", "pre_feedforward_layernorm", "post_feedforward_layernorm")
    norms = norms or {"glm4": ("input_layernorm", "post_self_attn_layernorm", "post_attention_layernorm",
                               "post_mlp_layernorm"),
                      "olmo2": ("post_attention_layernorm", "post_feedforward_layernorm"),
                      "olmo3": ("post_attention_layernorm", "post_feedforward_layernorm"),
                      "exaone4": ("post_attention_layernorm", "post_feedforward_layernorm"),
                      "cohere": ("input_layernorm",), "cohere2": ("input_layernorm",), "olmo": ()}.get(mt, ("input_layernorm", "post_attention_layernorm"))
    eagle = arch == "EagleLlamaForCausalLM"       # a draft head: no embedding / head / final norm of its own
    if eagle:
        norms = ("input_layernorm", "post_attention_layernorm")
        shapes["model.fc.weight"] = (h, 2 * h)
        if c.get("bias", True):
            shapes["model.fc.bias"] = (h,)
    else:
        shapes["model.embed_tokens.weight"] = (c["vocab_size"], h)
    if mt != "olmo" and not eagle:
        shapes["model.norm.weight"] = (h,)
    if norm_bias:
        shapes["model.norm.bias"] = (h,)
    if not c.get("tie_word_embeddings", False) and not eagle:
        shapes["lm_head.weight"] = (c["vocab_size"], h)
        if mt == "phimoe" and c.get("lm_head_bias"):
            shapes["lm_head.bias"] = (c["vocab_size"],)

    def linear(name: str, out_f: int, in_f: int, bias: bool) -> None:
        shapes[name + ".weight"] = (out_f, in_f)
        if bias:
            shapes[name + ".bias"] = (out_f,)

    for i in range(L):
        q = f"model.layers.{i}."
        if phi3:        # checkpoint stores [q | k | v] fused
            shapes[q + "self_attn.qkv_proj.weight"] = ((heads + 2 * kvh) * hd, h)
        else:
            linear(q + "self_attn.q_proj", heads * hd, h, qkv_bias)
            linear(q + "self_attn.k_proj", kvh * hd, h, qkv_bias)
            linear(q + "self_attn.v_proj", kvh * hd, h, qkv_bias)
        linear(q + "self_attn.o_proj", h, heads * hd, bool(o_bias))
        if mt in ("qwen3", "qwen3_moe", "gemma3_text", "exaone4", "dots1") or (mt == "glm4_moe" and c.get("use_qk_norm")):
            shapes[q + "self_attn.q_norm.weight"] = (hd,)
            shapes[q + "self_attn.k_norm.weight"] = (hd,)
        elif mt in ("olmo2", "olmo3", "olmoe"):
            shapes[q + "self_attn.q_norm.weight"] = (heads * hd,)
            shapes[q + "self_attn.k_norm.weight"] = (kvh * hd,)
        for n in norms:
            if eagle and i == 0 and n == "input_layernorm":
                continue
            shapes[q + n + ".weight"] = (h,)
            if norm_bias:
                shapes[q + n + ".bias"] = (h,)
        step = c.get("decoder_sparse_step", 1)
        sparse = moe and c.get("num_experts", 0) > 0 and i not in c.get("mlp_only_layers", []) \
            and (i + 1) % step == 0
        v3_routed = mt in ("glm4_moe", "dots1") and i >= int(c.get("first_k_dense_replace") or 0)
        if v3_routed:
            mi, ne, ns = c["moe_intermediate_size"], c["n_routed_experts"], int(c.get("n_shared_experts") or 0)
            shapes[q + "mlp.gate.weight"], shapes[q + "mlp.gate.e_score_correction_bias"] = (ne, h), (ne,)
            for e in range(ne):
                linear(q + f"mlp.experts.{e}.gate_proj", mi, h, False)
                linear(q + f"mlp.experts.{e}.up_proj", mi, h, False)
                linear(q + f"mlp.experts.{e}.down_proj", h, mi, False)
            for n, (o, k) in (("gate_proj", (mi * ns, h)), ("up_proj", (mi * ns, h)), ("down_proj", (h, mi * ns))):
                if ns:
                    linear(q + "mlp.shared_experts." + n, o, k, False)
        elif sparse:
            mi = c.get("moe_intermediate_size") or inter
            shapes[q + "mlp.gate.weight"] = (c["num_experts"], h)
            for e in range(c["num_experts"]):
                shapes[q + f"mlp.experts.{e}.gate_proj.weight"] = (mi, h)
                shapes[q + f"mlp.experts.{e}.up_proj.weight"] = (mi, h)
                shapes[q + f"mlp.experts.{e}.down_proj.weight"] = (h, mi)
            if arch == "Qwen2MoeForCausalLM":
                si = int(c.get("shared_expert_intermediate_size") or 0)
                shapes[q + "mlp.shared_expert_gate.weight"] = (1, h)
                if si > 0:
                    shapes[q + "mlp.shared_expert.gate_proj.weight"] = (si, h)
                    shapes[q + "mlp.shared_expert.up_proj.weight"] = (si, h)
                    shapes[q + "mlp.shared_expert.down_proj.weight"] = (h, si)
        elif mt == "gpt_oss":        # experts stacked, [in, out] with gate / up columns interleaved; biases everywhere
            ne = c["num_local_experts"]
            shapes[q + "self_attn.sinks"] = (heads,)
            shapes[q + "mlp.router.weight"], shapes[q + "mlp.router.bias"] = (ne, h), (ne,)
            shapes[q + "mlp.experts.gate_up_proj"], shapes[q + "mlp.experts.gate_up_proj_bias"] = (ne, h, 2 * inter), (ne, 2 * inter)
            shapes[q + "mlp.experts.down_proj"], shapes[q + "mlp.experts.down_proj_bias"] = (ne, inter, h), (ne, h)
        elif mixtral:
            shapes[q + "block_sparse_moe.gate.weight"] = (c["num_local_experts"], h)
            for e in range(c["num_local_experts"]):
                shapes[q + f"block_sparse_moe.experts.{e}.w1.weight"] = (inter, h)
                shapes[q + f"block_sparse_moe.experts.{e}.w3.weight"] = (inter, h)
                shapes[q + f"block_sparse_moe.experts.{e}.w2.weight"] = (h, inter)
        elif phi3 or mt in ("glm", "glm4"):      # [gate | up] fused
            shapes[q + "mlp.gate_up_proj.weight"] = (2 * inter, h)
            shapes[q + "mlp.down_proj.weight"] = (h, inter)
        else:
            if mt not in ("arcee", "nemotron"):  # (those: no gate)
                linear(q + "mlp.gate_proj", inter, h, mlp_bias)
            linear(q + "mlp.up_proj", inter, h, mlp_bias)
            linear(q + "mlp.down_proj", h, inter, mlp_bias)
    return shapes


_QUANTIZED_LINEAR = re.compile(
    r"\.(q_proj|k_proj|v_proj|qkv_proj|o_proj|gate_proj|up_proj|gate_up_proj|down_proj|w1|w2|w3)\.weight$")


def dummy_weights_iterator(mc: ModelConfig, device: torch.device,
                           want: Optional[Callable[[str], bool]] = None,
                           ) -> Iterator[Tuple[str, torch.Tensor]]:
    """Deterministic random checkpoint.  With an AWQ model config the decoder's
    linear layers come as random ``qweight/qzeros/scales`` triples instead of
    dense matrices (the same tensors an AWQ checkpoint would hold)."""
    dtype = mc.torch_dtype
    gen = torch.Generator(device=device)
    awq = mc.quantization in ("awq", "awq-dequant")
    nvfp4 = mc.quantization == "nvfp4"
    g = mc.quant_group_size
    for name, shape in hf_weight_shapes(mc).items():
        if want is not None and not want(name):
            continue
        gen.manual_seed(_seed_for(name, mc.seed))
        if awq and _QUANTIZED_LINEAR.search(name) and shape[1] % g == 0 and shape[0] % 8 == 0:
            out_f, in_f = shape
            base = name[:-len(".weight")]
            lo, hi = -(1 << 31), (1 << 31) - 1
            yield base + ".qweight", torch.randint(lo, hi, (in_f, out_f // 8), generator=gen, device=device,
                                                   dtype=torch.int64).to(torch.int32)
            yield base + ".qzeros", torch.randint(lo, hi, (in_f // g, out_f // 8), generator=gen, device=device,
                                                  dtype=torch.int64).to(torch.int32)
            # (q - z) has std ~6.5: keep activations near unit variance
            sc = (1.0 / in_f) ** 0.5 / 6.5
            yield base + ".scales", (sc * (0.5 + torch.rand((in_f // g, out_f), generator=gen, device=device,
                                                            dtype=torch.float32))).to(dtype)
            continue
        if nvfp4 and _QUANTIZED_LINEAR.search(name) and shape[1] % 16 == 0:
            out_f, in_f = shape
            base = name[:-len(".weight")]
            yield base + ".weight_packed", torch.randint(0, 256, (out_f, in_f // 2), generator=gen, device=device,
                                                         dtype=torch.int32).to(torch.uint8)
            bs = 32.0 + 64.0 * torch.rand((out_f, in_f // 16), generator=gen, device=device, dtype=torch.float32)
            yield base + ".weight_scale", bs.to(torch.float8_e4m3fn)
            # uniform E2M1 codes have std ~2.9 and the block scales average 64:
            # the divisor keeps activations near unit variance
            yield base + ".weight_global_scale", torch.full((1,), 64.0 * 2.93 * 1.05 * in_f ** 0.5,
                                                            dtype=torch.float32, device=device)
            continue
        if name == "d2t":
            # an EAGLE-3 draft vocabulary: a sorted random subset of the target's ids, as offsets from the draft ids
            ids = torch.randperm(int(mc.hf_config["vocab_size"]), generator=gen, device=device)[:shape[0]].sort().values
            yield name, (ids - torch.arange(shape[0], device=device)).to(torch.int64)
            continue
        if re.search(r"(layernorm|layer_norm|norm)\.weight$", name):
            w = 1.0 + 0.05 * (torch.rand(shape, generator=gen, device=device, dtype=torch.float32) - 0.5)
            w = w.to(dtype)
        elif name.endswith((".bias", "_proj_bias")):
            w = (0.02 * (torch.rand(shape, generator=gen, device=device, dtype=torch.float32) - 0.5)).to(dtype)
        elif name.endswith(".sinks"):
            w = (2.0 * (torch.rand(shape, generator=gen, device=device, dtype=torch.float32) - 0.5)).to(dtype)
        elif name.endswith((".experts.gate_up_proj", ".experts.down_proj")):        # stored [E, in, out]
            scale = (3.0 / shape[-2]) ** 0.5
            w = torch.empty(shape, device=device, dtype=dtype)
            w.uniform_(-scale, scale, generator=gen)
        else:
            fan_in = shape[-1]
            scale = (3.0 / fan_in) ** 0.5      # unit-variance-preserving uniform
            if "embed" in name:
                scale = 0.05
            w = torch.empty(shape, device=device, dtype=dtype)
            w.uniform_(-scale, scale, generator=gen)
        yield name, w


# ------------------------------------------------------------- safetensors
_AWQ_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]


def awq_dequantize(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                   dtype: torch.dtype) -> torch.Tensor:
    """AWQ GEMM-format 4-bit -> ``[out, in]`` dense.  ``qweight[in, out/8]`` int32
    packs 8 nibbles per word in the order 0,4,1,5,2,6,3,7; ``qzeros[in/g, out/8]``
    likewise; ``scales[in/g, out]``."""
    shifts = torch.tensor([4 * _AWQ_ORDER.index(i) for i in range(8)], dtype=torch.int32,
                          device=qweight.device)

    def unpack(x: torch.Tensor) -> torch.Tensor:
        u = (x.unsqueeze(-1) >> shifts) & 0xF           # [..., out/8, 8]
        return u.reshape(x.shape[0], -1)
    w = unpack(qweight).to(torch.float32)               # [in, out]
    z = unpack(qzeros).to(torch.float32)                # [in/g, out]
    g = w.shape[0] // scales.shape[0]
    z = z.repeat_interleave(g, dim=0)
    s = scales.to(torch.float32).repeat_interleave(g, dim=0)
    return ((w - z) * s).t().contiguous().to(dtype)


def awq_reorder_out_columns(t: torch.Tensor, perm: torch.Tensor, packed: bool) -> torch.Tensor:
    """Output channels of an AWQ tensor (the LAST dimension: ``qweight[in, out/8]``, ``qzeros[in/g, out/8]``
    packed eight nibbles a word, ``scales[in/g, out]`` plain) taken in the order ``perm``."""
    perm = perm.to(t.device)
    if not packed:
        return t.index_select(1, perm).contiguous()
    shifts = torch.tensor([4 * _AWQ_ORDER.index(i) for i in range(8)], dtype=torch.int32, device=t.device)
    u = ((t.unsqueeze(-1) >> shifts) & 0xF).reshape(t.shape[0], -1).index_select(1, perm)
    u = u.reshape(t.shape[0], -1, 8).to(torch.int64) << shifts.to(torch.int64)
    word = u.sum(-1)                                       # disjoint nibbles: a sum is an OR
    return torch.where(word >= (1 << 31), word - (1 << 32), word).to(torch.int32)


def _dequantize_stream(weights: Iterable[Tuple[str, torch.Tensor]], dtype: torch.dtype,
                       ) -> Iterator[Tuple[str, torch.Tensor]]:
    """Turn ``qweight/qzeros/scales`` triples of a weight stream into dense
    ``.weight`` tensors; everything else passes through."""
    pending: Dict[str, Dict[str, torch.Tensor]] = {}
    for name, w in weights:
        for kind in ("qweight", "qzeros", "scales"):
            if name.endswith("." + kind):
                base = name[:-len(kind) - 1]
                slot = pending.setdefault(base, {})
                slot[kind] = w
                if len(slot) == 3:
                    del pending[base]
                    yield base + ".weight", awq_dequantize(slot["qweight"], slot["qzeros"], slot["scales"], dtype)
                break
        else:
            yield name, w


def safetensors_weights_iterator(path: str, dtype: torch.dtype,
                                 want: Optional[Callable[[str], bool]] = None,
                                 keep_quantized: bool = False,
                                 ) -> Iterator[Tuple[str, torch.Tensor]]:
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    index = os.path.join(path, "model.safetensors.index.json")
    if os.path.isfile(index):
        with open(index) as f:
            wanted_files = sorted(set(json.load(f)["weight_map"].values()))
        files = [os.path.join(path, w) for w in wanted_files if os.path.isfile(os.path.join(path, w))]
    for fn in files:
        with safe_open(fn, framework="pt", device="cpu") as f:
            keys = list(f.keys())
            keyset = set(keys)
            awq_done = set()
            for k in keys:
                # FP8 checkpoints (one e4m3 byte per weight + fp32 scales: per 128 x 128 block as
                # ``weight_scale_inv``, or per tensor / per output channel as ``weight_scale``) are widened
                # here; ``--quantization fp8`` then re-quantises per channel for the W8A8 kernels
                if k.endswith((".weight_scale_inv", ".weight_scale", ".input_scale")) and \
                        k[:k.rindex(".")] + ".weight" in keyset and not keep_quantized:
                    continue
                if k.endswith(".weight") and not keep_quantized and \
                        (k + "_scale_inv" in keyset or k + "_scale" in keyset):
                    if want is not None and not want(k):
                        continue
                    yield k, fp8_checkpoint_dequantize(
                        f.get_tensor(k), f.get_tensor(k + "_scale_inv" if k + "_scale_inv" in keyset else k + "_scale"),
                        dtype)
                    continue
                if keep_quantized:
                    if want is None or want(k):
                        if k.endswith(".scales"):
                            yield k, f.get_tensor(k).to(dtype)
                        else:
                            yield k, LazyTensor(f, k)
                    continue
                if k.endswith((".qzeros", ".scales")):
                    continue
                if k.endswith(".qweight"):
                    base = k[:-len(".qweight")]
                    name = base + ".weight"
                    if want is not None and not want(name):
                        continue
                    yield name, awq_dequantize(f.get_tensor(k), f.get_tensor(base + ".qzeros"),
                                               f.get_tensor(base + ".scales"), dtype)
                    awq_done.add(base)
                    continue
                if want is not None and not want(k):
                    continue
                # not read yet: the weight loaders narrow it to this rank's shard first, and only
                # that hyper-rectangle is then read from the file
                yield k, LazyTensor(f, k)


def fp8_checkpoint_dequantize(w: torch.Tensor, scale: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``w [N, K]`` e4m3 times its scales: a scalar, one per output row (``[N]`` / ``[N, 1]``), or one per
    block of a regular grid (``[ceil(N / bn), ceil(K / bk)]``, block sizes inferred from the shapes:
    128 x 128 in the published checkpoints)."""
    wf = w.float()
    s = scale.float()
    if s.numel() == 1:
        return (wf * s.reshape(())).to(dtype)
    n, k = wf.shape
    if s.dim() <= 1 or s.shape[-1] == 1:
        if s.numel() != n:
            raise ValueError(f"fp8 weight [{n}, {k}] with {s.numel()} scales")
        return (wf * s.reshape(n, 1)).to(dtype)
    gn, gk = s.shape
    bn, bk = -(-n // gn), -(-k // gk)
    for b in (bn, bk):          # block sizes are powers of two in every published layout: round up to one
        if b & (b - 1):
            raise ValueError(f"fp8 weight [{n}, {k}] with a [{gn}, {gk}] scale grid: blocks of {bn} x {bk}?")
    full = s.repeat_interleave(bn, dim=0)[:n].repeat_interleave(bk, dim=1)[:, :k]
    return (wf * full).to(dtype)


def _gptq_as_awq(items: Iterable[Tuple[str, torch.Tensor]], mc: ModelConfig) -> Iterator[Tuple[str, torch.Tensor]]:
    """A 4-bit GPTQ checkpoint as the AWQ tensors the quantised linears load: same integers, scales and
    groups, repacked (``ops/quant.gptq_to_awq``).  Activation reordering (``desc_act`` with a
    non-sequential ``g_idx``) permutes the input rows per group and is refused."""
    from ..ops import quant
    q = mc.hf_config.get("quantization_config") or {}
    if int(q.get("bits", 4)) != 4:
        raise NotImplementedError(f"GPTQ with {q.get('bits')} bits (4-bit only)")
    g = int(q.get("group_size", 128))
    offset = 0 if str(q.get("checkpoint_format", "gptq")) == "gptq_v2" else 1
    for name, w in items:
        if name.endswith(".g_idx"):
            idx = materialize(w).long()
            group = g if g > 0 else idx.numel()
            if not torch.equal(idx, torch.arange(idx.numel()) // group):
                raise NotImplementedError(f"{name}: GPTQ activation reordering (desc_act) is not supported")
            continue
        if name.endswith((".qweight", ".qzeros")):
            w = quant.gptq_to_awq(name, materialize(w), offset)
        yield name, w


def _with_progress(items: Iterable, what: str, every: float = 20.0) -> Iterator:
    """Pass ``items`` through, logging a count every ``every`` seconds (large
    models take minutes to load; silence looks like a hang)."""
    import time
    t_last, n = time.monotonic(), 0
    for x in items:
        yield x
        n += 1
        now = time.monotonic()
        if now - t_last >= every:
            logger.info("loading: %d %s", n, what)
            t_last = now


# ------------------------------------------------------------------- entry
_POISONABLE = (torch.float32, torch.float16, torch.bfloat16)


def _poison(model: torch.nn.Module) -> None:
    for p in model.parameters():
        if p.dtype in _POISONABLE:
            p.data.fill_(float("nan"))


def _check_filled(model: torch.nn.Module, what: str) -> None:
    """Every element of every parameter must have come from the checkpoint.  A fused parameter can be
    filled in part (qkv_proj without k_proj, gate_up_proj without one half, an expert stack with
    experts absent from the file): any element still carrying the poison value counts."""
    missing, partial = [], []
    for n, p in model.named_parameters():
        if p.dtype not in _POISONABLE or not p.numel():
            continue
        nan = torch.isnan(p.data)
        if bool(nan.any().item()):
            (missing if bool(nan.all().item()) else partial).append(n)
    optional = getattr(model, "optional_parameters", ())
    if optional:
        # modules a draft head may share with its target instead of carrying a copy: reported, not refused
        model.missing_optional = [n for n in missing if n in optional]
        missing = [n for n in missing if n not in optional]
    if missing or partial:
        msg = []
        if missing:
            msg.append(f"{len(missing)} parameter(s) were not found in the checkpoint: "
                       + ", ".join(missing[:8]) + (" ..." if len(missing) > 8 else ""))
        if partial:
            msg.append(f"{len(partial)} fused parameter(s) were only partly filled (a shard of each is missing "
                       f"from the checkpoint): " + ", ".join(partial[:8]) + (" ..." if len(partial) > 8 else ""))
        raise RuntimeError(f"{what}: " + "; ".join(msg))


def load_model(vllm_config: VllmConfig, device: torch.device) -> torch.nn.Module:
    mc = vllm_config.model_config
    cls = get_model_class(mc.architecture)
    if mc.quantization in ("awq", "nvfp4") and not getattr(cls, "packed_quantization", False):
        # (those checkpoints still load with ``--quantization awq-dequant``: 16-bit weights, dense kernels)
        raise ValueError(f"{mc.architecture} has no packed 4-bit linear layers: quantization={mc.quantization!r} is "
                         f"supported for the Llama-shaped and expert-mixture families only")
    if mc.quantization == "fp8" and not getattr(cls, "packed_quantization", False):
        logger.warning("%s has no fp8 linear layers: weights are kept in %s", mc.architecture, mc.dtype)
    with torch.device(device):
        model = cls(vllm_config)
    model.rebind_buffers()
    model.eval()
    # parameters are allocated uninitialised; mark them, so that one the checkpoint never fills
    # (a bias the config promised, a norm with another name) is an error and not silent garbage
    _poison(model)

    def want(name: str) -> bool:
        mapped = model.map_weight_name(name)
        return mapped is not None and model._owns_layer(mapped)

    path = resolve_model_path(mc.model)
    if mc.load_format == "sharded_state":
        if path is None:
            raise FileNotFoundError(f"no sharded state for {mc.model!r}")
        from ..parallel.state import get_world_group
        n = load_sharded_state(model, path, get_world_group().rank, vllm_config)
        logger.info("loaded %d tensors of this rank's shard from %s", n, path)
        _check_filled(model, mc.model)
        return model
    if mc.load_format == "dummy" or path is None:
        if mc.load_format != "dummy":       # "auto" without weights was settled (or refused) by ModelConfig
            raise FileNotFoundError(f"no weights for {mc.model!r}")
        it: Iterable[Tuple[str, torch.Tensor]] = dummy_weights_iterator(mc, device, want)
        if mc.quantization == "awq-dequant":
            it = _dequantize_stream(it, mc.torch_dtype)
    else:
        it = safetensors_weights_iterator(path, mc.torch_dtype, want, keep_quantized=mc.quantization in ("awq", "nvfp4"))
        if (mc.hf_config.get("quantization_config") or {}).get("quant_method") == "gptq":
            it = _gptq_as_awq(it, mc)
    with torch.no_grad():
        loaded = model.load_weights(_with_progress(it, "checkpoint tensors read"))
        mods = [m for m in model.modules() if getattr(m, "finalize_weights", None) is not None]
        for m in _with_progress(mods, "modules finalised"):      # repack quantised shards into the kernel layout
            m.finalize_weights()
    _check_filled(model, mc.model)
    logger.info("loaded %d checkpoint tensors for %s (layers %d-%d, format=%s)", len(loaded),
                mc.model, getattr(model, "start_layer", 0), getattr(model, "end_layer", 0),
                mc.load_format if path is None else "safetensors")
    return model


def save_dummy_checkpoint(mc: ModelConfig, out_dir: str) -> None:
    """Write the deterministic dummy weights as a safetensors checkpoint (tests
    use it to check that the file loader and the dummy loader agree)."""
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    tensors = {n: w.contiguous() for n, w in dummy_weights_iterator(mc, torch.device("cpu"))}
    save_file(tensors, os.path.join(out_dir, "model.safetensors"))
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(mc.hf_config, f)


# ------------------------------------------------------------ sharded state
SHARDED_STATE_PATTERN = "model-rank-{rank}.safetensors"
SHARDED_STATE_META = "sharded_state.json"


def save_sharded_state(model: torch.nn.Module, path: str, rank: int, vllm_config: VllmConfig,
                       pattern: Optional[str] = None) -> str:
    """Write this rank's tensors exactly as it holds them (already cut to its TP shard and PP
    layer range, quantised layers already in kernel layout), so a later start with
    ``--load-format sharded_state`` reads one file per rank and repacks nothing."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    sd = {k: v.detach().to("cpu").contiguous() for k, v in model.state_dict().items()}
    fn = os.path.join(path, (pattern or SHARDED_STATE_PATTERN).format(rank=rank))
    save_file(sd, fn)
    if rank == 0:
        pc, mc = vllm_config.parallel_config, vllm_config.model_config
        with open(os.path.join(path, SHARDED_STATE_META), "w") as f:
            json.dump({"tensor_parallel_size": pc.tensor_parallel_size,
                       "pipeline_parallel_size": pc.pipeline_parallel_size,
                       "pp_layer_partition": pc.pp_layer_partition, "quantization": mc.quantization,
                       "dtype": mc.dtype, "pattern": pattern or SHARDED_STATE_PATTERN}, f)
        if not os.path.isfile(os.path.join(path, "config.json")):
            with open(os.path.join(path, "config.json"), "w") as f:
                json.dump(mc.hf_config, f)
    return fn


def load_sharded_state(model: torch.nn.Module, path: str, rank: int, vllm_config: VllmConfig) -> int:
    """Inverse of ``save_sharded_state``: the layout the files were written for (TP / PP
    degree, layer partition, quantisation) must be the layout of this start."""
    from safetensors import safe_open
    pattern = SHARDED_STATE_PATTERN
    meta_fn = os.path.join(path, SHARDED_STATE_META)
    if os.path.isfile(meta_fn):
        with open(meta_fn) as f:
            meta = json.load(f)
        pc, mc = vllm_config.parallel_config, vllm_config.model_config
        have = {"tensor_parallel_size": pc.tensor_parallel_size, "pipeline_parallel_size": pc.pipeline_parallel_size,
                "pp_layer_partition": pc.pp_layer_partition, "quantization": mc.quantization}
        for k, v in have.items():
            if meta.get(k) != v:
                raise ValueError(f"sharded state in {path} was saved with {k}={meta.get(k)!r}, "
                                 f"this engine runs with {k}={v!r}")
        pattern = meta.get("pattern", pattern)
    fn = os.path.join(path, pattern.format(rank=rank))
    if not os.path.isfile(fn):
        raise FileNotFoundError(f"no shard file for rank {rank}: {fn}")
    n = 0
    with safe_open(fn, framework="pt", device="cpu") as f, torch.no_grad():
        for name in f.keys():
            t = f.get_tensor(name)
            mod_name, _, attr = name.rpartition(".")
            mod = model.get_submodule(mod_name) if mod_name else model
            cur = getattr(mod, attr, None)
            if isinstance(cur, torch.Tensor) and tuple(cur.shape) == tuple(t.shape) and cur.dtype == t.dtype:
                cur.data.copy_(t)
            else:
                # tensors that only exist once weights are finalised (packed 4-bit tiles, fp8
                # weights and their scales) take the place of the checkpoint-layout parameters
                dev = next(model.parameters()).device
                t = t.to(dev)
                if attr in mod._buffers:
                    mod._buffers[attr] = t
                elif attr in mod._parameters:
                    mod._parameters[attr] = torch.nn.Parameter(t, requires_grad=False)
                else:
                    raise KeyError(f"sharded state names {name}, which this model does not have")
            n += 1
    for mod in model.modules():            # checkpoint-layout leftovers of layers loaded in kernel layout
        if getattr(mod, "qw", None) is not None:
            for raw in ("qweight", "qzeros", "scales", "weight_packed", "weight_block_scale", "weight_alpha"):
                if raw in mod._parameters:
                    del mod._parameters[raw]
    return n
