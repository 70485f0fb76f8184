"""Architecture name -> model class."""
from __future__ import annotations

import importlib
from typing import Dict, Tuple, Type

_MODELS: Dict[str, Tuple[str, str]] = {
    "LlamaForCausalLM": ("llama", "LlamaForCausalLM"),
    "MistralForCausalLM": ("llama", "LlamaForCausalLM"),
    "Qwen2ForCausalLM": ("llama", "LlamaForCausalLM"),
    "Qwen3ForCausalLM": ("llama", "LlamaForCausalLM"),
    # Llama-shaped with a rotary embedding over adjacent pairs (and, GLM, over half of each head):
    # folded into a row order of q_proj / k_proj at load (llama.rope_row_permutation)
    "HeliumForCausalLM": ("llama", "LlamaForCausalLM"),
    "Ernie4_5ForCausalLM": ("llama", "LlamaForCausalLM"),
    "GlmForCausalLM": ("llama", "LlamaForCausalLM"),
    "Qwen3MoeForCausalLM": ("qwen3_moe", "Qwen3MoeForCausalLM"),
    "Qwen2MoeForCausalLM": ("qwen2_moe", "Qwen2MoeForCausalLM"),
    "DeepseekV2ForCausalLM": ("deepseek", "DeepseekV2ForCausalLM"),
    "DeepseekV3ForCausalLM": ("deepseek", "DeepseekV3ForCausalLM"),
    "Ernie4_5_MoeForCausalLM": ("deepseek", "Ernie4_5_MoeForCausalLM"),
    "Glm4MoeForCausalLM": ("deepseek", "Glm4MoeForCausalLM"),
    "Dots1ForCausalLM": ("deepseek", "Dots1ForCausalLM"),
    "GraniteMoeForCausalLM": ("mixtral", "GraniteMoeForCausalLM"),
    "MixtralForCausalLM": ("mixtral", "MixtralForCausalLM"),
    "GptOssForCausalLM": ("gpt_oss", "GptOssForCausalLM"),
    # Phi-3 checkpoints store qkv and gate/up already fused: the fused loaders take them whole
    "Phi3ForCausalLM": ("llama", "LlamaForCausalLM"),
    "PhiForCausalLM": ("phi", "PhiForCausalLM"),
    "OPTForCausalLM": ("opt", "OPTForCausalLM"),
    "GemmaForCausalLM": ("llama_variants", "GemmaForCausalLM"),
    "Gemma2ForCausalLM": ("llama_variants", "Gemma2ForCausalLM"),
    "Gemma3ForCausalLM": ("llama_variants", "Gemma3ForCausalLM"),
    "GraniteForCausalLM": ("llama_variants", "GraniteForCausalLM"),
    # Llama's layer with other bias switches (attention_out_bias) / a window per layer_types entry
    "SeedOssForCausalLM": ("llama", "LlamaForCausalLM"),
    "MinistralForCausalLM": ("llama", "LlamaForCausalLM"),
    "SmolLM3ForCausalLM": ("llama_variants", "SmolLM3ForCausalLM"),
    "ArceeForCausalLM": ("llama_variants", "ArceeForCausalLM"),
    "Glm4ForCausalLM": ("llama_variants", "Glm4ForCausalLM"),
    "OlmoForCausalLM": ("llama_variants", "OlmoForCausalLM"),
    "OlmoeForCausalLM": ("qwen3_moe", "OlmoeForCausalLM"),
    "Olmo2ForCausalLM": ("llama_variants", "Olmo2ForCausalLM"),
    "Olmo3ForCausalLM": ("llama_variants", "Olmo3ForCausalLM"),
    "Exaone4ForCausalLM": ("llama_variants", "Exaone4ForCausalLM"),
    "StableLmForCausalLM": ("llama_variants", "StableLmForCausalLM"),
    "NemotronForCausalLM": ("llama_variants", "NemotronForCausalLM"),
    "CohereForCausalLM": ("llama_variants", "CohereForCausalLM"),
    "GPT2LMHeadModel": ("gpt2", "GPT2LMHeadModel"),
    "GPTBigCodeForCausalLM": ("gpt2", "GPTBigCodeForCausalLM"),
    "GPTJForCausalLM": ("phi", "GPTJForCausalLM"),
    "GPTNeoXForCausalLM": ("gpt_neox", "GPTNeoXForCausalLM"),
    "FalconForCausalLM": ("falcon", "FalconForCausalLM"),
    "RWForCausalLM": ("falcon", "FalconForCausalLM"),
    # the same block with ALiBi position biases and other bias / norm / activation switches
    "BloomForCausalLM": ("bloom", "BloomForCausalLM"),
    "MptForCausalLM": ("mpt", "MptForCausalLM"),
    "MPTForCausalLM": ("mpt", "MptForCausalLM"),
    "Starcoder2ForCausalLM": ("starcoder2", "Starcoder2ForCausalLM"),
    # vision-language: CLIP tower + projector in front of the Llama decoder
    "Cohere2ForCausalLM": ("llama_variants", "Cohere2ForCausalLM"),
    "PhimoeForCausalLM": ("mixtral", "PhimoeForCausalLM"),
    "DbrxForCausalLM": ("mixtral", "DbrxForCausalLM"),
    "GraniteMoeSharedForCausalLM": ("mixtral", "GraniteMoeSharedForCausalLM"),
    "LlavaForConditionalGeneration": ("llava", "LlavaForConditionalGeneration"),
    "LlavaNextForConditionalGeneration": ("llava", "LlavaNextForConditionalGeneration"),
    # draft heads of speculative decoding (never a served model by themselves; worker/draft_model.py)
    "EagleLlamaForCausalLM": ("spec_heads", "EagleLlamaForCausalLM"),
    "Eagle3LlamaForCausalLM": ("spec_heads", "Eagle3LlamaForCausalLM"),
    "LlamaForCausalLMEagle3": ("spec_heads", "Eagle3LlamaForCausalLM"),
    "DeepSeekMTP": ("spec_heads", "DeepSeekMTP"),
    "MedusaModel": ("spec_heads", "MedusaModel"),
}


def register_model(arch: str, module: str, cls: str) -> None:
    _MODELS[arch] = (module, cls)


def get_model_class(arch: str) -> Type:
    if arch not in _MODELS:
        raise ValueError(f"unsupported architecture {arch!r}; supported: {sorted(_MODELS)}")
    mod, cls = _MODELS[arch]
    if "." not in mod:
        mod = f"{__package__}.{mod}"
    return getattr(importlib.import_module(mod), cls)


def supported_architectures():
    return sorted(_MODELS)
