"""Configuration objects.

One picklable-by-value ``VllmConfig`` aggregates everything a worker needs;
the head ships it to spawned workers as a process argument and to remote
workers as an RPC argument (reference ``src/launch.py:27,85-87,162,238,280,
295,300``).  The fields the executor reads keep the reference's names:
``parallel_config.{world_size,tensor_parallel_size,pipeline_parallel_size}``,
``scheduler_config.async_scheduling``, ``kv_transfer_config``.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, List, Optional

from . import envs
from .logger import init_logger

logger = init_logger(__name__)

_CONFIG_DIR = os.path.join(os.path.dirname(__file__), "models", "configs")

# model-name -> bundled HF-style config file (no network on the GPU box)
_BUILTIN_ALIASES = {
    "meta-llama/meta-llama-3-8b": "llama-3-8b.json",
    "meta-llama/meta-llama-3-8b-instruct": "llama-3-8b.json",
    "meta-llama/llama-3.1-8b": "llama-3-8b.json",
    "llama-3-8b": "llama-3-8b.json",
    "meta-llama/meta-llama-3-70b": "llama-3-70b.json",
    "meta-llama/meta-llama-3-70b-instruct": "llama-3-70b.json",
    "llama-3-70b": "llama-3-70b.json",
    "facebook/opt-125m": "opt-125m.json",
    "opt-125m": "opt-125m.json",
    "llama-tiny": "llama-tiny.json",
    "opt-tiny": "opt-tiny.json",
    "qwen3-moe-tiny": "qwen3-moe-tiny.json",
    "mixtral-tiny": "mixtral-tiny.json",
    "phi3-tiny": "phi3-tiny.json",
    "mistralai/mixtral-8x7b-instruct-v0.1": "mixtral-8x7b.json",
    "mistralai/mixtral-8x7b-v0.1": "mixtral-8x7b.json",
    "mixtral-8x7b": "mixtral-8x7b.json",
    "microsoft/phi-3-mini-4k-instruct": "phi-3-mini-4k.json",
    "phi-3-mini-4k": "phi-3-mini-4k.json",
    "qwen/qwen3-coder-480b-a35b-instruct": "qwen3-coder-480b-a35b.json",
    "koushd/qwen3-coder-480b-a35b-instruct-awq": "qwen3-coder-480b-a35b.json",
    "qwen/qwen3-8b": "qwen3-8b.json",
    "qwen3-8b": "qwen3-8b.json",
    "mistralai/mistral-7b-v0.1": "mistral-7b.json",
    "mistral-7b": "mistral-7b.json",
}


def load_hf_config(model: str) -> Dict[str, Any]:
    """HF ``config.json`` as a dict: from a local directory, else a bundled file."""
    if os.path.isdir(model):
        path = os.path.join(model, "config.json")
        if os.path.isfile(path):
            with open(path) as f:
                return json.load(f)
    if os.path.isfile(model) and model.endswith(".json"):
        with open(model) as f:
            return json.load(f)
    # HF hub cache layout: $HF_HOME/hub/models--org--name/snapshots/<rev>/config.json
    hub = os.environ.get("HUGGINGFACE_HUB_CACHE") or os.path.join(
        os.environ.get("HF_HOME", os.path.expanduser("~/.cache/huggingface")), "hub")
    cand = os.path.join(hub, "models--" + model.replace("/", "--"), "snapshots")
    if os.path.isdir(cand):
        for rev in sorted(os.listdir(cand)):
            path = os.path.join(cand, rev, "config.json")
            if os.path.isfile(path):
                with open(path) as f:
                    return json.load(f)
    alias = _BUILTIN_ALIASES.get(model.lower())
    if alias is None and os.path.isfile(os.path.join(_CONFIG_DIR, os.path.basename(model.lower()) + ".json")):
        alias = os.path.basename(model.lower()) + ".json"       # any bundled config by its file name
    if alias is not None:
        with open(os.path.join(_CONFIG_DIR, alias)) as f:
            return json.load(f)
    raise FileNotFoundError(
        f"no config.json for model {model!r} (not a local directory, not in the HF cache, "
        f"not bundled; bundled: {sorted(set(_BUILTIN_ALIASES.values()))})")


def normalize_hf_config(cfg: Dict[str, Any]) -> None:
    """Key spellings that differ between ``transformers`` releases, mapped onto the ones the model
    code reads (in place): newer releases write ``num_local_experts`` / ``dtype`` / a ``rope_parameters``
    block where older checkpoints have ``num_experts`` / ``torch_dtype`` / ``rope_theta`` + ``rope_scaling``."""
    if cfg.get("model_type") in ("llava", "llava_next") and "mm_model_type" not in cfg:
        # a vision-language wrapper: the decoder's keys move to the top level, where every accessor reads them
        # (transformers fills an absent text_config / vision_config with the Llama-7B / CLIP ViT-B defaults)
        given = dict(cfg.get("text_config") or {})
        given.pop("architectures", None)
        defaults = {"model_type": "llama", "hidden_size": 4096, "intermediate_size": 11008, "num_hidden_layers": 32,
                    "num_attention_heads": 32, "rms_norm_eps": 1e-6, "vocab_size": 32000,
                    "max_position_embeddings": 2048, "rope_theta": 10000.0, "tie_word_embeddings": False}
        cfg["mm_model_type"] = cfg["model_type"]
        if cfg["mm_model_type"] == "llava_next":
            # (transformers' defaults: CLIP ViT-L/14 at 336 and these five tiled resolutions)
            cfg.setdefault("image_grid_pinpoints", [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]])
            cfg["vision_config"] = dict({"hidden_size": 1024, "intermediate_size": 4096, "num_hidden_layers": 24,
                                         "num_attention_heads": 16, "image_size": 336, "patch_size": 14},
                                        **(cfg.get("vision_config") or {}))
        cfg["model_type"] = given.get("model_type", "llama")
        for k, v in given.items():              # what the decoder's own config says wins over the wrapper's copy
            cfg[k] = v
        for k, v in defaults.items():
            if k not in given and (cfg.get(k) is None or k == "tie_word_embeddings"):
                cfg[k] = v
        cfg["vision_config"] = dict({"hidden_size": 768, "intermediate_size": 3072, "num_hidden_layers": 12,
                                     "num_attention_heads": 12, "image_size": 224, "patch_size": 32,
                                     "layer_norm_eps": 1e-5, "hidden_act": "quick_gelu"}, **(cfg.get("vision_config") or {}))
        cfg.setdefault("architectures", ["LlavaNextForConditionalGeneration" if cfg["mm_model_type"] == "llava_next"
                                         else "LlavaForConditionalGeneration"])
    if cfg.get("model_type") == "dbrx":
        # DBRX nests its attention and expert settings
        a, f = cfg.get("attn_config") or {}, cfg.get("ffn_config") or {}
        cfg.setdefault("num_key_value_heads", a.get("kv_n_heads"))
        if cfg.get("num_key_value_heads") is None:
            cfg["num_key_value_heads"] = a.get("kv_n_heads", 1)
        for k, v in (("clip_qkv", a.get("clip_qkv")), ("rope_theta", a.get("rope_theta", 10000.0)),
                     ("intermediate_size", f.get("ffn_hidden_size")), ("num_local_experts", f.get("moe_num_experts")),
                     ("num_experts_per_tok", f.get("moe_top_k")), ("hidden_act", (f.get("ffn_act_fn") or {}).get("name", "silu"))):
            if cfg.get(k) is None and v is not None:
                cfg[k] = v
        cfg.setdefault("tie_word_embeddings", False)
    # GPT-2 era spellings
    for old, new in (("n_embd", "hidden_size"), ("n_head", "num_attention_heads"), ("n_layer", "num_hidden_layers"),
                     ("n_positions", "max_position_embeddings"),
                     # MPT
                     ("d_model", "hidden_size"), ("n_heads", "num_attention_heads"), ("n_layers", "num_hidden_layers"),
                     ("max_seq_len", "max_position_embeddings")):
        if cfg.get(new) is None and cfg.get(old) is not None:
            cfg[new] = cfg[old]
    if cfg.get("num_experts") is None and cfg.get("num_local_experts") is not None:
        cfg["num_experts"] = cfg["num_local_experts"]
    if cfg.get("num_local_experts") is None and cfg.get("num_experts") is not None:
        cfg["num_local_experts"] = cfg["num_experts"]
    if "torch_dtype" not in cfg and isinstance(cfg.get("dtype"), str):
        cfg["torch_dtype"] = cfg["dtype"]
    if cfg.get("model_type") in ("ernie4_5", "ernie4_5_moe") and cfg.get("use_bias") is not None:
        cfg.setdefault("attention_bias", bool(cfg["use_bias"]))      # one switch for every linear layer
        cfg.setdefault("mlp_bias", bool(cfg["use_bias"]))
    rp = cfg.get("rope_parameters")
    if isinstance(rp, dict):
        if "rope_theta" in rp and cfg.get("rope_theta") is None:
            cfg["rope_theta"] = rp["rope_theta"]
        if cfg.get("rope_scaling") is None and rp.get("rope_type", "default") not in ("default", None):
            cfg["rope_scaling"] = {k: v for k, v in rp.items() if k != "rope_theta"}
        for k in ("partial_rotary_factor",):
            if k in rp and cfg.get(k) is None:
                cfg[k] = rp[k]
    rs = cfg.get("rope_scaling")
    if isinstance(rs, dict) and rs.get("rope_type", rs.get("type")) in ("longrope", "su"):
        # LongRoPE (Phi-3 128k, Phi-3.5, Phi-4-mini): the table builder needs the two context lengths of the config
        rs = cfg["rope_scaling"] = dict(rs)
        rs.setdefault("original_max_position_embeddings", cfg.get("original_max_position_embeddings"))
        rs.setdefault("max_position_embeddings", cfg.get("max_position_embeddings"))


def layer_sliding_window(c: Dict[str, Any], layer_idx: int) -> Optional[int]:
    """The attention window of decoder layer ``layer_idx`` of a model with config ``c``, or None for
    global attention.  Mistral / StarCoder2 / Phi-3 give every layer ``sliding_window``; Qwen2 only with
    ``use_sliding_window`` and only from layer ``max_window_layers`` on; newer configs list a
    ``layer_types`` entry per layer."""
    w = c.get("sliding_window")
    if not w or c.get("use_sliding_window") is False:
        return None
    types = c.get("layer_types")
    if types:
        return int(w) if layer_idx < len(types) and types[layer_idx] == "sliding_attention" else None
    if c.get("use_sliding_window") is True and c.get("max_window_layers") is not None:
        return int(w) if layer_idx >= int(c["max_window_layers"]) else None
    return int(w)


def is_bundled_architecture_name(model: str) -> bool:
    """Short names of the config files shipped in ``models/configs`` (``llama-tiny``,
    ``llama-3-70b``, ...): architecture descriptions for tests and benchmarks, which only
    ever run with random weights.  Hub ids (``org/name``) and paths are not."""
    name = model.lower()
    return ("/" not in name and os.sep not in name
            and os.path.isfile(os.path.join(_CONFIG_DIR, _BUILTIN_ALIASES.get(name, name + ".json"))))


def resolve_model_path(model: str) -> Optional[str]:
    """Directory holding weights/tokenizer for ``model`` or None."""
    if os.path.isdir(model):
        return model
    hub = os.environ.get("HUGGINGFACE_HUB_CACHE") or os.path.join(
        os.environ.get("HF_HOME", os.path.expanduser("~/.cache/huggingface")), "hub")
    cand = os.path.join(hub, "models--" + model.replace("/", "--"), "snapshots")
    if os.path.isdir(cand):
        revs = sorted(os.listdir(cand))
        if revs:
            return os.path.join(cand, revs[-1])
    return None


@dataclass
class ModelConfig:
    model: str = "llama-tiny"
    tokenizer: Optional[str] = None
    served_model_name: Optional[List[str]] = None
    dtype: str = "auto"                # auto|bfloat16|float16|float32
    seed: int = 0
    max_model_len: Optional[int] = None
    load_format: str = "auto"          # auto|safetensors|dummy
    # None | awq (4-bit weights stay packed, W4A16 kernels) | awq-dequant (dense at load)
    # | fp8 (W8A8: weights quantised per channel at load, activations per token at run time)
    # | nvfp4 (E2M1 weights with e4m3 scales per 16 columns stay packed; activations stay 16-bit)
    quantization: Optional[str] = None
    trust_remote_code: bool = False
    revision: Optional[str] = None
    # "auto": generation_config.json next to the weights; "vllm": none; else a directory (or file) to read
    generation_config: str = "auto"
    hf_overrides: Dict[str, Any] = field(default_factory=dict)
    # populated by __post_init__
    hf_config: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self) -> None:
        if not self.hf_config:
            self.hf_config = dict(load_hf_config(self.model))
        self.hf_config.update(self.hf_overrides or {})
        normalize_hf_config(self.hf_config)
        if self.tokenizer is None:
            self.tokenizer = self.model
        if self.served_model_name is None:
            self.served_model_name = [self.model]
        elif isinstance(self.served_model_name, str):
            self.served_model_name = [self.served_model_name]
        if self.dtype == "auto":
            td = self.hf_config.get("torch_dtype", "bfloat16")
            self.dtype = {"float16": "float16", "half": "float16", "float32": "float32",
                          "bfloat16": "bfloat16"}.get(str(td), "bfloat16")
        if self.dtype in ("half", "fp16"):
            self.dtype = "float16"
        if self.dtype in ("bf16",):
            self.dtype = "bfloat16"
        if self.dtype in ("float", "fp32"):
            self.dtype = "float32"
        derived = self.derived_max_model_len()
        if self.max_model_len is None:
            self.max_model_len = derived
        if self.quantization is None:
            q = self.hf_config.get("quantization_config") or {}
            if q.get("quant_method") == "awq":
                self.quantization = "awq"
            elif q.get("quant_method") == "gptq":
                self.quantization = "awq"        # same integers, repacked at load (models/loader._gptq_as_awq)
            elif "nvfp4" in str(q.get("format", "")).lower() or str(q.get("quant_algo", "")).upper() == "NVFP4":
                self.quantization = "nvfp4"      # compressed-tensors / modelopt exports
            elif q.get("quant_method") == "fp8" or "float-quantized" in str(q.get("format", "")).lower():
                # e4m3 weights with block / channel / tensor scales: widened by the loader, then
                # quantised per output channel for the W8A8 kernels like any ``--quantization fp8`` model
                self.quantization = "fp8"
        if self.load_format == "auto" and resolve_model_path(self.model) is None:
            if not is_bundled_architecture_name(self.model):
                # a typo, an id that is not in the cache, an unmounted volume: serving random
                # weights behind a healthy-looking API would be the worst possible answer
                raise FileNotFoundError(
                    f"no weights found for model {self.model!r} (not a local directory and not in the "
                    f"Hugging Face cache); pass --load-format dummy to run its architecture with random weights")
            logger.warning("%s names a bundled architecture description, not a checkpoint: using "
                           "deterministic random init (load_format=dummy)", self.model)
            self.load_format = "dummy"

    # ---------------------------------------------------- generation defaults
    def load_generation_config(self) -> Dict[str, Any]:
        """The model's ``generation_config.json`` ({} when absent or disabled)."""
        src = self.generation_config
        if src == "vllm":
            return {}
        path = resolve_model_path(self.model) if src == "auto" else src
        if path is None:
            return {}
        if os.path.isdir(path):
            path = os.path.join(path, "generation_config.json")
        if not os.path.isfile(path):
            return {}
        try:
            with open(path) as f:
                cfg = json.load(f)
            return cfg if isinstance(cfg, dict) else {}
        except (OSError, ValueError) as e:
            logger.warning("could not read %s: %s", path, e)
            return {}

    def default_sampling_params(self) -> Dict[str, Any]:
        """Sampling defaults the model's authors ship (applied to API requests that leave a field unset)."""
        g = self.load_generation_config()
        out: Dict[str, Any] = {}
        for key in ("temperature", "top_p", "top_k", "min_p", "repetition_penalty"):
            if g.get(key) is not None:
                out[key] = g[key]
        if g.get("max_new_tokens") is not None:
            out["max_tokens"] = int(g["max_new_tokens"])
        if g.get("do_sample") is False:             # greedy by declaration
            out["temperature"] = 0.0
        return out

    def eos_token_ids(self) -> List[int]:
        """Every id that ends a sequence: config.json and generation_config.json may each list several
        (Llama-3 instruct: <|end_of_text|> and <|eot_id|>)."""
        ids: List[int] = []
        for src in (self.hf_config.get("eos_token_id"), self.load_generation_config().get("eos_token_id")):
            for t in (src if isinstance(src, list) else [src]):
                if isinstance(t, int) and t not in ids:
                    ids.append(t)
        return ids

    # ------------------------------------------------------------ accessors
    def derived_max_model_len(self) -> int:
        c = self.hf_config
        for key in ("max_position_embeddings", "n_positions", "max_sequence_length", "seq_length"):
            if key in c and c[key]:
                return int(c[key])
        return 2048

    def layer_sliding_window(self, layer_idx: int) -> Optional[int]:
        return layer_sliding_window(self.hf_config, layer_idx)

    @property
    def architecture(self) -> str:
        archs = self.hf_config.get("architectures") or []
        if archs:
            return archs[0]
        mt = self.hf_config.get("model_type", "llama")
        return {"llama": "LlamaForCausalLM", "opt": "OPTForCausalLM",
                "qwen3_moe": "Qwen3MoeForCausalLM", "qwen2": "Qwen2ForCausalLM",
                "qwen2_moe": "Qwen2MoeForCausalLM", "phi": "PhiForCausalLM",
                "gemma3_text": "Gemma3ForCausalLM", "gemma2": "Gemma2ForCausalLM",
                "helium": "HeliumForCausalLM", "ernie4_5": "Ernie4_5ForCausalLM", "glm": "GlmForCausalLM",
                "seed_oss": "SeedOssForCausalLM", "ministral": "MinistralForCausalLM",
                "smollm3": "SmolLM3ForCausalLM", "arcee": "ArceeForCausalLM", "glm4": "Glm4ForCausalLM",
                "falcon": "FalconForCausalLM", "ernie4_5_moe": "Ernie4_5_MoeForCausalLM", "dbrx": "DbrxForCausalLM", "granitemoe": "GraniteMoeForCausalLM", "granitemoeshared": "GraniteMoeSharedForCausalLM", "gpt_bigcode": "GPTBigCodeForCausalLM", "gptj": "GPTJForCausalLM", "glm4_moe": "Glm4MoeForCausalLM", "dots1": "Dots1ForCausalLM", "deepseek_v2": "DeepseekV2ForCausalLM", "deepseek_v3": "DeepseekV3ForCausalLM",
                "olmo2": "Olmo2ForCausalLM", "olmo": "OlmoForCausalLM", "olmoe": "OlmoeForCausalLM", "olmo3": "Olmo3ForCausalLM", "exaone4": "Exaone4ForCausalLM",
                "stablelm": "StableLmForCausalLM", "nemotron": "NemotronForCausalLM",
                "cohere": "CohereForCausalLM", "cohere2": "Cohere2ForCausalLM", "phimoe": "PhimoeForCausalLM",
                "qwen3": "Qwen3ForCausalLM", "mistral": "MistralForCausalLM",
                "mixtral": "MixtralForCausalLM", "phi3": "Phi3ForCausalLM", "gpt_oss": "GptOssForCausalLM",
                "bloom": "BloomForCausalLM", "mpt": "MptForCausalLM"}.get(mt, mt)

    @property
    def hidden_size(self) -> int:
        return int(self.hf_config["hidden_size"])

    @property
    def num_hidden_layers(self) -> int:
        return int(self.hf_config.get("num_hidden_layers", self.hf_config.get("n_layer", 0)))

    @property
    def num_attention_heads(self) -> int:
        return int(self.hf_config["num_attention_heads"])

    @property
    def num_kv_heads(self) -> int:
        return int(self.hf_config.get("num_key_value_heads") or self.num_attention_heads)

    @property
    def head_dim(self) -> int:
        hd = self.hf_config.get("head_dim")
        return int(hd) if hd else self.hidden_size // self.num_attention_heads

    @property
    def vocab_size(self) -> int:
        return int(self.hf_config["vocab_size"])

    @property
    def quant_group_size(self) -> int:
        q = self.hf_config.get("quantization_config") or {}
        return int(q.get("group_size", q.get("q_group_size", 128)))

    @property
    def torch_dtype(self):
        import torch
        return {"bfloat16": torch.bfloat16, "float16": torch.float16,
                "float32": torch.float32}[self.dtype]

    @property
    def dtype_bytes(self) -> int:
        return {"bfloat16": 2, "float16": 2, "float32": 4}[self.dtype]

    def get_num_kv_heads(self, tp: int) -> int:
        return max(1, self.num_kv_heads // tp)


@dataclass
class CacheConfig:
    block_size: int = 16
    gpu_memory_utilization: float = 0.90
    num_gpu_blocks_override: Optional[int] = None
    enable_prefix_caching: bool = True
    cache_dtype: str = "auto"
    cpu_kv_bytes: int = 1 << 30        # KV budget when running on CPU
    # filled in by the engine after memory profiling
    num_gpu_blocks: Optional[int] = None

    def __post_init__(self) -> None:
        # the paged cache holds K/V in the model dtype, or in OCP e4m3 with per-head scales
        # (e5m2 and the fnuz encodings have no conversion instructions on gfx950)
        if self.cache_dtype not in ("auto", "bfloat16", "float16", "bf16", "fp16", "half", "fp8", "fp8_e4m3"):
            raise ValueError(f"kv cache dtype {self.cache_dtype!r} is not supported: use 'auto' (the model's "
                             "16-bit dtype) or 'fp8' / 'fp8_e4m3'")
        # the V cache stores keys in groups of 8 (one 16-byte MFMA operand chunk per head-dim row)
        if self.block_size < 8 or self.block_size % 8 != 0:
            raise ValueError("block_size must be a positive multiple of 8")

    @property
    def kv_fp8(self) -> bool:
        return self.cache_dtype in ("fp8", "fp8_e4m3")


@dataclass
class SpeculativeConfig:
    """Speculative decoding (vLLM's ``--speculative-config '{...}'``): the model scores a request's last
    token and a few draft tokens in one step and keeps the drafts it reproduces, plus one token of its own.

    ``method``
      * ``ngram``: prompt lookup - drafts are the tokens that followed the most recent earlier
        occurrence of the text's current n-gram suffix (``prompt_lookup_min`` .. ``prompt_lookup_max``);
      * ``draft_model``: a smaller model with the same vocabulary (``model``) runs on the workers next
        to the target, with its own KV cache over the same blocks, and proposes greedily;
      * ``eagle``: an EAGLE head (``model``: its checkpoint) drafts from the target's hidden states;
      * ``eagle3``: an EAGLE-3 head does, from the residual stream at three depths of the target;
      * ``mtp`` (``deepseek_mtp``): the multi-token-prediction module inside a DeepSeek-V3 checkpoint does
        the same (``model`` defaults to the target's checkpoint);
      * ``medusa``: Medusa heads (``model``: their checkpoint), one per draft position, over the target's
        last hidden state."""
    method: str = "ngram"
    num_speculative_tokens: int = 4
    prompt_lookup_max: int = 4
    prompt_lookup_min: int = 2
    model: Optional[str] = None              # draft model (a bundled name, a path or a hub id)
    load_format: Optional[str] = None        # of the draft model; default: the target's
    quantization: Optional[str] = None       # of the draft model

    METHODS = ("ngram", "draft_model", "eagle", "eagle3", "mtp", "medusa")

    def __post_init__(self) -> None:
        if self.method in (None, "") or (self.method == "ngram" and self.model):
            self.method = "draft_model" if self.model else "ngram"
        self.method = {"[ngram]": "ngram", "deepseek_mtp": "mtp"}.get(self.method, self.method)
        if self.method not in self.METHODS:
            raise ValueError(f"speculative method {self.method!r} is not supported ({', '.join(self.METHODS)})")
        if self.method in ("draft_model", "eagle", "eagle3", "medusa") and not self.model:
            raise ValueError(f"speculative method {self.method} needs a \"model\"")
        if not 1 <= self.num_speculative_tokens <= 16:
            raise ValueError("num_speculative_tokens must be between 1 and 16")
        if not 1 <= self.prompt_lookup_min <= self.prompt_lookup_max:
            raise ValueError("need 1 <= prompt_lookup_min <= prompt_lookup_max")

    @property
    def worker_side(self) -> bool:
        """Drafts come from a model on the workers (and travel back with each step's output)."""
        return self.method != "ngram"

    @property
    def reads_target_hidden(self) -> bool:
        return self.method in ("eagle", "eagle3", "mtp", "medusa")

    @classmethod
    def from_value(cls, value) -> Optional["SpeculativeConfig"]:
        import json
        if value in (None, "", {}):
            return None
        if isinstance(value, str):
            value = json.loads(value)
        known = {k: value[k] for k in ("method", "num_speculative_tokens", "prompt_lookup_max", "prompt_lookup_min",
                                       "model", "load_format", "quantization") if k in value}
        if "model" in known and "method" not in known:
            known["method"] = "draft_model"
        return cls(**known)


@dataclass
class ParallelConfig:
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1
    data_parallel_size: int = 1
    enable_expert_parallel: bool = False
    # uni | distributed | spmd  ("external_launcher" is accepted as "spmd",
    # "mp" as "distributed") or an Executor subclass
    distributed_executor_backend: Any = None
    pp_layer_partition: Optional[str] = None
    disable_custom_all_reduce: bool = False
    # dotted path of the worker class each wrapper instantiates (default: worker.worker.Worker)
    worker_cls: Optional[str] = None
    world_size: int = field(init=False, default=1)

    def __post_init__(self) -> None:
        self.world_size = self.tensor_parallel_size * self.pipeline_parallel_size
        if self.pp_layer_partition is None:
            self.pp_layer_partition = envs.VLLM_PP_LAYER_PARTITION
        b = self.distributed_executor_backend
        if isinstance(b, str):
            b = {"external_launcher": "spmd", "mp": "distributed", "ray": "distributed",
                 "uniproc": "uni"}.get(b, b)
            self.distributed_executor_backend = b

    def rank_of(self, pp_rank: int, tp_rank: int, dp_rank: int = 0) -> int:
        """Rank layout ``[DP][PP][TP]``; with DP=1 this is the reference's
        ``rank = pp_rank * TP + tp_rank`` (``src/launch.py:215,234,278``)."""
        return (dp_rank * self.pipeline_parallel_size + pp_rank) * self.tensor_parallel_size + tp_rank

    def coords_of(self, rank: int):
        tp = rank % self.tensor_parallel_size
        pp = (rank // self.tensor_parallel_size) % self.pipeline_parallel_size
        dp = rank // (self.tensor_parallel_size * self.pipeline_parallel_size)
        return dp, pp, tp


@dataclass
class SchedulerConfig:
    max_num_batched_tokens: Optional[int] = None
    max_num_seqs: int = 256
    max_model_len: int = 2048
    enable_chunked_prefill: bool = True
    async_scheduling: bool = False
    policy: str = "fcfs"               # fcfs | priority
    long_prefill_token_threshold: int = 0

    def __post_init__(self) -> None:
        if self.max_num_batched_tokens is None:
            self.max_num_batched_tokens = 8192 if self.enable_chunked_prefill else max(
                self.max_model_len, 8192)
        if not self.enable_chunked_prefill and self.max_num_batched_tokens < self.max_model_len:
            raise ValueError("max_num_batched_tokens must be >= max_model_len without chunked prefill")
        self.max_num_batched_tokens = max(self.max_num_batched_tokens, 1)


@dataclass
class DeviceConfig:
    device: str = "auto"               # auto|cuda|cpu

    def __post_init__(self) -> None:
        if self.device == "auto":
            from .utils import current_platform
            self.device = "cuda" if current_platform.is_rocm() else "cpu"


@dataclass
class KVTransferConfig:
    """Disaggregated prefill/decode: who produces and who consumes KV blocks."""
    kv_connector: str = "SharedStorageConnector"
    kv_role: str = "kv_both"           # kv_producer | kv_consumer | kv_both
    kv_connector_extra_config: Dict[str, Any] = field(default_factory=dict)


@dataclass
class CompilationConfig:
    """Decode-graph capture settings (hipGraph replaces a tracing compiler)."""
    use_hip_graph: bool = True
    capture_sizes: Optional[List[int]] = None
    max_capture_size: int = 512

    def sizes(self, max_num_seqs: int) -> List[int]:
        if self.capture_sizes:
            return sorted(set(int(s) for s in self.capture_sizes))
        out, s = [], 1
        top = min(self.max_capture_size, max_num_seqs)
        while s < 8 and s <= top:
            out.append(s)
            s *= 2
        s = 8
        while s <= top:
            out.append(s)
            s += 8 if s < 64 else (16 if s < 256 else 64)
        if top not in out and top >= 1:
            out.append(top)
        return sorted(set(out))


@dataclass
class ObservabilityConfig:
    collect_step_timing: bool = True
    otlp_traces_endpoint: Optional[str] = None


@dataclass
class LoRAConfig:
    """Adapter slots every worker keeps resident (``--enable-lora``)."""
    max_loras: int = 1
    max_lora_rank: int = 16


@dataclass
class VllmConfig:
    model_config: ModelConfig = field(default_factory=ModelConfig)
    cache_config: CacheConfig = field(default_factory=CacheConfig)
    parallel_config: ParallelConfig = field(default_factory=ParallelConfig)
    scheduler_config: SchedulerConfig = field(default_factory=SchedulerConfig)
    device_config: DeviceConfig = field(default_factory=DeviceConfig)
    compilation_config: CompilationConfig = field(default_factory=CompilationConfig)
    observability_config: ObservabilityConfig = field(default_factory=ObservabilityConfig)
    kv_transfer_config: Optional[KVTransferConfig] = None
    lora_config: Optional[LoRAConfig] = None
    speculative_config: Optional[SpeculativeConfig] = None

    def __post_init__(self) -> None:
        self.scheduler_config.max_model_len = self.model_config.max_model_len
        pc, mc = self.parallel_config, self.model_config
        if self.speculative_config is not None:
            if pc.pipeline_parallel_size > 1:
                raise ValueError("speculative decoding is not supported with pipeline parallelism")
            if self.scheduler_config.async_scheduling:
                # drafts are looked up in the tokens generated so far: the scheduler cannot run ahead of them
                logger.info("speculative decoding: async scheduling off")
                self.scheduler_config.async_scheduling = False
        if mc.num_attention_heads % pc.tensor_parallel_size != 0:
            raise ValueError(f"num_attention_heads ({mc.num_attention_heads}) must be divisible "
                             f"by tensor_parallel_size ({pc.tensor_parallel_size})")
        if self.scheduler_config.async_scheduling and pc.pipeline_parallel_size > 1:
            logger.warning("async scheduling needs the sampled tokens on the first pipeline stage; "
                           "disabled with pipeline_parallel_size > 1")
            self.scheduler_config.async_scheduling = False
        if pc.pipeline_parallel_size > mc.num_hidden_layers:
            raise ValueError("pipeline_parallel_size exceeds the number of layers")

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        be = d["parallel_config"].get("distributed_executor_backend")
        if be is not None and not isinstance(be, str):
            d["parallel_config"]["distributed_executor_backend"] = getattr(be, "__name__", str(be))
        return d


EngineConfig = VllmConfig
