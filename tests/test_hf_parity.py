"""Every model family against the implementation in the installed ``transformers`` package: a tiny
random-init Hugging Face model is saved as a safetensors checkpoint, loaded through the engine's own
loader and run on the CPU in fp32; first-token log-probabilities and greedy continuations must agree
with the Hugging Face forward pass on the same weights."""
import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.parallel import state as pstate

transformers = pytest.importorskip("transformers")

BASE = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
            num_key_value_heads=2, vocab_size=320, max_position_embeddings=128)
CASES = {
    "llama": ("LlamaConfig", dict(BASE, rope_theta=10000.0)),
    "llama3-rope": ("LlamaConfig", dict(BASE, rope_theta=500000.0, rope_scaling=dict(
        rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
        original_max_position_embeddings=32))),
    "llama-bias-tied": ("LlamaConfig", dict(BASE, attention_bias=True, mlp_bias=True, tie_word_embeddings=True)),
    "mistral": ("MistralConfig", dict(BASE, sliding_window=None)),
    # sliding windows shorter than the prompts: every query loses its oldest keys
    "mistral-window": ("MistralConfig", dict(BASE, sliding_window=16)),
    "starcoder2-window": ("Starcoder2Config", dict(BASE, sliding_window=24, use_bias=True)),
    "qwen2-window": ("Qwen2Config", dict(BASE, use_sliding_window=True, sliding_window=16, max_window_layers=1)),
    "qwen2": ("Qwen2Config", dict(BASE)),
    "qwen3": ("Qwen3Config", dict(BASE, head_dim=32)),
    "phi3": ("Phi3Config", dict(BASE, pad_token_id=0)),
    "opt": ("OPTConfig", dict(hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=4, vocab_size=320,
                              max_position_embeddings=128, word_embed_proj_dim=64)),
    "mixtral": ("MixtralConfig", dict(BASE, num_local_experts=4, num_experts_per_tok=2, sliding_window=None)),
    "gemma": ("GemmaConfig", dict(BASE, head_dim=32, hidden_activation="gelu_pytorch_tanh")),
    "granite": ("GraniteConfig", dict(BASE, embedding_multiplier=6.0, residual_multiplier=0.4,
                                      attention_multiplier=0.1, logits_scaling=4.0, tie_word_embeddings=True)),
    "gpt2": ("GPT2Config", dict(n_embd=64, n_layer=2, n_head=4, vocab_size=320, n_positions=128)),
    "gpt-neox": ("GPTNeoXConfig", dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                       num_attention_heads=4, vocab_size=320, max_position_embeddings=128,
                                       rotary_pct=0.5, use_parallel_residual=True)),
    "gpt-neox-serial": ("GPTNeoXConfig", dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                              num_attention_heads=4, vocab_size=320, max_position_embeddings=128,
                                              rotary_pct=0.25, use_parallel_residual=False)),
    "gpt-bigcode": ("GPTBigCodeConfig", dict(n_embd=64, n_layer=2, n_head=4, vocab_size=320, n_positions=128,
                                             multi_query=True)),
    "gptj": ("GPTJConfig", dict(n_embd=64, n_layer=2, n_head=4, vocab_size=320, n_positions=128, rotary_dim=8)),
    # the three Falcon block shapes: 40B (two norms, grouped KV), 7B (one norm, one KV head), sequential
    "falcon-new": ("FalconConfig", dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, vocab_size=320, alibi=False, bias=False, new_decoder_architecture=True, num_kv_heads=2)),
    "falcon-7b": ("FalconConfig", dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, vocab_size=320, alibi=False, bias=False, new_decoder_architecture=False, multi_query=True, parallel_attn=True)),
    "falcon-seq": ("FalconConfig", dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, vocab_size=320, alibi=False, new_decoder_architecture=False, multi_query=False, parallel_attn=False,
                                         bias=True)),
    # ALiBi instead of rotary embeddings: Falcon-RW (bias scaled with the scores), Bloom (both residual forms,
    # a head count that is not a power of two), MPT (its own slope sequence, no biases anywhere, q/k/v clamp)
    "falcon-rw": ("FalconConfig", dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, vocab_size=320, alibi=True,
                                       new_decoder_architecture=False, multi_query=False, parallel_attn=False, bias=True)),
    "bloom": ("BloomConfig", dict(hidden_size=96, n_layer=2, n_head=6, vocab_size=320)),
    "bloom-post-ln": ("BloomConfig", dict(hidden_size=64, n_layer=2, n_head=4, vocab_size=320,
                                          apply_residual_connection_post_layernorm=True)),
    "mpt": ("MptConfig", dict(d_model=96, n_heads=6, n_layers=2, vocab_size=320, max_seq_len=128,
                              attn_config=dict(alibi=True, alibi_bias_max=8, clip_qkv=0.8))),
    "starcoder2": ("Starcoder2Config", dict(BASE, sliding_window=None, use_bias=True)),
    "qwen3-moe": ("Qwen3MoeConfig", dict(BASE, head_dim=32, moe_intermediate_size=64, num_experts=8,
                                         num_experts_per_tok=2, norm_topk_prob=True, decoder_sparse_step=1,
                                         mlp_only_layers=[])),
    # sliding layers (own rotary base) and a global one, q/k norms, norms on both sides of each branch
    "gemma3": ("Gemma3TextConfig", dict(BASE, head_dim=32, sliding_window=16, query_pre_attn_scalar=24,
                                        num_hidden_layers=3, final_logit_softcapping=20.0,
                                        layer_types=["sliding_attention", "full_attention", "sliding_attention"])),
    "gemma2": ("Gemma2Config", dict(BASE, head_dim=32, sliding_window=16, query_pre_attn_scalar=24,
                                    num_hidden_layers=3, attn_logit_softcapping=5.0, final_logit_softcapping=20.0,
                                    layer_types=["sliding_attention", "full_attention", "sliding_attention"])),
    # rotary embedding over adjacent pairs (and, GLM, over half of every head, with q / k / v biases and a fused
    # gate_up weight): folded into the row order of q_proj / k_proj at load
    "helium": ("HeliumConfig", dict(BASE, head_dim=16)),
    "ernie4_5": ("Ernie4_5Config", dict(BASE, use_bias=True)),
    "glm": ("GlmConfig", dict(BASE, head_dim=16, pad_token_id=0)),
    "phi": ("PhiConfig", dict(BASE, partial_rotary_factor=0.5, layer_norm_eps=1e-5)),
    # Phi-2's head size of 80: padded to 96 for the cache and the attention kernels
    "phi-head80": ("PhiConfig", dict(BASE, hidden_size=320, num_key_value_heads=4, partial_rotary_factor=0.4)),
    # routed experts + a sigmoid-gated shared expert; one dense layer in between (mlp_only_layers)
    "qwen2-moe": ("Qwen2MoeConfig", dict(BASE, moe_intermediate_size=64, shared_expert_intermediate_size=96,
                                         num_experts=8, num_experts_per_tok=2, norm_topk_prob=False,
                                         decoder_sparse_step=1, mlp_only_layers=[1])),
    # --- families that are Llama's layer with norms, biases, MLP or positional embedding changed
    "smollm3": ("SmolLM3Config", dict(BASE, num_hidden_layers=3, no_rope_layers=[1, 0, 1], pad_token_id=0)),
    "seed-oss": ("SeedOssConfig", dict(BASE, head_dim=32, attention_bias=True, attention_out_bias=False,
                                       pad_token_id=0, bos_token_id=1, eos_token_id=2)),
    "ministral": ("MinistralConfig", dict(BASE, head_dim=16, sliding_window=16,
                                          layer_types=["sliding_attention", "full_attention"])),
    "arcee": ("ArceeConfig", dict(BASE, mlp_bias=True)),
    "glm4": ("Glm4Config", dict(BASE, head_dim=16, pad_token_id=0)),
    # multi-head latent attention (low-rank q and kv, one shared rotated key part), a dense first layer, then
    # routed experts (V3: sigmoid scores + choice bias, group-limited) + shared experts
    "deepseek-v3": ("DeepseekV3Config", dict(BASE, num_hidden_layers=3, num_key_value_heads=4, q_lora_rank=24, kv_lora_rank=16,
                                             qk_nope_head_dim=16, qk_rope_head_dim=8, v_head_dim=12,
                                             moe_intermediate_size=32, n_routed_experts=8, n_shared_experts=2,
                                             num_experts_per_tok=3, n_group=4, topk_group=2, first_k_dense_replace=1,
                                             routed_scaling_factor=2.5, norm_topk_prob=True, pad_token_id=0)),
    "deepseek-v2": ("DeepseekV2Config", dict(BASE, num_hidden_layers=3, q_lora_rank=None, kv_lora_rank=16,
                                             qk_nope_head_dim=16, qk_rope_head_dim=8, v_head_dim=24,
                                             moe_intermediate_size=32, n_routed_experts=8, n_shared_experts=1,
                                             num_experts_per_tok=2, n_group=4, topk_group=2, first_k_dense_replace=1,
                                             topk_method="group_limited_greedy", routed_scaling_factor=1.5,
                                             pad_token_id=0)),
    # V3's expert block behind grouped-query attention (GLM-4.5: half-rotated heads, q/k norms, q/k/v biases)
    "glm4-moe": ("Glm4MoeConfig", dict(BASE, num_hidden_layers=3, head_dim=16, use_qk_norm=True, attention_bias=True,
                                       moe_intermediate_size=32, n_routed_experts=8, n_shared_experts=1,
                                       num_experts_per_tok=2, n_group=2, topk_group=1, first_k_dense_replace=1,
                                       routed_scaling_factor=1.5, norm_topk_prob=True, pad_token_id=0)),
    "dots1": ("Dots1Config", dict(BASE, num_hidden_layers=3, head_dim=16, moe_intermediate_size=32,
                                  n_routed_experts=8, n_shared_experts=1, num_experts_per_tok=2, n_group=2,
                                  topk_group=1, first_k_dense_replace=1, routed_scaling_factor=1.5,
                                  norm_topk_prob=True, pad_token_id=0, sliding_window=16,
                                  layer_types=["full_attention", "sliding_attention", "full_attention"])),
    "granite-moe": ("GraniteMoeConfig", dict(BASE, intermediate_size=32, num_local_experts=4, num_experts_per_tok=2,
                                             embedding_multiplier=6.0, residual_multiplier=0.4,
                                             attention_multiplier=0.1, logits_scaling=4.0, tie_word_embeddings=True,
                                             attention_bias=True, pad_token_id=0)),
    "granite-moe-shared": ("GraniteMoeSharedConfig", dict(BASE, intermediate_size=32, num_local_experts=4,
                                                          num_experts_per_tok=2, shared_intermediate_size=48,
                                                          embedding_multiplier=6.0, residual_multiplier=0.4,
                                                          attention_multiplier=0.1, logits_scaling=4.0,
                                                          tie_word_embeddings=True, pad_token_id=0)),
    "ernie4_5-moe": ("Ernie4_5_MoeConfig", dict(BASE, num_hidden_layers=3, moe_intermediate_size=32, moe_num_experts=8,
                                                moe_k=2, moe_num_shared_experts=1, moe_layer_start_index=1,
                                                pad_token_id=0)),
    "olmo": ("OlmoConfig", dict(BASE, clip_qkv=0.6, pad_token_id=0)),
    "olmoe": ("OlmoeConfig", dict(BASE, intermediate_size=64, num_experts=8, num_experts_per_tok=2, clip_qkv=1.5,
                                  pad_token_id=0)),
    "olmo2": ("Olmo2Config", dict(BASE, pad_token_id=0)),
    "olmo3": ("Olmo3Config", dict(BASE, num_hidden_layers=3, sliding_window=16, pad_token_id=0,
                                  layer_types=["sliding_attention", "full_attention", "sliding_attention"])),
    "exaone4": ("Exaone4Config", dict(BASE, num_hidden_layers=3, sliding_window=16, pad_token_id=0,
                                      layer_types=["sliding_attention", "full_attention", "sliding_attention"])),
    "stablelm": ("StableLmConfig", dict(BASE, use_qkv_bias=True)),
    "nemotron": ("NemotronConfig", dict(BASE, attention_bias=True, mlp_bias=True)),
    "cohere": ("CohereConfig", dict(BASE, logit_scale=0.25, pad_token_id=0)),
    # nested attention / expert settings, fused and clamped Wqkv, experts stacked in three tensors (down transposed)
    "dbrx": ("DbrxConfig", dict(d_model=64, n_heads=4, n_layers=2, max_seq_len=128, vocab_size=320, pad_token_id=0,
                                attn_config=dict(kv_n_heads=2, clip_qkv=0.8, rope_theta=10000.0),
                                ffn_config=dict(ffn_hidden_size=48, moe_num_experts=6, moe_top_k=2,
                                                moe_normalize_expert_weights=1.0))),
    # LayerNorms with biases, biases on q / k / v / o and the LM head, the two-softmax "sparsemixer" router
    "phimoe": ("PhimoeConfig", dict(BASE, intermediate_size=48, num_local_experts=6, num_experts_per_tok=2,
                                    attention_bias=True, lm_head_bias=True, router_jitter_noise=0.05, pad_token_id=0)),
    # a window (shorter than the prompts) on three of four layers, no positional embedding on the global one
    "cohere2": ("Cohere2Config", dict(BASE, num_hidden_layers=4, logit_scale=0.25, pad_token_id=0, sliding_window=16,
                                      layer_types=["sliding_attention"] * 3 + ["full_attention"])),
    # attention sinks, q/k/v/o biases, a window on every other layer, YaRN; experts with biases and the clamped
    # SwiGLU, stacked and column-interleaved in the checkpoint; router with a bias, softmax over the chosen logits
    "gpt-oss": ("GptOssConfig", dict(BASE, head_dim=16, num_hidden_layers=3, intermediate_size=48, num_local_experts=6,
                                     num_experts_per_tok=2, sliding_window=16, pad_token_id=0,
                                     layer_types=["sliding_attention", "full_attention", "sliding_attention"],
                                     rope_scaling=dict(rope_type="yarn", factor=4.0, beta_fast=32.0, beta_slow=1.0,
                                                       original_max_position_embeddings=32, truncate=False))),
}
PROMPTS = [[5, 9, 200, 17, 33, 2, 90], list(range(10, 60)), [7, 7, 300, 2, 2, 41, 8, 100, 101]]


def _hf_model(name, tmp_path):
    cls, kw = CASES[name]
    cfg_cls = getattr(transformers, cls, None)
    if cfg_cls is None:
        pytest.skip(f"transformers has no {cls}")
    torch.manual_seed(1234)
    cfg = cfg_cls(**kw)
    model = transformers.AutoModelForCausalLM.from_config(cfg).eval().float()
    # the default initialisation leaves norms at one and biases at zero: perturb them so that a missing
    # bias, a swapped norm or an unapplied scale shows
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 or (p.dim() == 2 and n.endswith("_proj_bias")):      # (gpt-oss: one bias row per expert)
                p.add_(torch.randn_like(p) * 0.1)
        for n, b in model.named_buffers():
            if n.endswith("e_score_correction_bias"):       # DeepSeek-V3's choice bias is stored as a buffer
                b.add_(torch.randn_like(b) * 0.2)
        for n, p in model.named_parameters():
            if n.endswith("e_score_correction_bias"):       # (ERNIE: a parameter of shape [1, E])
                p.add_(torch.randn_like(p) * 0.2)
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    return model


@pytest.mark.parametrize("name", sorted(CASES))
def test_family_matches_transformers(name, tmp_path):
    model = _hf_model(name, tmp_path)
    ref_lp, ref_tokens = [], []
    with torch.no_grad():
        for p in PROMPTS:
            ids = torch.tensor([p])
            ref_lp.append(model(ids).logits[0, -1].float().log_softmax(-1))
            gen = model.generate(ids, max_new_tokens=4, do_sample=False, pad_token_id=0)
            ref_tokens.append(gen[0, len(p):].tolist())
    llm = LLM(model=str(tmp_path), dtype="float32", device="cpu", max_model_len=128, num_gpu_blocks_override=64,
              max_num_seqs=8)
    try:
        outs = llm.generate(PROMPTS, SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True, logprobs=10))
    finally:
        llm.shutdown()
        pstate.destroy_model_parallel()
    for out, lp, toks in zip(outs, ref_lp, ref_tokens):
        got = out.outputs[0].logprobs[0]
        top = torch.topk(lp, 5).indices.tolist()
        assert set(top) <= set(got), (name, top, sorted(got))
        for t in top:
            assert abs(got[t] - lp[t].item()) < 2e-3, (name, t, got[t], lp[t].item())
        assert list(out.outputs[0].token_ids) == toks, (name, out.outputs[0].token_ids, toks)


@pytest.mark.parametrize("mode", ["short", "long"])
def test_phi3_longrope_matches_transformers(mode, tmp_path):
    """LongRoPE (Phi-3 128k / Phi-3.5 / Phi-4-mini: per-frequency divisors, a short and a long set, a temperature on
    cos / sin, partial rotation): served inside the original context the short set applies, served beyond it the
    long one - each against transformers on sequences of that kind (prompts shorter / longer than the original
    context of 32)."""
    if not hasattr(transformers, "Phi3Config"):
        pytest.skip("transformers has no Phi-3")
    torch.manual_seed(99)
    n = 12 // 2                                     # rotated pairs: head_dim 16, partial_rotary_factor 0.75
    scaling = dict(rope_type="longrope", short_factor=[1.0 + 0.05 * i for i in range(n)],
                   long_factor=[1.5 + 0.6 * i for i in range(n)])
    kw = dict(BASE, pad_token_id=0, max_position_embeddings=256, original_max_position_embeddings=32,
              partial_rotary_factor=0.75)
    try:
        cfg = transformers.Phi3Config(**kw, rope_parameters=dict(scaling, rope_theta=10000.0, partial_rotary_factor=0.75,
                                                                 original_max_position_embeddings=32))
    except (TypeError, ValueError, KeyError):
        cfg = transformers.Phi3Config(**kw, rope_scaling=dict(scaling, type="longrope"))
    model = transformers.AutoModelForCausalLM.from_config(cfg).eval().float()
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    prompts = [[5, 9, 200, 17, 33, 2, 90], list(range(10, 30))] if mode == "short" else \
        [list(range(10, 60)), [7 + (3 * i) % 300 for i in range(75)]]
    ref_lp, ref_tokens = [], []
    with torch.no_grad():
        for p in prompts:
            ids = torch.tensor([p])
            ref_lp.append(model(ids).logits[0, -1].float().log_softmax(-1))
            gen = model.generate(ids, max_new_tokens=4, do_sample=False, pad_token_id=0)
            ref_tokens.append(gen[0, len(p):].tolist())
    llm = LLM(model=str(tmp_path), dtype="float32", device="cpu", max_model_len=32 if mode == "short" else 128,
              num_gpu_blocks_override=64, max_num_seqs=8)
    try:
        outs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True, logprobs=10))
    finally:
        llm.shutdown()
        pstate.destroy_model_parallel()
    for out, lp, toks in zip(outs, ref_lp, ref_tokens):
        got = out.outputs[0].logprobs[0]
        top = torch.topk(lp, 5).indices.tolist()
        assert set(top) <= set(got), (mode, top, sorted(got))
        for t in top:
            assert abs(got[t] - lp[t].item()) < 4e-3, (mode, t, got[t], lp[t].item())      # (scores carry the squared temperature)
        assert list(out.outputs[0].token_ids) == toks, (mode, out.outputs[0].token_ids, toks)


def test_llava_matches_transformers_with_images(tmp_path):
    """A small random LLaVA (CLIP tower, tapped at the second-to-last layer, two-layer projector, Llama decoder)
    against ``transformers.LlavaForConditionalGeneration`` on the same weights: pixels from this package's own
    preprocessing, one and two images per prompt, next to a text-only prompt; then the same prompts prefilled in
    8-token chunks, which cut through the images."""
    import numpy as np
    from PIL import Image
    if not hasattr(transformers, "LlavaForConditionalGeneration"):
        pytest.skip("transformers has no LLaVA")
    text = transformers.LlamaConfig(**dict(BASE, rope_theta=10000.0))
    vision = transformers.CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                                           image_size=28, patch_size=7, projection_dim=16)
    cfg = transformers.LlavaConfig(vision_config=vision, text_config=text, image_token_index=319,
                                   vision_feature_layer=-2, vision_feature_select_strategy="default")
    torch.manual_seed(77)
    model = transformers.LlavaForConditionalGeneration(cfg).eval().float()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    rng = np.random.default_rng(3)
    imgs = [Image.fromarray((rng.random(s) * 255).astype("uint8")) for s in ((40, 60, 3), (30, 29, 3), (64, 64, 3))]
    cases = [([5, 9, 319, 17, 33, 2, 90], [imgs[0]]), ([7, 319, 300, 2, 319, 41, 8], [imgs[1], imgs[2]]),
             (list(range(10, 40)), [])]
    from vllm_distributed_amd.multimodal import ClipImageProcessor
    proc = ClipImageProcessor(28, 28)
    ref_lp, ref_tokens = [], []
    with torch.no_grad():
        for ids, pics in cases:
            full = []
            for t in ids:
                full += [319] * 16 if t == 319 else [t]
            kw = {}
            if pics:
                kw["pixel_values"] = torch.from_numpy(np.stack([proc(im).astype(np.float32) for im in pics]))
            x = torch.tensor([full])
            ref_lp.append(model(input_ids=x, **kw).logits[0, -1].float().log_softmax(-1))
            gen = model.generate(input_ids=x, max_new_tokens=4, do_sample=False, pad_token_id=0, **kw)
            ref_tokens.append(gen[0, len(full):].tolist())
    prompts = [{"prompt_token_ids": ids, **({"multi_modal_data": {"image": pics}} if pics else {})} for ids, pics in cases]
    sp = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True, logprobs=10)
    for extra in ({}, {"max_num_batched_tokens": 8}):
        llm = LLM(model=str(tmp_path), dtype="float32", device="cpu", max_model_len=128, num_gpu_blocks_override=64,
                  max_num_seqs=8, **extra)
        try:
            outs = llm.generate(prompts, sp)
        finally:
            llm.shutdown()
            pstate.destroy_model_parallel()
        for out, lp, toks in zip(outs, ref_lp, ref_tokens):
            got = out.outputs[0].logprobs[0]
            top = torch.topk(lp, 5).indices.tolist()
            assert set(top) <= set(got), (top, sorted(got))
            for t in top:
                assert abs(got[t] - lp[t].item()) < 3e-3, (t, got[t], lp[t].item())
            assert list(out.outputs[0].token_ids) == toks
        assert len(outs[0].prompt_token_ids) == 6 + 16 and len(outs[1].prompt_token_ids) == 5 + 32


def test_llava_next_matches_transformers_with_images_of_several_aspects(tmp_path):
    """A small random LLaVA-NeXT against ``transformers.LlavaNextForConditionalGeneration`` on the same weights:
    wide, tall and near-square images (each picks another of the tiled resolutions and unpads another side), one
    and two images per prompt, whole and in 8-token chunks.  The views come from this package's own any-resolution
    preprocessing, which is compared with transformers' image processor first; the number of placeholder tokens
    per image comes from ``multimodal.anyres_tokens`` (transformers refuses a prompt whose count is off)."""
    import numpy as np
    from PIL import Image
    if not hasattr(transformers, "LlavaNextForConditionalGeneration"):
        pytest.skip("transformers has no LLaVA-NeXT")
    from vllm_distributed_amd.multimodal import AnyResImageProcessor, ClipImageProcessor, anyres_tokens
    pins = [[28, 56], [56, 28], [56, 56], [84, 28], [28, 84]]
    text = transformers.LlamaConfig(**dict(BASE, rope_theta=10000.0))
    vision = transformers.CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                                           image_size=28, patch_size=7, projection_dim=16)
    cfg = transformers.LlavaNextConfig(vision_config=vision, text_config=text, image_token_index=319,
                                       vision_feature_layer=-2, vision_feature_select_strategy="default",
                                       image_grid_pinpoints=pins)
    torch.manual_seed(78)
    model = transformers.LlavaNextForConditionalGeneration(cfg).eval().float()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    rng = np.random.default_rng(4)
    imgs = [Image.fromarray((rng.random(s) * 255).astype("uint8")) for s in ((40, 90, 3), (70, 29, 3), (50, 46, 3), (20, 37, 3))]
    proc = AnyResImageProcessor(ClipImageProcessor(28, 28), pins)
    mine = [proc(im) for im in imgs]
    sizes = [(im.size[1], im.size[0]) for im in imgs]
    n_tok = [anyres_tokens({"vision_config": {"image_size": 28, "patch_size": 7}, "image_grid_pinpoints": pins}, s) for s in sizes]
    assert len(set(m.shape[0] for m in mine)) > 1 and len(set(n_tok)) > 2, (n_tok, [m.shape for m in mine])
    try:
        hf_proc = transformers.LlavaNextImageProcessor(size={"shortest_edge": 28}, crop_size={"height": 28, "width": 28},
                                                       image_grid_pinpoints=pins)
        for im, px, sz in zip(imgs, mine, sizes):
            ref = hf_proc(images=im, return_tensors="np")
            assert tuple(ref["image_sizes"][0]) == sz and ref["pixel_values"][0].shape == px.shape
            assert np.abs(ref["pixel_values"][0] - px.astype(np.float32)).max() < 2e-3     # (fp16 rounding)
    except (ImportError, AttributeError):
        pass                         # (no PIL-based image processor in this transformers build)
    cases = [([5, 9, 319, 17, 33, 2, 90], [0]), ([7, 319, 300, 2, 319, 41, 8], [1, 2]), ([3, 319, 4], [3]),
             (list(range(10, 40)), [])]
    ref_lp, ref_tokens = [], []
    with torch.no_grad():
        for ids, pics in cases:
            full, k = [], 0
            for t in ids:
                if t == 319:
                    full += [319] * n_tok[pics[k]]
                    k += 1
                else:
                    full.append(t)
            kw = {}
            if pics:
                most = max(mine[i].shape[0] for i in pics)
                px = np.zeros((len(pics), most, 3, 28, 28), dtype=np.float32)
                for j, i in enumerate(pics):
                    px[j, :mine[i].shape[0]] = mine[i].astype(np.float32)
                kw = {"pixel_values": torch.from_numpy(px), "image_sizes": torch.tensor([sizes[i] for i in pics])}
            x = torch.tensor([full])
            ref_lp.append(model(input_ids=x, **kw).logits[0, -1].float().log_softmax(-1))
            gen = model.generate(input_ids=x, max_new_tokens=4, do_sample=False, pad_token_id=0, **kw)
            ref_tokens.append(gen[0, len(full):].tolist())
    prompts = [{"prompt_token_ids": ids, **({"multi_modal_data": {"image": [imgs[i] for i in pics]}} if pics else {})}
               for ids, pics in cases]
    sp = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True, logprobs=10)
    for extra in ({}, {"max_num_batched_tokens": 8}):
        llm = LLM(model=str(tmp_path), dtype="float32", device="cpu", max_model_len=256, num_gpu_blocks_override=64,
                  max_num_seqs=8, **extra)
        try:
            outs = llm.generate(prompts, sp)
        finally:
            llm.shutdown()
            pstate.destroy_model_parallel()
        for out, lp, toks in zip(outs, ref_lp, ref_tokens):
            got = out.outputs[0].logprobs[0]
            top = torch.topk(lp, 5).indices.tolist()
            assert set(top) <= set(got), (top, sorted(got))
            for t in top:
                assert abs(got[t] - lp[t].item()) < 3e-3, (t, got[t], lp[t].item())
            assert list(out.outputs[0].token_ids) == toks
        assert len(outs[0].prompt_token_ids) == 6 + n_tok[0] and len(outs[1].prompt_token_ids) == 5 + n_tok[1] + n_tok[2]


@pytest.mark.parametrize("name", ["deepseek-v3", "deepseek-v2"])
@pytest.mark.parametrize("latent", [True, False])
def test_deepseek_latent_and_expanded_caches_match_transformers(name, latent, tmp_path, monkeypatch):
    """Multi-head latent attention served both ways: the latent row in the cache with the up-projections
    absorbed into the query and the output (kv_lora_rank + rope + kv_lora_rank values per token), and per-head
    keys / values expanded in front of an ordinary cache.  Both against transformers, prompts whole and in
    16-token chunks (a chunk's rows attend over the cached latent of the chunks before it)."""
    monkeypatch.setenv("VDAMD_MLA_LATENT", "1" if latent else "0")
    model = _hf_model(name, tmp_path)
    ref_tokens = []
    with torch.no_grad():
        for p in PROMPTS:
            gen = model.generate(torch.tensor([p]), max_new_tokens=6, do_sample=False, pad_token_id=0)
            ref_tokens.append(gen[0, len(p):].tolist())
    for extra in ({}, {"max_num_batched_tokens": 16}):
        llm = LLM(model=str(tmp_path), dtype="float32", device="cpu", max_model_len=128, num_gpu_blocks_override=64,
                  max_num_seqs=8, **extra)
        try:
            runner = llm.engine.core.executor.driver.worker.model_runner
            attn = runner.model.attention_layers()[0]
            assert bool(getattr(attn, "latent", False)) == latent
            k, v = runner.kv_caches[0]
            rank, rope = CASES[name][1]["kv_lora_rank"], CASES[name][1]["qk_rope_head_dim"]
            if latent:
                assert k.shape[1] == 1 and k.shape[-1] == rank + rope and v.shape[2] * v.shape[4] == rank
                assert runner.get_kv_cache_spec().bytes_per_token == 3 * (2 * rank + rope) * 4      # 3 layers, fp32
            else:
                assert k.shape[1] == 4
            outs = llm.generate(PROMPTS, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        finally:
            llm.shutdown()
            pstate.destroy_model_parallel()
        assert [list(o.outputs[0].token_ids) for o in outs] == ref_tokens
