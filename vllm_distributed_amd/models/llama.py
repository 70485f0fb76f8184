"""Llama-family decoder (Llama-2/3, Mistral, Qwen2, Qwen3 dense).

Per layer: fused add+RMSNorm -> fused QKV GEMM -> fused rotary + KV-cache
write -> paged attention -> O GEMM (+all-reduce) -> fused add+RMSNorm ->
gate/up GEMM -> SiLU*mul -> down GEMM (+all-reduce).  A pipeline stage owns a
contiguous slice of layers; the first stage embeds, the last applies the final
norm and the LM head, and ``(hidden, residual)`` crosses stage boundaries.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import envs, ops
from ..config import VllmConfig, layer_sliding_window
from ..parallel.state import get_pp_group, get_tp_group
from .layers.attention import Attention, RMSNorm, get_forward_context
from .layers.moe import FusedMoE
from .layers.linear import (ColumnParallelLinear, ParallelLMHead, QKVParallelLinear, QuantSpec,
                            RowParallelLinear, VocabParallelEmbedding, nvfp4_rename, quantized_linears)
from .utils import IntermediateTensors, get_pp_layer_range, materialize


# families whose rotary embedding pairs ADJACENT dimensions (2i, 2i + 1) instead of (i, i + d/2)
INTERLEAVED_ROPE_TYPES = ("helium", "ernie4_5", "ernie4_5_moe", "glm", "glm4", "cohere", "cohere2")
# families whose ``attention_bias`` means q / k / v only
NO_O_BIAS_TYPES = ("glm", "glm4", "glm4_moe")
# families with an RMSNorm over each head of q and k before the rotation ...
HEAD_QK_NORM_TYPES = ("qwen3", "qwen3_moe", "gemma3_text", "exaone4", "dots1")
# ... and with one over the whole width of the q and of the k projection
FULL_QK_NORM_TYPES = ("olmo2", "olmo3", "olmoe")


def has_head_qk_norm(cfg: dict) -> bool:
    return cfg.get("model_type") in HEAD_QK_NORM_TYPES or (cfg.get("model_type") == "glm4_moe"
                                                             and bool(cfg.get("use_qk_norm")))


def rope_row_permutation(head_dim: int, rotary_dim: int, interleaved: bool) -> Optional[torch.Tensor]:
    """Row order of one head of ``q_proj`` / ``k_proj`` under which a rotary embedding over the first
    ``rotary_dim`` dimensions, pairing adjacent (``interleaved``) or half-apart dimensions, becomes the
    full-width half-apart rotation the fused rotary kernel applies: new row ``j`` is old row
    ``perm[j]``.  Scores ``q . k`` do not change when q and k are permuted alike, so this is a pure
    change of basis done once, on the weights; the cos / sin table gets cos = 1, sin = 0 for the
    pairs that are not rotated (``padded_cos_sin``).  None when nothing has to move."""
    rd, d = rotary_dim, head_dim
    if rd == d and not interleaved:
        return None
    first = [2 * i for i in range(rd // 2)] if interleaved else list(range(rd // 2))
    second = [2 * i + 1 for i in range(rd // 2)] if interleaved else list(range(rd // 2, rd))
    rest = list(range(rd, d))
    keep = len(rest) // 2
    return torch.tensor(first + rest[:keep] + second + rest[keep:], dtype=torch.long)


def padded_cos_sin(table: torch.Tensor, head_dim: int) -> torch.Tensor:
    """``[max_pos, rotary_dim]`` (cos | sin) -> ``[max_pos, head_dim]`` with identity rotations appended."""
    rd = table.shape[1]
    if rd == head_dim:
        return table
    extra = (head_dim - rd) // 2
    cos, sin = table[:, :rd // 2], table[:, rd // 2:]
    one = torch.ones(table.shape[0], extra, dtype=table.dtype, device=table.device)
    zero = torch.zeros_like(one)
    return torch.cat([cos, one, sin, zero], dim=1).contiguous()


def identity_cos_sin(table: torch.Tensor) -> torch.Tensor:
    """A cos / sin table of ``table``'s shape that rotates nothing: what layers without a positional
    embedding hand to the fused rotary + cache kernel."""
    half = table.shape[1] // 2
    out = torch.zeros_like(table)
    out[:, :half] = 1
    return out


class FullWidthRMSNorm(nn.Module):
    """RMSNorm over the whole q (or k) projection, all heads together (OLMo-2).  Under tensor
    parallelism a rank holds ``width`` of the columns: the sum of squares is all-reduced, and counted
    ``1 / replicas`` on each rank when KV heads are replicated."""

    def __init__(self, width: int, total_width: int, shard_index: int, replicas: int, eps: float, dtype: torch.dtype):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width, dtype=dtype), requires_grad=False)
        self.weight.weight_loader = self._load
        self.width, self.total, self.index, self.replicas, self.eps = width, total_width, shard_index, replicas, eps

    def _load(self, param: nn.Parameter, w, shard_id=None) -> None:
        param.data.copy_(materialize(w.narrow(0, self.index * self.width, self.width)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xf = x.float()
        ss = xf.pow(2).sum(-1, keepdim=True)
        if self.total != self.width:
            ss = get_tp_group().all_reduce(ss / self.replicas)
        return (xf * torch.rsqrt(ss / self.total + self.eps)).to(x.dtype) * self.weight


class LlamaMLP(nn.Module):
    def __init__(self, hidden: int, inter: int, dtype: torch.dtype, bias: bool = False, act: str = "silu"):
        super().__init__()
        self.gate_up_proj = ColumnParallelLinear(hidden, [inter, inter], bias, dtype)
        self.down_proj = RowParallelLinear(inter, hidden, bias, dtype)
        self.act = act                      # "silu" (SwiGLU) or a GELU flavour (GeGLU: Gemma)

    supports_residual = True

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False,
                residual: Optional[torch.Tensor] = None, partials: bool = False):
        """With ``residual``: None is returned when the down projection added its product into it.
        ``partials``: the caller's next step is an RMSNorm with a residual, which also takes the
        projection's K-split partial sums (``PartialSums``)."""
        if self.act != "silu":
            return self.down_proj(ops.act_and_mul(self.gate_up_proj(x), self.act), reduce, staged)
        act = self.gate_up_proj.forward_silu(x)       # large batches: activation formed in the GEMM epilogue
        if act is not None:
            if residual is not None and self.down_proj.forward_add(act, residual):
                return None
            if partials and act.shape[0] <= 256:
                part = self.down_proj.forward_partials(act)
                if part is not None:
                    return part
            return self.down_proj(act, reduce, staged)
        gate_up = self.gate_up_proj(x)
        if partials and gate_up.dim() == 2 and gate_up.shape[0] <= 256:
            # short batches: the K-split slabs go to the next add + RMSNorm unreduced
            act = ops.silu_and_mul(gate_up)
            part = self.down_proj.forward_partials(act)
            return part if part is not None else self.down_proj(act, reduce, staged)
        return self.down_proj.forward_gated(gate_up, reduce, staged)


class LlamaAttention(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        hidden = cfg["hidden_size"]
        heads = cfg["num_attention_heads"]
        kv_heads = cfg.get("num_key_value_heads") or heads
        self.head_dim = cfg.get("head_dim") or hidden // heads
        bias = bool(cfg.get("attention_bias", False)) or cfg.get("model_type") in ("qwen2", "qwen2_moe")
        self.qkv_proj = QKVParallelLinear(hidden, self.head_dim, heads, kv_heads, bias, dtype)
        # (GLM biases q / k / v only; Seed-OSS says which in the config)
        o_bias = cfg.get("attention_out_bias")
        if o_bias is None:
            o_bias = bool(cfg.get("attention_bias", False)) and cfg.get("model_type") not in NO_O_BIAS_TYPES
        self.o_proj = RowParallelLinear(heads * self.head_dim, hidden, bool(o_bias), dtype)
        self.num_heads = self.qkv_proj.num_heads
        self.num_kv_heads = self.qkv_proj.num_kv_heads
        self.q_size, self.kv_size = self.qkv_proj.q_size, self.qkv_proj.kv_size
        self.cos_sin = cos_sin
        self.clip_qkv = float(cfg["clip_qkv"]) if cfg.get("clip_qkv") else None      # OLMo: clamp on q, k and v
        self.qk_norm = has_head_qk_norm(cfg)
        self.qk_norm_full = cfg.get("model_type") in FULL_QK_NORM_TYPES
        eps = cfg.get("rms_norm_eps", 1e-6)
        if self.qk_norm:
            self.q_norm = RMSNorm(self.head_dim, eps, dtype)
            self.k_norm = RMSNorm(self.head_dim, eps, dtype)
        elif self.qk_norm_full:
            qp = self.qkv_proj
            self.q_norm = FullWidthRMSNorm(qp.q_size, heads * self.head_dim, qp.tp_rank, 1, eps, dtype)
            self.k_norm = FullWidthRMSNorm(qp.kv_size, kv_heads * self.head_dim, qp.tp_rank // qp.kv_replicas,
                                           qp.kv_replicas, eps, dtype)
        # Granite scales the scores by ``attention_multiplier`` instead of 1 / sqrt(head_dim)
        scale = cfg.get("attention_multiplier") or self.head_dim ** -0.5
        self.attn = Attention(self.num_heads, self.head_dim, self.num_kv_heads, float(scale), layer_idx,
                              layer_sliding_window(cfg, layer_idx), softcap=cfg.get("attn_logit_softcapping"))

    def forward(self, positions: torch.Tensor, hidden: torch.Tensor, reduce: bool = True,
                staged: bool = False, residual: Optional[torch.Tensor] = None, partials: bool = False):
        """With ``residual``: None is returned when the output projection added its product into it.
        ``partials``: the output projection may return its K-split partial sums (``PartialSums``)."""
        md = get_forward_context().attn_metadata
        if (md is not None and self.attn.k_cache is not None and self.clip_qkv is None and not self.qk_norm_full
                and ops.rope_cache_fused_ok(self.head_dim)):
            # decode-sized batches: the projection's K-split partial sums are added up by the kernel that
            # normalises, rotates and writes the cache
            part = self.qkv_proj.forward_partials(hidden)
            if part is not None:
                nw = (self.q_norm.weight, self.k_norm.weight, self.q_norm.eps) if self.qk_norm else (None, None, 0.0)
                q = ops.rope_and_cache_partials(positions, part.slabs, part.dtype, self.cos_sin, self.attn.k_cache,
                                                self.attn.v_cache, md.slot_mapping, self.num_heads,
                                                self.num_kv_heads, self.head_dim, *nw, kv_scales=self.attn.kv_scales)
                return self._project_out(self.attn.forward_cached(q), reduce, staged, residual, partials)
        qkv = self.qkv_proj(hidden)
        if self.clip_qkv is not None and not self.qk_norm_full:
            qkv = qkv.clamp_(-self.clip_qkv, self.clip_qkv)
        if self.qk_norm and md is not None and self.attn.k_cache is not None:
            # per-head RMSNorm of q and k, rotation and the cache write in one kernel
            q = ops.rope_and_cache(positions, qkv, self.cos_sin, self.attn.k_cache, self.attn.v_cache,
                                   md.slot_mapping, self.num_heads, self.num_kv_heads, self.head_dim,
                                   self.q_norm.weight, self.k_norm.weight, self.q_norm.eps,
                                   kv_scales=self.attn.kv_scales)
            out = self.attn.forward_cached(q)
        elif self.qk_norm:
            t = qkv.shape[0]
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            q = self.q_norm(q.reshape(t * self.num_heads, self.head_dim)).view(t, self.q_size)
            k = self.k_norm(k.reshape(t * self.num_kv_heads, self.head_dim)).view(t, self.kv_size)
            ops.rotary_embedding(positions, q, k, self.cos_sin, self.head_dim)
            out = self.attn(q, k, v)
        elif self.qk_norm_full:
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            q, k = self.q_norm(q).contiguous(), self.k_norm(k).contiguous()
            if self.clip_qkv is not None:
                q, k, v = (t.clamp(-self.clip_qkv, self.clip_qkv) for t in (q, k, v))
            ops.rotary_embedding(positions, q, k, self.cos_sin, self.head_dim)
            out = self.attn(q, k, v)
        elif md is None:
            q = qkv[:, :self.q_size]
            out = self.attn.forward_cached(q)
        else:
            q = ops.rope_and_cache(positions, qkv, self.cos_sin, self.attn.k_cache,
                                   self.attn.v_cache, md.slot_mapping, self.num_heads,
                                   self.num_kv_heads, self.head_dim, kv_scales=self.attn.kv_scales)
            out = self.attn.forward_cached(q)
        return self._project_out(out, reduce, staged, residual, partials)

    def _project_out(self, out: torch.Tensor, reduce: bool, staged: bool, residual: Optional[torch.Tensor],
                     partials: bool):
        if residual is not None and self.o_proj.forward_add(out, residual):
            return None
        if partials:
            part = self.o_proj.forward_partials(out)
            if part is not None:
                return part
        return self.o_proj(out, reduce, staged)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        eps = cfg.get("rms_norm_eps", 1e-5)
        self.self_attn = self._make_attn(cfg, layer_idx, dtype, cos_sin)
        self.mlp = self._make_mlp(cfg, dtype)
        self.input_layernorm = RMSNorm(cfg["hidden_size"], eps, dtype)
        self.post_attention_layernorm = RMSNorm(cfg["hidden_size"], eps, dtype)
        # both branches end in a projection whose output goes straight into an RMSNorm with the residual:
        # they may hand over K-split partial sums (unsharded layers; see RowParallelLinear.forward_partials)
        self._attn_partials = type(self.self_attn) is LlamaAttention
        self._mlp_partials = type(self.mlp) is LlamaMLP
        self._mlp_deferred = type(self.mlp) is FusedMoE     # may leave its weighted sum to the next norm

    def _make_attn(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor) -> nn.Module:
        return LlamaAttention(cfg, layer_idx, dtype, cos_sin)

    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        act = cfg.get("hidden_activation") or cfg.get("hidden_act") or "silu"
        return LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype,
                        bool(cfg.get("mlp_bias", False)), "silu" if act in ("silu", "swish") else act)

    # the layer split in two at its all-reduces, for micro-batch overlap
    def attn_phase(self, positions: torch.Tensor, hidden: torch.Tensor,
                   residual: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        if residual is None:
            residual = hidden
            hidden = self.input_layernorm(hidden)
        else:
            hidden, residual = self.input_layernorm(hidden, residual)
        return self.self_attn(positions, hidden, reduce=False), residual

    def mlp_phase(self, hidden: torch.Tensor, residual: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        hidden, residual = self.post_attention_layernorm(hidden, residual)
        return self.mlp(hidden, reduce=False), residual

    def forward(self, positions: torch.Tensor, hidden: Optional[torch.Tensor],
                residual: Optional[torch.Tensor]) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        """``hidden`` None (in or out): the producing GEMM has already added its product into
        ``residual`` (large unsharded batches), so the norm that follows is a plain one."""
        if residual is None:
            residual = hidden
            hidden = self.input_layernorm(hidden)
        elif hidden is None:
            hidden = self.input_layernorm(residual)
        else:
            hidden, residual = self.input_layernorm(hidden, residual)
        if self._attn_partials:
            hidden = self.self_attn(positions, hidden, residual=residual, partials=True)
        else:
            hidden = self.self_attn(positions, hidden, residual=residual)
        if hidden is None:
            hidden = self.post_attention_layernorm(residual)
        else:
            hidden, residual = self.post_attention_layernorm(hidden, residual)
        if self._mlp_partials:
            hidden = self.mlp(hidden, residual=residual, partials=True)
        elif self._mlp_deferred:
            hidden = self.mlp(hidden, deferred=True)
        elif getattr(self.mlp, "supports_residual", False):
            hidden = self.mlp(hidden, residual=residual)
        else:
            hidden = self.mlp(hidden)
        return hidden, residual


class LlamaForCausalLM(nn.Module):
    layer_cls = LlamaDecoderLayer
    # the linear layers are built under ``quantized_linears``: AWQ / NVFP4 / fp8 weights stay packed
    packed_quantization = True
    # checkpoint name fragment -> (fused param fragment, shard id)
    stacked_params = [
        (".q_proj", ".qkv_proj", "q"), (".k_proj", ".qkv_proj", "k"), (".v_proj", ".qkv_proj", "v"),
        (".gate_proj", ".gate_up_proj", 0), (".up_proj", ".gate_up_proj", 1),
    ]

    def __init__(self, vllm_config: VllmConfig):
        super().__init__()
        mc = vllm_config.model_config
        cfg = mc.hf_config
        dtype = mc.torch_dtype
        self.cfg = cfg
        pp = get_pp_group()
        self.is_first, self.is_last = pp.is_first_rank, pp.is_last_rank
        self.start_layer, self.end_layer = self.layer_range(cfg, vllm_config)
        head_dim = cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"]
        rotary_dim = int(head_dim * float(cfg.get("partial_rotary_factor") or 1.0))
        self.rope_perm = rope_row_permutation(head_dim, rotary_dim, cfg.get("model_type") in INTERLEAVED_ROPE_TYPES)
        self.head_dim = head_dim
        cos_sin = ops.build_cos_sin_cache(head_dim, mc.max_model_len, cfg.get("rope_theta", 10000.0),
                                          cfg.get("rope_scaling"), rotary_dim=rotary_dim)
        self.register_buffer("cos_sin_cache", padded_cos_sin(cos_sin, head_dim), persistent=False)
        tie = bool(cfg.get("tie_word_embeddings", False))
        if self.is_first or (tie and self.is_last):
            self.embed_tokens = VocabParallelEmbedding(cfg["vocab_size"], cfg["hidden_size"], dtype,
                                                       out_scale=self.embedding_scale(cfg))
        spec = {"awq": QuantSpec("awq", mc.quant_group_size), "fp8": QuantSpec("fp8"),
                "nvfp4": QuantSpec("nvfp4", 16, 1)}.get(mc.quantization)
        self.quant_method = spec.method if spec is not None else None
        with quantized_linears(spec):
            self.layers = nn.ModuleDict({
                str(i): self.layer_cls(cfg, i, dtype, self.cos_sin_cache)
                for i in range(self.start_layer, self.end_layer)})
        if self.is_last:
            self.norm = self.make_final_norm(cfg, dtype)
            if tie:
                self.lm_head = None
            else:
                self.lm_head = ParallelLMHead(cfg["vocab_size"], cfg["hidden_size"], dtype)
        self.tie = tie
        # forwards (eager calls and graph captures) per form of the TP collectives: get_stats reports it
        import collections
        self.path_counts = collections.Counter()
        # layers whose INPUT (the residual stream in front of them) a draft head reads (EAGLE-3): a forward leaves
        # them side by side in ``last_aux [T, len * H]``; such a model runs the plain layer loop on one stage
        self.aux_hidden_layers: Tuple[int, ...] = ()
        self.last_aux: Optional[torch.Tensor] = None

    # ---- what differs between the Llama-shaped families (see llama_variants.py)
    def layer_range(self, cfg: dict, vllm_config: VllmConfig) -> Tuple[int, int]:
        """The decoder layers this instance builds: its pipeline stage's slice."""
        pp = get_pp_group()
        return get_pp_layer_range(cfg["num_hidden_layers"], pp.rank_in_group, pp.world_size,
                                  vllm_config.parallel_config.pp_layer_partition)

    def embedding_scale(self, cfg: dict) -> Optional[float]:
        """Factor applied to the embedding output (Gemma: sqrt(hidden), Granite: a config value)."""
        return None

    def embed_input(self, input_ids: torch.Tensor) -> torch.Tensor:
        """Embeddings of a step's tokens (vision-language models overwrite the image rows here)."""
        return self.embed_tokens(input_ids)

    def transform_weight(self, name: str, w):
        """Checkpoint tensor -> the tensor this implementation wants under the (mapped) ``name``."""
        return w

    def make_final_norm(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return RMSNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-5), dtype)

    def rebind_buffers(self) -> None:
        """After ``.to(device)`` the layers must see the moved cos/sin table."""
        for layer in self.layers.values():
            layer.self_attn.cos_sin = self.cos_sin_cache

    def attention_layers(self) -> List[Attention]:
        return [layer.self_attn.attn for layer in self.layers.values()]

    # False for families whose layers are not "norm -> branch -> add into the residual" (norms on the
    # branch outputs as well): they run the plain layer loop, without the fused / sharded / overlapped
    # forms of the TP all-reduces, which are written against that layer shape
    tp_fusions = True

    def forward(self, input_ids: Optional[torch.Tensor], positions: torch.Tensor,
                intermediate: Optional[IntermediateTensors] = None):
        aux = self.aux_hidden_layers
        fused = self.fused_tp_mode(positions.shape[0]) if self.tp_fusions else None
        if aux and fused:
            fused = None        # (row-sharded residual stream: the taps would need a gather - the plain loop instead)
        if fused is not None:
            return self.forward_fused_tp(input_ids, positions, intermediate, sharded=fused)
        if intermediate is None and not aux and self.sequence_parallel_ok(positions.shape[0]):
            self.path_counts["sequence_parallel"] += 1
            return self.forward_sequence_parallel(input_ids, positions)
        self.path_counts["plain"] += 1
        if self.is_first:
            hidden = self.embed_input(input_ids)
            residual = None
        else:
            hidden, residual = intermediate.hidden, intermediate.residual
        taps = {}
        for idx, layer in self.layers.items():
            if aux and int(idx) in aux:
                h = hidden if hidden is None or isinstance(hidden, torch.Tensor) else hidden.materialize()
                taps[int(idx)] = residual if h is None else (h if residual is None else h + residual)
            hidden, residual = layer(positions, hidden, residual)
        if aux:
            self.last_aux = torch.cat([taps[i] for i in aux], dim=-1)
        if not self.is_last:
            if hidden is None:                  # the last GEMM already added its product into the residual
                hidden = torch.zeros_like(residual)
            elif not isinstance(hidden, torch.Tensor):
                hidden = hidden.materialize()   # K-split partial sums do not cross a stage boundary
            return IntermediateTensors(hidden, residual)
        if hidden is None:
            return self.norm(residual)
        hidden, _ = self.norm(hidden, residual)
        return hidden

    # ------------------------------------------ fused all-reduce + add + norm
    def fused_tp_mode(self, num_tokens: int) -> Optional[bool]:
        """None: the TP group's peer-mapped collectives do not take this batch (no such path,
        or the messages are too large for it).  Otherwise the batch runs ``forward_fused_tp``:
        False = residual stream replicated (small messages, one pass over every peer's partial
        sums), True = residual stream row-sharded (reduce-scatter + norm on the shard + gather)."""
        tp = get_tp_group()
        h = self.cfg["hidden_size"]
        dtype = self.norm.weight.dtype if self.is_last else next(iter(self.parameters())).dtype
        if tp.world_size == 1 or num_tokens == 0 or not tp.fused_all_reduce_norm_ok(num_tokens, h, dtype):
            return None
        # larger messages: residual stream row-sharded inside the stage; at a stage boundary the
        # shards are gathered once, so pipeline stages exchange the same replicated pair as ever
        return not (num_tokens * h * 2 <= tp.custom_ar.one_shot_bytes)

    def forward_fused_tp(self, input_ids: Optional[torch.Tensor], positions: torch.Tensor,
                         intermediate: Optional[IntermediateTensors], sharded: bool):
        """The decoder with every TP all-reduce folded into the add + RMSNorm that follows it:
        row-parallel GEMMs (and the vocab-parallel embedding) leave their partial sums in the
        peer-mapped stage area and one kernel per norm does reduce + residual add + norm
        (``csrc/custom_all_reduce.hip``).  Per layer that is 2 kernels where the unfused form
        runs 2 copies + 2 all-reduces + 2 norms."""
        tp = get_tp_group()
        layers = list(self.layers.values())
        t, h = positions.shape[0], self.cfg["hidden_size"]
        self.path_counts["fused_sharded" if sharded else "fused_replicated"] += 1
        per, mine = tp.custom_ar.shard_rows(t) if sharded else (t, t)
        lo = min(tp.rank_in_group * per, t)
        if self.is_first:
            part = self.embed_tokens(input_ids, reduce=False, staged=True)
            # the residual starts at zero so that the embedding is reduced by the same fused kernel
            res = torch.zeros((per, h), dtype=part.dtype, device=part.device)
            reduced = False
        else:
            part, res = intermediate.hidden, intermediate.residual
            reduced = True                      # what crossed the stage boundary is already summed

        def add_norm(x: torch.Tensor, norm) -> torch.Tensor:
            nonlocal reduced, res
            if reduced:
                reduced = False
                normed, res = norm(x, res)      # replicated inputs from the previous stage
                if sharded:                     # from here on this rank keeps its rows of the residual only
                    shard = torch.zeros((per, h), dtype=res.dtype, device=res.device)
                    shard[:mine] = res[lo:lo + mine]
                    res = shard
                return normed
            return tp.all_reduce_add_rms_norm(x, res, norm.weight, norm.eps, sharded)

        norm = layers[0].input_layernorm if layers else self.norm
        aux, taps = self.aux_hidden_layers, {}
        for li, layer in enumerate(layers):
            normed = add_norm(part, norm)
            if aux and self.start_layer + li in aux:
                taps[self.start_layer + li] = res.clone()       # the residual stream in front of this layer (replicated)
            part = layer.self_attn(positions, normed, reduce=False, staged=True)
            normed = add_norm(part, layer.post_attention_layernorm)
            part = layer.mlp(normed, reduce=False, staged=True)
            norm = layers[li + 1].input_layernorm if li + 1 < len(layers) else (self.norm if self.is_last else None)
        if aux:
            self.last_aux = torch.cat([taps[i] for i in aux], dim=-1)
        if not self.is_last:
            hidden = part if reduced else tp.all_reduce(part)
            if sharded and not reduced:
                res = tp.all_gather(res, dim=0)[:t]      # the next stage takes the pair replicated
            return IntermediateTensors(hidden, res)
        return add_norm(part, self.norm)

    # ---------------------------------------------------- sequence parallel
    def sequence_parallel_ok(self, num_tokens: int) -> bool:
        """Whether a batch of ``num_tokens`` rows runs with the residual stream
        row-sharded over the TP group.  The all-reduce after each row-parallel
        GEMM becomes reduce-scatter + all-gather (the same bytes on the wire)
        and the add+RMSNorm between them touches ``1/TP`` of the rows, where
        the replicated form makes every rank normalise all of them."""
        mode = envs.VDAMD_SEQUENCE_PARALLEL
        if mode == "0" or not self.tp_fusions or not (self.is_first and self.is_last):
            return False
        w = get_tp_group().world_size
        if num_tokens % w:
            return False
        if mode == "force":
            return True
        return w > 1 and num_tokens >= envs.VDAMD_SP_MIN_TOKENS

    def forward_sequence_parallel(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        tp = get_tp_group()
        layers = list(self.layers.values())
        part = self.embed_tokens(input_ids, reduce=False)
        # residual shard starts at zero so every step below is the same fused add+norm
        res = torch.zeros((part.shape[0] // tp.world_size, part.shape[1]), dtype=part.dtype, device=part.device)
        norm = layers[0].input_layernorm if layers else self.norm
        for li, layer in enumerate(layers):
            normed, res = norm(tp.reduce_scatter(part), res)
            part = layer.self_attn(positions, tp.all_gather(normed, dim=0), reduce=False)
            normed, res = layer.post_attention_layernorm(tp.reduce_scatter(part), res)
            part = layer.mlp(tp.all_gather(normed, dim=0), reduce=False)
            norm = layers[li + 1].input_layernorm if li + 1 < len(layers) else self.norm
        hidden, _ = norm(tp.reduce_scatter(part), res)
        return tp.all_gather(hidden, dim=0)

    def forward_overlapped_sp(self, input_ids: torch.Tensor, positions: torch.Tensor, metadatas, bounds,
                              comm_stream) -> torch.Tensor:
        """``forward_overlapped`` with the sequence-parallel residual stream: the
        communication stream runs reduce-scatter -> add+RMSNorm on the row
        shard -> all-gather for one micro-batch while the compute stream runs
        the other micro-batch's GEMMs.  Buffers the communication stream
        writes are allocated on the compute stream and kept referenced until
        the compute stream has waited for the chain that uses them."""
        from .layers.attention import set_forward_context
        tp = get_tp_group()
        w = tp.world_size
        main = torch.cuda.current_stream()
        layers = list(self.layers.values())
        n = len(bounds)
        pos = [positions[s:e] for s, e in bounds]
        part_all = self.embed_tokens(input_ids, reduce=False)
        res = [torch.zeros(((e - s) // w, part_all.shape[1]), dtype=part_all.dtype, device=part_all.device)
               for s, e in bounds]
        h: List[Optional[torch.Tensor]] = [None] * n
        ready: List[Optional[torch.cuda.Event]] = [None] * n
        keep: List[Optional[tuple]] = [None] * n

        def reduce_norm_gather_async(i: int, part: torch.Tensor, norm) -> None:
            shard = torch.empty_like(res[i])
            full = torch.empty_like(part)
            produced = torch.cuda.Event()
            produced.record(main)
            comm_stream.wait_event(produced)
            with torch.cuda.stream(comm_stream):
                tp.reduce_scatter(part, out=shard)
                normed, res[i] = norm(shard, res[i])
                tp.all_gather_into(full, normed)
                done = torch.cuda.Event()
                done.record(comm_stream)
            h[i], ready[i], keep[i] = full, done, (part, shard, normed)

        first_norm = layers[0].input_layernorm if layers else self.norm
        for i, (s, e) in enumerate(bounds):
            reduce_norm_gather_async(i, part_all[s:e], first_norm)
        for li, layer in enumerate(layers):
            nxt = layers[li + 1].input_layernorm if li + 1 < len(layers) else self.norm
            for i in range(n):
                main.wait_event(ready[i])
                set_forward_context(metadatas[i], pos[i])
                reduce_norm_gather_async(i, layer.self_attn(pos[i], h[i], reduce=False),
                                         layer.post_attention_layernorm)
            for i in range(n):
                main.wait_event(ready[i])
                reduce_norm_gather_async(i, layer.mlp(h[i], reduce=False), nxt)
        for i in range(n):
            main.wait_event(ready[i])
        set_forward_context(None)
        return torch.cat(h)

    def forward_overlapped(self, input_ids: torch.Tensor, positions: torch.Tensor, metadatas, bounds,
                           comm_stream) -> torch.Tensor:
        """Decode forward with the batch cut into micro-batches whose tensor-
        parallel all-reduces run on ``comm_stream`` while the compute stream
        works on the other micro-batch: per layer
        ``attn(A) | attn(B)+AR(A) | mlp(A)+AR(B) | mlp(B)+AR(A) | ...``.
        xGMI all-reduces of large activations cost about as much as the layer's
        GEMMs, so hiding them is worth up to ~2x at TP=8.  ``metadatas[i]`` /
        ``bounds[i] = (start, end)`` describe micro-batch ``i``.  Single pipeline
        stage only."""
        from ..parallel.state import get_tp_group
        from .layers.attention import set_forward_context
        tp = get_tp_group()
        main = torch.cuda.current_stream()
        hidden = self.embed_input(input_ids)
        n = len(bounds)
        h = [hidden[s:e] for s, e in bounds]
        pos = [positions[s:e] for s, e in bounds]
        res: List[Optional[torch.Tensor]] = [None] * n
        ready: List[Optional[torch.cuda.Event]] = [None] * n

        def all_reduce_async(i: int, part: torch.Tensor) -> None:
            produced = torch.cuda.Event()
            produced.record(main)
            comm_stream.wait_event(produced)
            with torch.cuda.stream(comm_stream):
                tp.all_reduce_(part)
                done = torch.cuda.Event()
                done.record(comm_stream)
            h[i], ready[i] = part, done

        for layer in self.layers.values():
            for i in range(n):
                if ready[i] is not None:
                    main.wait_event(ready[i])
                set_forward_context(metadatas[i], pos[i])
                part, res[i] = layer.attn_phase(pos[i], h[i], res[i])
                all_reduce_async(i, part)
            for i in range(n):
                main.wait_event(ready[i])
                part, res[i] = layer.mlp_phase(h[i], res[i])
                all_reduce_async(i, part)
        for i in range(n):
            main.wait_event(ready[i])
        set_forward_context(None)
        hidden, _ = self.norm(torch.cat(h), torch.cat(res))
        return hidden

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        if self.lm_head is not None:
            return self.lm_head.logits(hidden)
        w = self.embed_tokens
        y = torch.nn.functional.linear(hidden, w.weight)
        if w.tp > 1:
            from ..parallel.state import tensor_model_parallel_all_gather
            y = tensor_model_parallel_all_gather(y, dim=-1)
        return y[..., :w.vocab_size]

    def compute_greedy_tokens(self, hidden: torch.Tensor) -> torch.Tensor:
        if self.lm_head is not None:
            return self.lm_head.greedy_tokens(hidden)
        return ops.argmax(self.compute_logits(hidden))

    # ------------------------------------------------------------- weights
    def _owns_layer(self, name: str) -> bool:
        # (names arrive with or without the checkpoint's "model." prefix)
        parts = name.split(".")
        if "layers" not in parts[:-1]:
            return True
        idx = int(parts[parts.index("layers") + 1])
        return self.start_layer <= idx < self.end_layer

    def map_weight_name(self, name: str) -> Optional[str]:
        if name.startswith("model."):
            name = name[len("model."):]
        if "rotary_emb.inv_freq" in name:
            return None
        return name

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        params: Dict[str, nn.Parameter] = dict(self.named_parameters())
        loaded: List[str] = []
        for name, w in weights:
            name = self.map_weight_name(name)
            if name is None or not self._owns_layer(name):
                continue
            if self.quant_method == "nvfp4":
                renamed = nvfp4_rename(name, w)
                if renamed is None:
                    continue
                name, w = renamed
            if self.rope_perm is not None and name.endswith((".weight", ".bias", ".weight_packed", ".weight_block_scale")) and (
                    ".q_proj." in name or ".k_proj." in name
                    or (name.endswith((".q_norm.weight", ".k_norm.weight")) and has_head_qk_norm(self.cfg))):
                # re-order the rows of every head (and the per-head norm weights that act on them) into the
                # basis the rotary kernel works in; NVFP4 bytes and block scales are row-major like the dense weight
                t = materialize(w)
                w = t.reshape(-1, self.head_dim, *t.shape[1:]).index_select(1, self.rope_perm.to(t.device)).reshape(t.shape)
            elif self.rope_perm is not None and (".q_proj." in name or ".k_proj." in name) \
                    and name.endswith((".qweight", ".qzeros", ".scales")):
                # AWQ keeps output channels in the last dimension, eight to a word
                from .loader import awq_reorder_out_columns
                t = materialize(w)
                out = t.shape[1] * (1 if name.endswith(".scales") else 8)
                full = (torch.arange(out // self.head_dim).unsqueeze(1) * self.head_dim + self.rope_perm.cpu()).reshape(-1)
                w = awq_reorder_out_columns(t, full, packed=not name.endswith(".scales"))
            w = self.transform_weight(name, w)
            if name.endswith((".k_scale", ".v_scale")) and ".self_attn." in name:
                # per-tensor scales of an fp8 KV cache (read only when the cache is e4m3)
                layer = self.layers[name.split("layers.")[1].split(".")[0]]
                layer.self_attn.attn.set_kv_scale(name[-7], materialize(w))
                loaded.append(name)
                continue
            target, shard = name, None
            for frag, fused, sid in self.stacked_params:
                if frag in name:
                    target, shard = name.replace(frag, fused), sid
                    break
            p = params.get(target)
            if p is None:
                continue             # e.g. lm_head on a non-last stage
            loader = getattr(p, "weight_loader", None)
            if loader is None:
                p.data.copy_(materialize(w))
            else:
                loader(p, w, shard)
            loaded.append(name)
        return loaded
