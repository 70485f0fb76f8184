"""DeepSeek-V2 / V3: multi-head latent attention (MLA) and a routed expert mixture with shared experts.

Attention: queries come through an optional low-rank pair ``q_a_proj -> RMSNorm -> q_b_proj``; keys and
values come from one latent ``kv_a_proj_with_mqa(x) = [c_kv (kv_lora_rank) | k_rope (qk_rope_head_dim)]``:
``kv_b_proj(RMSNorm(c_kv))`` gives every head its ``k_nope`` and ``v``, and the single rotated ``k_rope``
is shared by all heads.  Two forms are served.  *Latent cache* (default): the cache holds the normalised
``c_kv`` and the rotated ``k_rope`` of a token - once as a key row, once as V^T fragments - and ``kv_b_proj`` is
absorbed: its key half is folded into the query (``q_nope . W_UK``), its value half applied to the attention
output, so every head attends over one shared ``rank + rope``-wide key (``layers.attention.LatentAttention``,
``csrc/mla_attention.hip``).  *Expanded* (``VDAMD_MLA_LATENT=0``, or a latent width the kernel is not built for):
per-head K / V (``[k_nope | k_rope]``, V zero-padded to the same width, both padded to the next size the attention
kernels take) in front of the ordinary paged cache, at ``heads x 2 x 256`` values per token.
The rotary pairs are adjacent dimensions in these checkpoints: the rope rows of ``q_b_proj`` / ``q_proj``
and of ``kv_a_proj_with_mqa`` are re-ordered at load so that the half-apart rotation applies.

Routing: V2 takes the top-k of a softmax (optionally within the best ``topk_group`` of ``n_group`` expert
groups, a group scored by its best expert); V3 scores with a sigmoid, chooses with a per-expert bias
added (``e_score_correction_bias``; a group scored by the sum of its two best), weights with the unbiased
scores, renormalises and multiplies by ``routed_scaling_factor``.  ``n_shared_experts`` always-on experts
are one dense MLP; the first ``first_k_dense_replace`` layers are dense.

GLM-4.5 (``Glm4MoeForCausalLM``) and dots.llm1 (``Dots1ForCausalLM``) put V3's expert block behind ordinary
grouped-query attention.

Parity is pinned against ``transformers``' DeepseekV2 / DeepseekV3 models (``tests/test_hf_parity.py``).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import envs, ops
from ..config import VllmConfig
from ..parallel.state import get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce
from . import qwen3_moe
from .layers.attention import Attention, LatentAttention, RMSNorm, get_forward_context
from .layers.linear import ColumnParallelLinear, ReplicatedLinear, RowParallelLinear, divide
from .layers.moe import FusedMoE
from .llama import LlamaMLP, rope_row_permutation
from .qwen3_moe import Qwen3MoeDecoderLayer, Qwen3MoeForCausalLM
from .utils import materialize


# model types that route the way DeepSeek-V3 does
V3_ROUTING_TYPES = ("deepseek_v3", "glm4_moe", "dots1")


def _yarn_mscale(scale: float, mscale: float) -> float:
    return 1.0 if scale <= 1 else 0.1 * mscale * math.log(scale) + 1.0


class MLAAttention(nn.Module):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__()
        hidden, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        self.nope, self.rope, self.v_dim = cfg["qk_nope_head_dim"], cfg["qk_rope_head_dim"], cfg["v_head_dim"]
        self.qk_dim = self.nope + self.rope
        if self.v_dim > self.qk_dim:
            raise NotImplementedError("value heads wider than the key heads")
        self.kv_rank, q_rank = cfg["kv_lora_rank"], cfg.get("q_lora_rank")
        bias = bool(cfg.get("attention_bias", False))
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = divide(heads, tp)
        if q_rank:
            self.q_a_proj = ReplicatedLinear(hidden, q_rank, bias, dtype)
            self.q_a_layernorm = RMSNorm(q_rank, 1e-6, dtype)
            self.q_b_proj = ColumnParallelLinear(q_rank, [heads * self.qk_dim], False, dtype)
        else:
            self.q_a_proj = None
            self.q_proj = ColumnParallelLinear(hidden, [heads * self.qk_dim], False, dtype)
        self.kv_a_proj_with_mqa = ReplicatedLinear(hidden, self.kv_rank + self.rope, bias, dtype)
        self.kv_a_layernorm = RMSNorm(self.kv_rank, 1e-6, dtype)
        self.kv_b_proj = ColumnParallelLinear(self.kv_rank, [heads * (self.nope + self.v_dim)], False, dtype)
        self.o_proj = RowParallelLinear(heads * self.v_dim, hidden, bias, dtype)
        scale = self.qk_dim ** -0.5
        rs = cfg.get("rope_scaling") or {}
        if rs.get("rope_type", rs.get("type", "default")) != "default" and rs.get("mscale_all_dim"):
            scale *= _yarn_mscale(float(rs["factor"]), float(rs["mscale_all_dim"])) ** 2
        # latent cache (absorbed projections) wherever the kernel for this latent width exists - always on the
        # CPU, where the PyTorch reference serves any width - else per-head keys / values in an ordinary cache
        dk, dv = self.kv_rank + self.rope, self.kv_rank
        on_gpu = torch.empty(0).device.type != "cpu"       # (modules are built under the target device)
        self.latent = bool(envs.VDAMD_MLA_LATENT) and dv % 16 == 0 and (
            not on_gpu or ops.mla_kernel_ok(dk, dv, dtype))
        if self.latent:
            self.attn = LatentAttention(self.num_heads, self.kv_rank, self.rope, float(scale), layer_idx)
        else:
            self.attn = Attention(self.num_heads, self.qk_dim, self.num_heads, float(scale), layer_idx, pad_head_dim=True)
        self._absorbed = None
        self.cos_sin = cos_sin

    def _absorbed_weights(self):
        """``kv_b_proj`` per local head, split into the part that makes keys (folded into the query:
        ``w_uk[h]`` is ``[nope, rank]``) and the part that makes values (applied to the attention output:
        ``w_uv_t[h]`` is ``[rank, v_dim]``)."""
        if self._absorbed is None:
            w = self.kv_b_proj.weight.view(self.num_heads, self.nope + self.v_dim, self.kv_rank)
            self._absorbed = (w[:, :self.nope, :].contiguous(), w[:, self.nope:, :].transpose(1, 2).contiguous())
        return self._absorbed

    FRESH_PREFILL_MIN_TOKENS = 64
    fresh_prefills = 0          # calls of the expanded path in this process (tests and stats read it)

    def _expand_fresh_prefill(self, md) -> bool:
        # Only where the expanded heads fit the LDS-staged prefill kernel (head size <= 128).  DeepSeek-V2 / V3's
        # 192-wide keys would pad to 256, which the register-resident prefill kernel serves at ~135 TFLOP/s:
        # measured on DeepSeek-V2-Lite, 4.1 ms per layer for an 8k prompt against 2.4 ms on the latent kernel
        # (four tokens per workgroup), so those models stay on the latent kernel for prefill as well.
        return (bool(envs.VDAMD_MLA_EXPAND_FRESH_PREFILL) and md.num_prefills > 0 and md.prefill_fresh
                and md.num_prefill_tokens >= self.FRESH_PREFILL_MIN_TOKENS
                and self._expanded_dim() <= 128 and ops.hip_supports_head_dim(self._expanded_dim()))

    def _expanded_dim(self) -> int:
        return next((d for d in Attention.KERNEL_HEAD_DIMS if d >= self.qk_dim), self.qk_dim)

    @staticmethod
    def _fresh_tables(md, positions: torch.Tensor, nd: int, bs: int):
        """A scratch paged layout for the step's fresh prompts: sequence i gets whole blocks of its own, one after
        the other.  (block table ``[prefills, blocks]``, slot of every prefill token, number of scratch blocks);
        computed once per step."""
        cached = getattr(md, "_fresh_prefill_tables", None)
        if cached is None:
            lens = md.seq_lens[nd:].long()
            nblk = (lens + bs - 1) // bs
            start = torch.cumsum(nblk, 0) - nblk
            width = (md.max_prefill_query_len + bs - 1) // bs
            table = (start[:, None] + torch.arange(width, device=lens.device)[None, :]).to(torch.int32)
            rows = md.token_rows(positions)[0][nd:].long() - nd
            slots = (start[rows] * bs + positions[nd:].long()).to(torch.int32)
            total = (md.num_prefill_tokens + bs - 1) // bs + md.num_prefills
            cached = md._fresh_prefill_tables = (table, slots, total, {})
        return cached

    def _fresh_prefill(self, md, positions: torch.Tensor, q: torch.Tensor, latent: torch.Tensor) -> torch.Tensor:
        """Attention of the step's prefill rows (all of them prompt starts) in the expanded form: per-head
        ``[k_nope | k_rope]`` and ``v`` from ``kv_b_proj``, written to a scratch cache in the ordinary layout and
        read back by the paged prefill kernel.  ``q [Tp, H, nope + rope]`` unrotated -> ``[Tp, H * v_dim]``."""
        tp, h, d = q.shape[0], self.num_heads, self._expanded_dim()
        nd = md.num_decode_tokens
        pos = positions[nd:]
        bs = self.attn.k_cache.shape[2]
        table, slots, total, scratch = self._fresh_tables(md, positions, nd, bs)
        key = (h, d, q.dtype)
        if key not in scratch:
            # one pair per step, shared by the layers.  Zeroed: a V^T chunk holds 8 keys, and the ones past a
            # sequence's end are multiplied by probabilities of exactly 0 - which only helps if they are finite
            scratch[key] = (torch.zeros(ops.k_cache_shape(total, h, d, bs), dtype=q.dtype, device=q.device),
                            torch.zeros(ops.v_cache_shape(total, h, d, bs), dtype=q.dtype, device=q.device))
        MLAAttention.fresh_prefills += 1
        sk, sv = scratch[key]
        c = self.kv_a_layernorm(latent[:, :self.kv_rank].contiguous())
        kv = self.kv_b_proj(c).view(tp, h, self.nope + self.v_dim)
        q_rot = q[..., self.nope:].reshape(tp, h * self.rope)
        k_rot = latent[:, self.kv_rank:].contiguous()
        ops.rotary_embedding(pos, q_rot, k_rot, self.cos_sin, self.rope)
        pad = d - self.qk_dim
        qp = F.pad(torch.cat([q[..., :self.nope], q_rot.view(tp, h, self.rope)], dim=-1), (0, pad))
        kp = F.pad(torch.cat([kv[..., :self.nope], k_rot.view(tp, 1, self.rope).expand(tp, h, self.rope)], dim=-1), (0, pad))
        vp = F.pad(kv[..., self.nope:], (0, d - self.v_dim))
        ops.reshape_and_cache(kp, vp, sk, sv, slots)
        out = ops.paged_attention(qp, sk, sv, table, md.prefill_query_start_loc, md.seq_lens[nd:], self.attn.scale,
                                  md.max_prefill_query_len, md.max_prefill_query_len)
        return out.view(tp, h, d)[..., :self.v_dim].reshape(tp, h * self.v_dim)

    def _forward_latent(self, positions: torch.Tensor, q: torch.Tensor, latent: torch.Tensor) -> torch.Tensor:
        t, h = q.shape[0], self.num_heads
        w_uk, w_uv_t = self._absorbed_weights()
        if ops.mla_prep_ok(q) and self.cos_sin.dtype == torch.float32:
            # GPU: norm, both rotations, the query's assembly and the cache write are one kernel
            md = get_forward_context().attn_metadata
            q_lat = torch.bmm(q[..., :self.nope].transpose(0, 1), w_uk)
            q_full = ops.mla_prep(q, self.nope, q_lat, latent, self.kv_a_layernorm.weight, self.kv_a_layernorm.eps,
                                  self.cos_sin, positions, self.attn.k_cache, self.attn.v_cache,
                                  None if md is None else md.slot_mapping)
            if md is not None and self._expand_fresh_prefill(md):
                # prompts that start in this step: their keys and values are all in hand, so attention runs in
                # the per-head form on the LDS-staged prefill kernel (narrow heads only, see
                # _expand_fresh_prefill); the latent rows above are cached for the steps that follow.  Decode
                # rows of a mixed step still read the latent cache.
                nd = md.num_decode_tokens
                fresh = self._fresh_prefill(md, positions, q[nd:], latent[nd:])
                if nd == 0:
                    return fresh
                o_lat = ops.mla_attention(q_full[:nd], self.attn.k_cache, self.attn.v_cache, md.block_table, None,
                                          md.seq_lens, self.attn.scale, md.max_decode_seq_len)
                dec = torch.bmm(o_lat.transpose(0, 1), w_uv_t).transpose(0, 1).reshape(nd, h * self.v_dim)
                return torch.cat([dec, fresh], dim=0)
            o_lat = q_full[..., :self.kv_rank].contiguous() if md is None else self.attn.attend(q_full)
            return torch.bmm(o_lat.transpose(0, 1), w_uv_t).transpose(0, 1).reshape(t, h * self.v_dim)
        c = self.kv_a_layernorm(latent[:, :self.kv_rank].contiguous())
        q_rot = q[..., self.nope:].reshape(t, h * self.rope)
        k_rot = latent[:, self.kv_rank:].contiguous()
        ops.rotary_embedding(positions, q_rot, k_rot, self.cos_sin, self.rope)
        q_lat = torch.bmm(q[..., :self.nope].transpose(0, 1), w_uk)                  # [h, t, rank]
        q_full = torch.cat([q_lat.transpose(0, 1), q_rot.view(t, h, self.rope)], dim=-1)
        o_lat = self.attn.forward_latent(q_full, torch.cat([c, k_rot], dim=-1))      # [t, h, rank]
        return torch.bmm(o_lat.transpose(0, 1), w_uv_t).transpose(0, 1).reshape(t, h * self.v_dim)

    def forward(self, positions: torch.Tensor, hidden: torch.Tensor, reduce: bool = True, staged: bool = False,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        t, h = hidden.shape[0], self.num_heads
        if self.q_a_proj is not None:
            q = self.q_b_proj(self.q_a_layernorm(self.q_a_proj(hidden)))
        else:
            q = self.q_proj(hidden)
        q = q.view(t, h, self.qk_dim)
        latent = self.kv_a_proj_with_mqa(hidden)
        if self.latent:
            return self.o_proj(self._forward_latent(positions, q, latent), reduce, staged)
        kv = self.kv_b_proj(self.kv_a_layernorm(latent[:, :self.kv_rank].contiguous())).view(t, h, self.nope + self.v_dim)
        q_rot = q[..., self.nope:].reshape(t, h * self.rope)
        k_rot = latent[:, self.kv_rank:].contiguous()
        ops.rotary_embedding(positions, q_rot, k_rot, self.cos_sin, self.rope)
        q = torch.cat([q[..., :self.nope], q_rot.view(t, h, self.rope)], dim=-1)
        k = torch.cat([kv[..., :self.nope], k_rot.view(t, 1, self.rope).expand(t, h, self.rope)], dim=-1)
        v = F.pad(kv[..., self.nope:], (0, self.qk_dim - self.v_dim))
        out = self.attn(q.reshape(t, -1), k.reshape(t, -1), v.reshape(t, -1))
        out = out.view(t, h, self.qk_dim)[..., :self.v_dim].reshape(t, h * self.v_dim)
        return self.o_proj(out, reduce, staged)


class DeepseekMoE(FusedMoE):
    """Routed experts + shared experts under one all-reduce; DeepSeek's routing rules."""
    custom_routing = True

    def __init__(self, cfg: dict, dtype: torch.dtype, expert_parallel: bool):
        n = cfg.get("n_routed_experts") or cfg.get("num_experts") or cfg["num_local_experts"]
        super().__init__(cfg["hidden_size"], cfg["moe_intermediate_size"], n, cfg["num_experts_per_tok"],
                         bool(cfg.get("norm_topk_prob", False)), dtype, expert_parallel=expert_parallel)
        self.v3 = cfg.get("model_type") in V3_ROUTING_TYPES
        self.n_group, self.topk_group = int(cfg.get("n_group") or 1), int(cfg.get("topk_group") or 1)
        self.grouped = self.v3 or cfg.get("topk_method") == "group_limited_greedy"
        self.scaling = float(cfg.get("routed_scaling_factor") or 1.0)
        if self.v3:
            self.gate.e_score_correction_bias = nn.Parameter(torch.zeros(n, dtype=torch.float32), requires_grad=False)
        shared = int(cfg.get("n_shared_experts") or 0)
        self.shared_experts = LlamaMLP(cfg["hidden_size"], cfg["moe_intermediate_size"] * shared, dtype) if shared else None

    def _group_mask(self, scores: torch.Tensor, group_scores: torch.Tensor) -> torch.Tensor:
        idx = group_scores.topk(self.topk_group, dim=-1).indices
        mask = torch.zeros_like(group_scores, dtype=torch.bool).scatter_(1, idx, True)
        return mask.unsqueeze(-1).expand(-1, self.n_group, self.num_experts // self.n_group).reshape(scores.shape)

    def route(self, logits: torch.Tensor):
        if self.v3:
            scores = logits.sigmoid()
            choice = scores + self.gate.e_score_correction_bias
            per_group = choice.view(-1, self.n_group, self.num_experts // self.n_group)
            mask = self._group_mask(scores, per_group.topk(min(2, per_group.shape[-1]), dim=-1).values.sum(-1))
            topi = choice.masked_fill(~mask, float("-inf")).topk(self.top_k, dim=-1).indices
            topw = scores.gather(1, topi)
            if self.renormalize:
                topw = topw / (topw.sum(-1, keepdim=True) + 1e-20)
        else:
            scores = torch.softmax(logits, dim=-1)
            if self.grouped:
                per_group = scores.view(-1, self.n_group, self.num_experts // self.n_group)
                scores = scores.masked_fill(~self._group_mask(scores, per_group.max(-1).values), 0.0)
            topw, topi = scores.topk(self.top_k, dim=-1)
            if self.renormalize and self.top_k > 1:
                # DeepSeek-V2's own gate: the picks' scores are renormalised INSTEAD of being scaled
                return topw / (topw.sum(-1, keepdim=True) + 1e-20), topi
        return topw * self.scaling, topi

    def _route_hip(self, logits: torch.Tensor):
        """``route`` as one kernel (``csrc/moe_route.hip:moe_topk_grouped_kernel``): (weights fp32, experts int32)."""
        grouped = self.grouped and self.n_group > 1
        if self.v3:
            scale, renorm = self.scaling, self.renormalize
        else:       # V2's own gate renormalises INSTEAD of scaling
            renorm = self.renormalize and self.top_k > 1
            scale = 1.0 if renorm else self.scaling
        bias = self.gate.e_score_correction_bias if self.v3 else None
        return ops.moe_topk_grouped(logits, bias, self.top_k, self.n_group if grouped else 1,
                                    self.topk_group if grouped else 1, self.v3, self.v3, renorm, scale)

    def _forward_routed(self, x: torch.Tensor) -> torch.Tensor:
        """The routed experts' weighted sum (unreduced).  On the GPU: router GEMM, one routing kernel, slot sort +
        tile table, row gather, two grouped GEMMs with the activation between them, weighted combine."""
        tile_rows = self._grouped_tile_rows(x) if x.dim() == 2 else 0
        if not tile_rows or self.num_experts > 512 or self.top_k > 16 or self.n_group > 64:
            return super().forward(x, reduce=False)
        topw, topi = self._route_hip(self.router_logits(x))
        local = self.local_experts if self.expert_parallel else -1
        src_row, inv, tiles = ops.moe_align_fused(topi, self.num_experts, tile_rows, self.expert_start, local)
        y = self._experts_grouped(x.index_select(0, src_row), tiles, tile_rows)
        return ops.moe_combine(y, topw, topi, inv, self.expert_start, local)

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False) -> torch.Tensor:
        out = self._forward_routed(x)
        if self.shared_experts is not None:
            out = out + self.shared_experts(x, reduce=False)
        if reduce and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class DeepseekDecoderLayer(Qwen3MoeDecoderLayer):
    def _make_attn(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor) -> nn.Module:
        return MLAAttention(cfg, layer_idx, dtype, cos_sin)

    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        n = cfg.get("n_routed_experts") or cfg.get("num_experts") or cfg.get("num_local_experts")
        dense = (not n or self._layer_idx < int(cfg.get("first_k_dense_replace") or 0)
                 or self._layer_idx % int(cfg.get("moe_layer_freq") or 1) != 0)
        if dense:
            return LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype)
        return DeepseekMoE(cfg, dtype, qwen3_moe._EP)


class DeepseekV2ForCausalLM(Qwen3MoeForCausalLM):
    layer_cls = DeepseekDecoderLayer
    # the MLA block has its own projections: nothing of Llama's q / k / v stacking applies
    stacked_params = [(".gate_proj", ".gate_up_proj", 0), (".up_proj", ".gate_up_proj", 1)]

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        cfg["head_dim"] = cfg["qk_rope_head_dim"]           # what the cos / sin table spans
        if vllm_config.model_config.quantization in ("awq", "nvfp4"):
            raise NotImplementedError("4-bit DeepSeek checkpoints (the rope rows of packed projections cannot be "
                                      "re-ordered at load)")
        super().__init__(vllm_config)
        interleaved = cfg.get("model_type") == "deepseek_v2" or bool(cfg.get("rope_interleave", True))
        self.mla_perm = rope_row_permutation(cfg["qk_rope_head_dim"], cfg["qk_rope_head_dim"], interleaved)

    def transform_weight(self, name: str, w):
        if self.mla_perm is None or not name.endswith((".weight", ".bias")):
            return w
        nope, rope = self.cfg["qk_nope_head_dim"], self.cfg["qk_rope_head_dim"]
        if ".self_attn.q_b_proj." in name or ".self_attn.q_proj." in name:
            t = materialize(w)
            per_head = t.reshape(-1, nope + rope, *t.shape[1:])
            perm = torch.cat([torch.arange(nope, device=t.device), nope + self.mla_perm.to(t.device)])
            return per_head.index_select(1, perm).reshape(t.shape)
        if ".self_attn.kv_a_proj_with_mqa." in name:
            t = materialize(w)
            rank = self.cfg["kv_lora_rank"]
            perm = torch.cat([torch.arange(rank, device=t.device), rank + self.mla_perm.to(t.device)])
            return t.index_select(0, perm)
        return w


class DeepseekV3ForCausalLM(DeepseekV2ForCausalLM):
    pass


class V3RoutedDecoderLayer(Qwen3MoeDecoderLayer):
    """Llama's attention (with the family's own switches) + DeepSeek-V3's expert block."""
    _make_mlp = DeepseekDecoderLayer._make_mlp


class Glm4MoeForCausalLM(Qwen3MoeForCausalLM):
    """GLM-4.5: rotary embedding over half of each head, optional per-head q/k norms, q/k/v biases."""
    layer_cls = V3RoutedDecoderLayer


class Dots1ForCausalLM(Qwen3MoeForCausalLM):
    """dots.llm1: Qwen3's attention (per-head q/k norms), windows per ``layer_types``."""
    layer_cls = V3RoutedDecoderLayer


class ErnieMoE(FusedMoE):
    """ERNIE-4.5 MoE block: softmax scores; the top-k is taken on ``scores + bias`` (a stored per-expert
    balance term), the weights are the unbiased scores of the picks, renormalised with a floor on the sum."""
    custom_routing = True

    def __init__(self, cfg: dict, dtype: torch.dtype, expert_parallel: bool):
        n = int(cfg.get("moe_num_experts") or cfg["num_experts"])
        super().__init__(cfg["hidden_size"], cfg["moe_intermediate_size"], n, cfg["moe_k"],
                         True, dtype, expert_parallel=expert_parallel)
        self.norm_min = float(cfg.get("moe_norm_min") or 1e-12)
        # the router runs in fp32 (weights stored and multiplied in fp32): the biased top-k over near-tied
        # softmax scores picks other experts at bf16 resolution
        self.gate.weight.data = self.gate.weight.data.float()
        self.gate.e_score_correction_bias = nn.Parameter(torch.zeros(n, dtype=torch.float32),
                                                         requires_grad=False)
        self.gate.e_score_correction_bias.weight_loader = \
            lambda p, w, shard=None: p.data.copy_(materialize(w).reshape(-1))          # stored [1, E]
        shared = int(cfg.get("moe_num_shared_experts") or 0)
        self.shared_experts = LlamaMLP(cfg["hidden_size"], cfg["moe_intermediate_size"] * shared, dtype,
                                       bool(cfg.get("mlp_bias", False))) if shared else None

    def router_logits(self, x: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.linear(x.float(), self.gate.weight.float())

    def route(self, logits: torch.Tensor):
        scores = torch.softmax(logits.float(), dim=-1)
        topi = (scores + self.gate.e_score_correction_bias).topk(self.top_k, dim=-1).indices
        topw = scores.gather(1, topi)
        return topw / topw.sum(-1, keepdim=True).clamp(min=self.norm_min), topi

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False) -> torch.Tensor:
        out = FusedMoE.forward(self, x, reduce=False)
        if self.shared_experts is not None:
            out = out + self.shared_experts(x, reduce=False)
        if reduce and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class ErnieMoeDecoderLayer(Qwen3MoeDecoderLayer):
    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        i, last = self._layer_idx, cfg.get("moe_layer_end_index")
        last = cfg["num_hidden_layers"] - 1 if last in (None, -1) else int(last)
        sparse = ((i + 1) % int(cfg.get("moe_layer_interval") or 1) == 0
                  and int(cfg.get("moe_layer_start_index") or 0) <= i <= last)
        if sparse:
            return ErnieMoE(cfg, dtype, qwen3_moe._EP)
        return LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype, bool(cfg.get("mlp_bias", False)))


class Ernie4_5_MoeForCausalLM(Qwen3MoeForCausalLM):
    """ERNIE-4.5 MoE (text): ERNIE-4.5's attention (rotary embedding over adjacent pairs) + ``ErnieMoE`` on the
    layers ``moe_layer_start_index .. moe_layer_end_index`` (every ``moe_layer_interval``-th)."""
    layer_cls = ErnieMoeDecoderLayer

    def __init__(self, vllm_config: VllmConfig):
        cfg = vllm_config.model_config.hf_config
        if cfg.get("use_bias"):
            raise NotImplementedError("ERNIE-4.5 MoE checkpoints with use_bias (biased experts and LM head)")
        cfg.setdefault("moe_k", cfg.get("num_experts_per_tok") or 2)
        cfg["num_experts_per_tok"] = cfg["moe_k"]
        super().__init__(vllm_config)

    def map_weight_name(self, name: str) -> Optional[str]:
        name = super().map_weight_name(name)
        # (the balance term sits under the router in newer checkpoints, next to it in the original ones)
        return None if name is None else name.replace(".gate.moe_statics.", ".gate.").replace(".mlp.moe_statics.", ".mlp.gate.")
